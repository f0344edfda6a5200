// BasicEncoder / SmallEncoder forward (reference tf_raft/layers/extractor.py:6-49, 88-175) on the
// halo-tiled fp32-MFMA convolution of conv_halo.h, for gfx950.
//
//   stem   7x7 stride-2 convolution of the 3-channel image: the image is first padded to 4 channels
//          (prep kernel, optionally fusing the model's 2*(x/255)-1), so that one kernel row of one output
//          pixel is 7 px x 4 ch = 28 contiguous floats = one 32-wide K chunk (4 zero columns).
//   norm   'batch' (inference) is folded into the convolution weights by the host packer, so those
//          encoders are convolutions with relu / residual epilogues only.
//          'instance' needs per-(image, channel) moments of every convolution output: the convolution
//          epilogue writes per-tile (sum, sum of squares), a small kernel finalises them in fp64 into
//          scale = gamma * rsqrt(var + 1e-3), shift = beta - mean * scale, and the CONSUMER applies
//          relu(x * scale + shift) while staging its input tile (no normalised tensor is ever written),
//          except at ResBlock outputs, which are materialised once by res_merge_kernel.
//   TF semantics: stride-2 'same' convolutions pad asymmetrically (extra pixel after); the 1x1 stride-2
//   down-sampling convolution is 'valid'.
#include "conv_halo.h"
#include "conv_wino.h"

// F(4x4, 3x3) launcher (conv_wino4.hip; the kernel header is not needed here)
int raft_launch_conv_wino4(const ConvArgs &a, int epi, hipStream_t s, int decide_npad = 0, int ks_hint = 0);

namespace {

// ---------------------------------------------------------------- small kernels
// pixels [0, npix_a) come from img, the rest from img_b (the feature encoder's [image1, image2] batch: extractor.py:114-116
// concatenates them; here they are staged from where they lie)
__global__ void __launch_bounds__(256) enc_prep_kernel(const float *__restrict__ img, const float *__restrict__ img_b,
                                                       int64_t npix_a, f32x4 *__restrict__ out, int64_t npix, int affine) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const float *src = i < npix_a ? img + 3 * i : img_b + 3 * (i - npix_a);
    float r = src[0], g = src[1], b = src[2];
    if (affine) {   // reference model.py:70-71: 2 * (image / 255) - 1, same operation order
#pragma clang fp contract(off)
        r = 2.f * (r / 255.f) - 1.f;
        g = 2.f * (g / 255.f) - 1.f;
        b = 2.f * (b / 255.f) - 1.f;
    }
    out[i] = f32x4{r, g, b, 0.f};
}

// partial (sum, sumsq) [n_img * tiles][npad][2] -> scale/shift [n_img][C]; one workgroup per (image, 64 channels)
__global__ void __launch_bounds__(1024) in_finalize_kernel(const float *__restrict__ part, int tiles, int npad, int C,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           double inv_count, float *__restrict__ scale,
                                                           float *__restrict__ shift) {
    __shared__ double sh[2][16][64];
    const int img = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    double a = 0.0, q = 0.0;
    if (c < C) {
        const float2 *src = (const float2 *)part + ((int64_t)img * tiles) * npad + c;
        // four loads in flight per thread (the partial records are 56 .. 900 per image: a dependent chain of L2 round
        // trips otherwise); fixed summation order, so the result stays deterministic
        double a1 = 0.0, q1 = 0.0, a2 = 0.0, q2 = 0.0, a3 = 0.0, q3 = 0.0;
        int t = grp;
        for (; t + 48 < tiles; t += 64) {
            const float2 v0 = src[(int64_t)t * npad], v1 = src[(int64_t)(t + 16) * npad];
            const float2 v2 = src[(int64_t)(t + 32) * npad], v3 = src[(int64_t)(t + 48) * npad];
            a += (double)v0.x; q += (double)v0.y;
            a1 += (double)v1.x; q1 += (double)v1.y;
            a2 += (double)v2.x; q2 += (double)v2.y;
            a3 += (double)v3.x; q3 += (double)v3.y;
        }
        for (; t < tiles; t += 16) {
            const float2 v = src[(int64_t)t * npad];
            a += (double)v.x;
            q += (double)v.y;
        }
        a = (a + a1) + (a2 + a3);
        q = (q + q1) + (q2 + q3);
    }
    sh[0][grp][threadIdx.x & 63] = a;
    sh[1][grp][threadIdx.x & 63] = q;
    __syncthreads();
    if (grp == 0 && c < C) {
        for (int g = 1; g < 16; ++g) {
            a += sh[0][g][threadIdx.x];
            q += sh[1][g][threadIdx.x];
        }
        const double mean = a * inv_count;
        double var = q * inv_count - mean * mean;   // biased variance (tfa InstanceNormalization)
        if (var < 0.0) var = 0.0;
        const double sc = (double)gamma[c] / sqrt(var + 1e-3);
        scale[(int64_t)img * C + c] = (float)sc;
        shift[(int64_t)img * C + c] = (float)((double)beta[c] - mean * sc);
    }
}

// y = relu(T(x) + relu(f * fs + fh));  T(x) = x (xs == nullptr) or x * xs + xh (normalised shortcut, no relu)
// mode 1: y = relu(f * fs + fh) (stem output)
__global__ void __launch_bounds__(256) res_merge_kernel(const f32x4 *__restrict__ x, const float *__restrict__ xs,
                                                        const float *__restrict__ xh, const f32x4 *__restrict__ f,
                                                        const float *__restrict__ fs, const float *__restrict__ fh,
                                                        f32x4 *__restrict__ y, int64_t total4, int c4n, int64_t pix_per_img,
                                                        int mode) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c4 = (int)(i % c4n);
    const int64_t img = (i / c4n) / pix_per_img;
    const f32x4 s = *(const f32x4 *)(fs + (img * c4n + c4) * 4), h = *(const f32x4 *)(fh + (img * c4n + c4) * 4);
    f32x4 v = f[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], s[e], h[e]), 0.f);
    if (mode == 0) {
        f32x4 xv = x[i];
        if (xs) {
            const f32x4 s2 = *(const f32x4 *)(xs + (img * c4n + c4) * 4), h2 = *(const f32x4 *)(xh + (img * c4n + c4) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = fmaf(xv[e], s2[e], h2[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(xv[e] + v[e], 0.f);
    }
    y[i] = v;
}

// ---------------------------------------------------------------- convolution dispatch
enum EncKind { ENC_3x3_S1, ENC_3x3_S2, ENC_1x1_S2, ENC_1x1_S1, ENC_STEM };

template <int KH, int KW, int EPI, int STRIDE, int PRE, int STATS, int STEM>
int launch_tile(const ConvArgs &a, int th, int tn, hipStream_t s) {
    const int tiles = a.B * ((a.H + th - 1) / th) * ((a.W + 15) / 16);
    const int grid = tiles * (a.npad / (64 * tn));
    const int key = th * 10 + tn + (raft_conv_deep(a, th, tn, grid) ? 100 : 0);
    switch (key) {
        case 171: conv_halo_kernel<KH, KW, 7, 1, EPI, STRIDE, PRE, STATS, STEM, 1><<<grid, 256, 0, s>>>(a); break;
        case 181: conv_halo_kernel<KH, KW, 8, 1, EPI, STRIDE, PRE, STATS, STEM, 1><<<grid, 256, 0, s>>>(a); break;
        case 71: conv_halo_kernel<KH, KW, 7, 1, EPI, STRIDE, PRE, STATS, STEM><<<grid, 256, 0, s>>>(a); break;
        case 72: conv_halo_kernel<KH, KW, 7, 2, EPI, STRIDE, PRE, STATS, STEM><<<grid, 256, 0, s>>>(a); break;
        case 81: conv_halo_kernel<KH, KW, 8, 1, EPI, STRIDE, PRE, STATS, STEM><<<grid, 256, 0, s>>>(a); break;
        case 82: conv_halo_kernel<KH, KW, 8, 2, EPI, STRIDE, PRE, STATS, STEM><<<grid, 256, 0, s>>>(a); break;
        default: return RAFT_E_UNSUPPORTED;
    }
    return raft_launch_status();
}

// epi: EPI_LINEAR (+stats when a.stats), EPI_RELU, EPI_RES; pre = a.pre_scale != nullptr
int enc_conv(const ConvArgs &a, EncKind kind, int epi, int th, int tn, hipStream_t s) {
    const bool pre = a.pre_scale != nullptr, stats = a.stats != nullptr;
    switch (kind) {
        case ENC_3x3_S1:
            if (epi == EPI_LINEAR && stats && !pre) return launch_tile<3, 3, EPI_LINEAR, 1, 0, 1, 0>(a, th, tn, s);
            if (epi == EPI_LINEAR && stats && pre) return launch_tile<3, 3, EPI_LINEAR, 1, 1, 1, 0>(a, th, tn, s);
            if (epi == EPI_RELU && !pre) return launch_tile<3, 3, EPI_RELU, 1, 0, 0, 0>(a, th, tn, s);
            if (epi == EPI_RES && !pre) return launch_tile<3, 3, EPI_RES, 1, 0, 0, 0>(a, th, tn, s);
            break;
        case ENC_3x3_S2:
            if (epi == EPI_LINEAR && stats && !pre) return launch_tile<3, 3, EPI_LINEAR, 2, 0, 1, 0>(a, th, tn, s);
            if (epi == EPI_RELU && !pre) return launch_tile<3, 3, EPI_RELU, 2, 0, 0, 0>(a, th, tn, s);
            break;
        case ENC_1x1_S2:
            if (epi == EPI_LINEAR && stats && !pre) return launch_tile<1, 1, EPI_LINEAR, 2, 0, 1, 0>(a, th, tn, s);
            if (epi == EPI_LINEAR && !stats && !pre) return launch_tile<1, 1, EPI_LINEAR, 2, 0, 0, 0>(a, th, tn, s);
            break;
        case ENC_1x1_S1:
            if (epi == EPI_LINEAR && !stats && !pre) return launch_tile<1, 1, EPI_LINEAR, 1, 0, 0, 0>(a, th, tn, s);
            break;
        case ENC_STEM:
            if (epi == EPI_LINEAR && stats) return launch_tile<1, 1, EPI_LINEAR, 2, 0, 1, 1>(a, th, tn, s);
            if (epi == EPI_RELU) return launch_tile<1, 1, EPI_RELU, 2, 0, 0, 1>(a, th, tn, s);
            break;
    }
    return RAFT_E_UNSUPPORTED;
}

// tile choice: 7-row tiles when they divide the height (56 = 8 x 7, 112, 224), else 8-row tiles; channel
// blocks of 128 (TN = 2) only when npad is not a multiple of... kept simple: TN = 1 unless overridden.
void enc_pick(int H, int npad, int *th, int *tn) {
    *th = (H % 7 == 0) ? 7 : 8;
    *tn = 1;
    (void)npad;
}

void same_pad(int in, int k, int stride, int *out, int *before) {
    *out = (in + stride - 1) / stride;
    int total = (*out - 1) * stride + k - in;
    if (total < 0) total = 0;
    *before = total / 2;
}

struct EncBufs {
    float *img4, *x, *y, *r1, *r2, *rd, *part, *ss;   // ss: scale/shift slots
};

int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

}   // namespace

extern "C" int64_t raft_encoder_workspace_floats(const raft_encoder_weights *w, int n, int H, int W) {
    if (!w || n <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t h1 = (H + 1) / 2, w1 = (W + 1) / 2;
    int cmax = w->c0 > w->c1 ? w->c0 : w->c1;
    if (w->c2 > cmax) cmax = w->c2;
    if (w->c3 > cmax) cmax = w->c3;
    const int64_t act = align4((int64_t)n * h1 * w1 * cmax);
    int64_t tiles = (int64_t)n * ((h1 + 6) / 7) * ((w1 + 15) / 16);
    const int64_t tiles_w = (int64_t)n * 2 * ((h1 + 3) / 4) * ((w1 + 31) / 32);   // winograd kernel: entries per row block
    if (tiles_w > tiles) tiles = tiles_w;
    const int64_t part = align4(tiles * 256 * 2);
    const int64_t ss = align4((int64_t)n * 256 * 2 * 3);
    return align4((int64_t)n * H * W * 4) + 5 * act + part + ss;
}

static int encoder_impl(const raft_encoder_weights *w, const float *images, const float *images_b, int n_a, int n, int H, int W,
                        int input_affine, float *out, float *workspace, void *stream);

extern "C" int raft_encoder_f32(const raft_encoder_weights *w, const float *images, int n, int H, int W,
                                int input_affine, float *out, float *workspace, void *stream) {
    return encoder_impl(w, images, images, n, n, H, W, input_affine, out, workspace, stream);
}

extern "C" int raft_encoder_pair_f32(const raft_encoder_weights *w, const float *images_a, const float *images_b, int n_each,
                                     int H, int W, int input_affine, float *out, float *workspace, void *stream) {
    RAFT_REQUIRE_PTR(images_b);
    RAFT_REQUIRE(n_each > 0, RAFT_E_SHAPE);
    return encoder_impl(w, images_a, images_b, n_each, 2 * n_each, H, W, input_affine, out, workspace, stream);
}

static int encoder_impl(const raft_encoder_weights *w, const float *images, const float *images_b, int n_a, int n, int H, int W,
                        int input_affine, float *out, float *workspace, void *stream) {
    RAFT_REQUIRE_PTR(w);
    RAFT_REQUIRE_PTR(images);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE_PTR(workspace);
    RAFT_REQUIRE(n > 0 && H > 0 && W > 0, RAFT_E_SHAPE);
    RAFT_REQUIRE(w->c0 % 32 == 0 && w->c1 % 32 == 0 && w->c2 % 32 == 0 && w->c3 % 32 == 0, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(w->c0 <= 256 && w->c1 <= 256 && w->c2 <= 256 && w->c3 <= 256 && w->cout > 0, RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE(w->norm == RAFT_NORM_NONE || w->norm == RAFT_NORM_INSTANCE || w->norm == RAFT_NORM_FOLDED,
                 RAFT_E_UNSUPPORTED);
    RAFT_REQUIRE((int64_t)n * H * W * 4 * 4 < ((int64_t)1 << 31), RAFT_E_UNSUPPORTED);
    hipStream_t s = (hipStream_t)stream;
    const bool inorm = w->norm == RAFT_NORM_INSTANCE;

    // ---- workspace carve-up (same arithmetic as raft_encoder_workspace_floats)
    const int64_t h1 = (H + 1) / 2, w1 = (W + 1) / 2;
    int cmax = w->c0 > w->c1 ? w->c0 : w->c1;
    if (w->c2 > cmax) cmax = w->c2;
    if (w->c3 > cmax) cmax = w->c3;
    const int64_t act = align4((int64_t)n * h1 * w1 * cmax);
    int64_t tiles_max = (int64_t)n * ((h1 + 6) / 7) * ((w1 + 15) / 16);
    {
        const int64_t tiles_w = (int64_t)n * 2 * ((h1 + 3) / 4) * ((w1 + 31) / 32);
        if (tiles_w > tiles_max) tiles_max = tiles_w;
    }
    const bool use_wino = raft_opt(RAFT_OPT_ENC_WINO, 1) != 0;   // 0: direct 3x3 kernels everywhere (A/B timing, parity tests)
    // stages whose stride-1 3x3 layers run the F(4x4, 3x3) kernel (bit 0 = layer1 ...).  Default: every layer whose launch is
    // MORE than one round of the kernel's 8 x 64-pixel x 64-channel workgroups on the chip (> 256) -- at 4 pairs fnet's layer1 / layer2
    // (896 / 448) and cnet's layer1 (448).  Per kernel at 4 pairs (profiles/r07u_encoder_kernels_b4.txt): fnet's ten layers
    // 1308 -> 1120 us, the 64-channel half-resolution layers ~190 -> ~150 us each (K = 64 is only four 16-channel chunks:
    // prologue and output transform are 40 % of a workgroup); a layer3 launch with the K-split variant is slower than F(2x2)
    // (41 against 28 us).  Launches of a single round gain nothing measurable (one / two pairs: 137.2 / 203.8 pairs/s without,
    // 137.2 / 204.5 with every stage on it, profiles/r08k_round3_options.txt) and F(4x4) is the noisier algorithm (3.3e-6 against
    // 1.9e-6 of the output scale), so they stay on F(2x2).  An explicit RAFT_ENC_WINO4 is taken as given.
    const int wino4_mask = use_wino ? raft_opt(RAFT_OPT_ENC_WINO4, 7) : 0;
    const bool wino4_forced = raft_opt_is_set(RAFT_OPT_ENC_WINO4);
    EncBufs b;
    float *p = workspace;
    b.img4 = p; p += align4((int64_t)n * H * W * 4);
    b.x = p; p += act;
    b.y = p; p += act;
    b.r1 = p; p += act;
    b.r2 = p; p += act;
    b.rd = p; p += act;
    b.part = p; p += align4(tiles_max * 256 * 2);
    b.ss = p;
    float *ss_slot[3][2];
    for (int k = 0; k < 3; ++k) {
        ss_slot[k][0] = b.ss + (int64_t)k * n * 256 * 2;
        ss_slot[k][1] = ss_slot[k][0] + (int64_t)n * 256;
    }

    {   // image -> 4-channel padded (and normalised) image
        const int64_t npix = (int64_t)n * H * W;
        enc_prep_kernel<<<raft_ceil_div(npix, 256), 256, 0, s>>>(images, images_b, (int64_t)n_a * H * W, (f32x4 *)b.img4, npix, input_affine);
        RAFT_TRY(raft_launch_status());
    }

    // one convolution (+ instance-norm moments -> scale/shift slot `slot`)
    auto conv = [&](EncKind kind, const raft_conv_weights &cw_direct, const float *in, int cin, int Hi, int Wi, int Ho, int Wo,
                    int pt, int pl, int cout, int epi, float *dst, const float *res, const float *pre_sc,
                    const float *pre_sh, int slot, const float *gamma, const float *beta,
                    const raft_conv_weights *cw_wino = nullptr, const raft_conv_weights *cw_wino4 = nullptr) -> int {
        const bool wino4 = kind == ENC_3x3_S1 && cw_wino4 && cw_wino4->wp != nullptr && cin % 16 == 0;
        const bool wino = !wino4 && use_wino && kind == ENC_3x3_S1 && cw_wino && cw_wino->wp != nullptr;
        const raft_conv_weights &cw = wino4 ? *cw_wino4 : (wino ? *cw_wino : cw_direct);
        ConvArgs a = {};
        a.a0 = in; a.lda0 = (kind == ENC_STEM) ? 4 : cin; a.c0 = (kind == ENC_STEM) ? 7 * 32 : cin;
        a.wp = cw.wp; a.bias = cw.bias; a.npad = cw.npad; a.nvalid = cout;
        a.B = n; a.H = Ho; a.W = Wo; a.Hi = Hi; a.Wi = Wi; a.pt = pt; a.pl = pl;
        a.scale = 1.0f; a.o0 = dst; a.ldo0 = cout;
        a.e0 = res; a.lde0 = cout;
        a.pre_scale = pre_sc; a.pre_shift = pre_sh;
        a.stats = (inorm && slot >= 0) ? b.part : nullptr;
        int th, tn;
        enc_pick(Ho, cw.npad, &th, &tn);
        int rc = wino4 ? raft_launch_conv_wino4(a, epi, s) : (wino ? raft_launch_conv_wino(a, epi, s) : enc_conv(a, kind, epi, th, tn, s));
        if (rc != RAFT_OK) return rc;
        if (a.stats) {
            const int tiles = wino4  ? 2 * ((Ho + 7) / 8) * ((Wo + 63) / 64)
                              : wino ? 2 * ((Ho + 3) / 4) * ((Wo + 31) / 32)
                                     : ((Ho + th - 1) / th) * ((Wo + 15) / 16);
            dim3 grid((cout + 63) / 64, n);
            in_finalize_kernel<<<grid, 1024, 0, s>>>(b.part, tiles, cw.npad, cout, gamma, beta, 1.0 / ((double)Ho * Wo),
                                                     ss_slot[slot][0], ss_slot[slot][1]);
            rc = raft_launch_status();
        }
        return rc;
    };
    auto merge = [&](const float *x, const float *xs, const float *xh, const float *f, const float *fs, const float *fh,
                     float *y, int Ho, int Wo, int C, int mode) -> int {
        const int64_t total4 = (int64_t)n * Ho * Wo * C / 4;
        res_merge_kernel<<<raft_ceil_div(total4, 256), 256, 0, s>>>((const f32x4 *)x, xs, xh, (const f32x4 *)f, fs, fh,
                                                                    (f32x4 *)y, total4, C / 4, (int64_t)Ho * Wo, mode);
        return raft_launch_status();
    };

    // ---- stem: conv1 7x7/2 + norm1 + relu (extractor.py:116-118)
    int Hc, Wc, pt, pl;
    same_pad(H, 7, 2, &Hc, &pt);
    same_pad(W, 7, 2, &Wc, &pl);
    int C = w->c0;
    if (inorm) {
        RAFT_TRY(conv(ENC_STEM, w->conv1, b.img4, 0, H, W, Hc, Wc, pt, pl, C, EPI_LINEAR, b.r1, nullptr, nullptr, nullptr,
                          0, w->in_gamma[0], w->in_beta[0]));
        RAFT_TRY(merge(nullptr, nullptr, nullptr, b.r1, ss_slot[0][0], ss_slot[0][1], b.x, Hc, Wc, C, 1));
    } else {
        RAFT_TRY(conv(ENC_STEM, w->conv1, b.img4, 0, H, W, Hc, Wc, pt, pl, C, EPI_RELU, b.x, nullptr, nullptr, nullptr,
                          -1, nullptr, nullptr));
    }

    // ---- layer1..3: 2 ResBlocks each (extractor.py:41-49, 120-125)
    float *x = b.x, *y = b.y;
    const int widths[3] = {w->c1, w->c2, w->c3};
    for (int blk = 0; blk < 6; ++blk) {
        const int F = widths[blk / 2];
        const int stride = (blk == 2 || blk == 4) ? 2 : 1;
        const raft_conv_weights &c1 = w->block[blk][0], &c2 = w->block[blk][1], &cd = w->block[blk][2];
        int Ho = Hc, Wo = Wc, p1t = 1, p1l = 1;
        if (stride == 2) {
            same_pad(Hc, 3, 2, &Ho, &p1t);
            same_pad(Wc, 3, 2, &Wo, &p1l);
        }
        const EncKind k1 = stride == 2 ? ENC_3x3_S2 : ENC_3x3_S1;
        const int ni = 1 + blk * 3;   // index of this block's norm1 in in_gamma / in_beta
        const bool w4 = ((wino4_mask >> (blk / 2)) & 1) &&
                        (wino4_forced || (int64_t)n * ((Ho + 7) / 8) * ((Wo + 63) / 64) * ((F + 63) / 64) * raft_concurrency() > 256);   // loops sharing the chip: the launch counts raft_concurrency() times (378.7 against 375.6 pairs/s with every stage on F(4x4) under three lanes, profiles/r12l_*)
        const raft_conv_weights *w44a = w4 ? &w->block_w44[blk][0] : nullptr, *w44b = w4 ? &w->block_w44[blk][1] : nullptr;
        if (inorm) {
            RAFT_TRY(conv(k1, c1, x, C, Hc, Wc, Ho, Wo, p1t, p1l, F, EPI_LINEAR, b.r1, nullptr, nullptr, nullptr, 0,
                              w->in_gamma[ni], w->in_beta[ni], &w->block_w[blk][0], w44a));
            RAFT_TRY(conv(ENC_3x3_S1, c2, b.r1, F, Ho, Wo, Ho, Wo, 1, 1, F, EPI_LINEAR, b.r2, nullptr, ss_slot[0][0],
                              ss_slot[0][1], 1, w->in_gamma[ni + 1], w->in_beta[ni + 1], &w->block_w[blk][1], w44b));
            if (stride == 2) {
                RAFT_REQUIRE(cd.wp != nullptr, RAFT_E_NULL);
                RAFT_TRY(conv(ENC_1x1_S2, cd, x, C, Hc, Wc, Ho, Wo, 0, 0, F, EPI_LINEAR, b.rd, nullptr, nullptr, nullptr, 2,
                                  w->in_gamma[ni + 2], w->in_beta[ni + 2]));
                RAFT_TRY(merge(b.rd, ss_slot[2][0], ss_slot[2][1], b.r2, ss_slot[1][0], ss_slot[1][1], y, Ho, Wo, F, 0));
            } else {
                RAFT_TRY(merge(x, nullptr, nullptr, b.r2, ss_slot[1][0], ss_slot[1][1], y, Ho, Wo, F, 0));
            }
        } else {
            RAFT_TRY(conv(k1, c1, x, C, Hc, Wc, Ho, Wo, p1t, p1l, F, EPI_RELU, b.r1, nullptr, nullptr, nullptr, -1, nullptr,
                              nullptr, &w->block_w[blk][0], w44a));
            const float *shortcut = x;
            if (stride == 2) {
                RAFT_REQUIRE(cd.wp != nullptr, RAFT_E_NULL);
                RAFT_TRY(conv(ENC_1x1_S2, cd, x, C, Hc, Wc, Ho, Wo, 0, 0, F, EPI_LINEAR, b.rd, nullptr, nullptr, nullptr, -1,
                                  nullptr, nullptr));
                shortcut = b.rd;
            }
            RAFT_TRY(conv(ENC_3x3_S1, c2, b.r1, F, Ho, Wo, Ho, Wo, 1, 1, F, EPI_RES, y, shortcut, nullptr, nullptr, -1,
                              nullptr, nullptr, &w->block_w[blk][1], w44b));
        }
        float *t = x; x = y; y = t;
        C = F; Hc = Ho; Wc = Wo;
    }
    // ---- conv2 1x1 'valid' (extractor.py:127)
    return conv(ENC_1x1_S1, w->conv2, x, C, Hc, Wc, Hc, Wc, 0, 0, w->cout, EPI_LINEAR, out, nullptr, nullptr, nullptr, -1,
                nullptr, nullptr);
}
