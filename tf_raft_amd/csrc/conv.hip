// Convolution launchers, the two small special convolutions, and the BasicUpdateBlock sequencing
// (reference tf_raft/layers/update.py:5-153, tf_raft/model.py:84-109) for gfx950.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "conv_halo.h"
#include "conv_wino.h"
#include "conv_wino1d.h"
#include "conv_wino4.h"

// mask.2 + RAFT.upsample_flow in one kernel (mask_upsample.hip)
int raft_launch_mask_upsample(const float *a, int lda, const float *wp, const float *bias, int npad, const float *flow, int B,
                              int h, int w, float scale, float *out, hipStream_t s, int max_wgs = 0);
// RAFT_MASK_FUSED: the prediction loops run mask.2 and the convex upsampling as one kernel.  Default: from 2 pairs (2 x 3584
// feature pixels) on.  Launched one workgroup per tile the fused kernel only pays from 4 pairs (single pair 152.7 pairs/s with two
// kernels, 137.8 - 142.0 fused; two pairs 210.7 / 205.4; four 282.7 / 288.4: profiles/r08k_round3_options.txt, r07q); as the
// 32-workgroup background branch of the three-stream loop (struct Overlap) it pays from 2 pairs: 225.5 -> 244.1 pairs/s at two,
// 246.3 -> 256.3 at three (profiles/r09e_small_batch_mask.txt); a single pair stays on the two-kernel path (152 - 153 against
// 148 - 155).
static bool mask_is_fused(const raft_basic_update_weights *wts, int64_t pixels) {
    return raft_opt(RAFT_OPT_MASK_FUSED, pixels * raft_concurrency() >= 2 * 3584 ? 1 : 0) != 0 && wts->mask2.wp != nullptr && wts->mask2.npad == 576;
}

// ------------------------------------------------------------------------------------------------
// tile selection + dispatch of the direct (halo-tiled) convolution kernel
// ------------------------------------------------------------------------------------------------
// Tuning / test override: RAFT_CONV_TILE is either one code (applies to every convolution whose npad
// it divides) or a comma-separated list of `npad:taps:code` entries, e.g. "256:5:171,128:5:141".
// code = 100 + 10*TH + TN: TH x 16-pixel x 64*TN-channel workgroups of the halo-tiled kernel.
static bool code_valid(int code, int npad) {
    if (code >= 100) {
        const int th = (code - 100) / 10, tn = (code - 100) % 10;
        return (th == 4 || th == 7 || th == 8) && (tn == 1 || tn == 2) && npad % (64 * tn) == 0;
    }
    return false;
}
static int forced_code(int npad, int taps) { return raft_opt_conv_tile(npad, taps, code_valid); }

// Pick the halo tile (TH x 16 pixels, 64*TN channels) by a small cost model of MI355X (256 CUs):
//   time ~ (workgroups per CU, rounded up) x (MFMA work of one tile) x (latency-hiding penalty),
// where the penalty reflects how many workgroups (= waves per SIMD) are co-resident on a CU: one wave
// per SIMD exposes prologue / barrier / epilogue latency, three or more hide it (docs/NOTEBOOK.md section 4.1).
static int pick_code(const ConvArgs &a, int kh, int kw) {
    const int f = forced_code(a.npad, kh * kw);
    if (f >= 100) return f;
    static const int ths[3] = {4, 7, 8};
    double best = 1e30;
    int best_code = 141;
    for (int ti = 0; ti < 3; ++ti)
        for (int tn = 1; tn <= 2; ++tn) {
            const int th = ths[ti];
            if (a.npad % (64 * tn)) continue;
            const int64_t blocks = (int64_t)a.B * ((a.H + th - 1) / th) * ((a.W + 15) / 16) * (a.npad / (64 * tn)) * raft_concurrency();
            const int64_t per_cu = (blocks + 255) / 256;
            const int lds = 2 * ((th + kh - 1) * (16 + kw - 1) * 40 + 8) * 4;
            int resident = 160 * 1024 / lds;
            const int reg_limit = (tn == 2 && th >= 7) ? 1 : (th >= 7 ? 2 : 3);   // VGPR + AGPR budget per SIMD
            if (resident > reg_limit) resident = reg_limit;
            const int64_t conc = per_cu < resident ? per_cu : resident;
            const double pen = conc >= 3 ? 1.0 : (conc == 2 ? 1.05 : 1.15);
            const double eff = (th == 4 && kh * kw > 1) ? 0.88 : 1.0;   // short tiles: more halo traffic per MFMA (measured; a 1x1 kernel has no halo)
            const double cost = (double)per_cu * th * 16 * 64 * tn * pen / eff;
            if (cost < best) {
                best = cost;
                best_code = 100 + th * 10 + tn;
            }
        }
    return best_code;
}

int raft_launch_conv(const ConvArgs &a_in, int kh, int kw, int epi, hipStream_t s) {
    ConvArgs a = a_in;
    if (a.Hi == 0) {   // plain stride-1 'same' convolution
        a.Hi = a.H;
        a.Wi = a.W;
        a.pt = (kh - 1) / 2;
        a.pl = (kw - 1) / 2;
    }
    if (a.c0 <= 0 || a.c0 % 32 || a.c1 < 0 || a.c1 % 32 || a.npad <= 0 || a.npad % 64) return RAFT_E_UNSUPPORTED;
    if (a.lda0 % 4 || (a.c1 && a.lda1 % 4)) return RAFT_E_ALIGN;
    if (!raft_aligned16(a.a0) || !raft_aligned16(a.wp) || (a.c1 && !raft_aligned16(a.a1))) return RAFT_E_ALIGN;
    {   // every operand is addressed through 32-bit buffer offsets: each must span < 2 GiB
        const int64_t M = (int64_t)a.B * a.H * a.W;
        const int64_t lim = (int64_t)1 << 31;
        const int64_t e0 = ((M - 1) * a.lda0 + a.c0) * 4, e1 = a.c1 ? ((M - 1) * a.lda1 + a.c1) * 4 : 0;
        if (e0 >= lim || e1 >= lim) return RAFT_E_UNSUPPORTED;
        if (M * a.ldo0 * 4 >= lim || (a.o1 && M * a.ldo1 * 4 >= lim) || (a.e0 && M * a.lde0 * 4 >= lim) ||
            (a.e1 && M * a.lde1 * 4 >= lim))
            return RAFT_E_UNSUPPORTED;
        if ((int64_t)kh * kw * (a.c0 + a.c1) * a.npad * 4 >= lim) return RAFT_E_UNSUPPORTED;
    }
    const bool known = (kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 1 && kw == 5) || (kh == 5 && kw == 1);
    if (!known) return RAFT_E_UNSUPPORTED;
    const int code = pick_code(a, kh, kw);
    if (a.init) {
        const int64_t M = (int64_t)a.B * a.H * a.W;
        if (M * a.ldi * 4 >= ((int64_t)1 << 31)) return RAFT_E_UNSUPPORTED;
    }
    if (code >= 100) {
        const int th = (code - 100) / 10, tn = (code - 100) % 10;
        if (kh == 1 && kw == 1) return raft_launch_conv_halo_1x1(a, th, tn, epi, s);
        if (kh == 3 && kw == 3) return raft_launch_conv_halo_3x3(a, th, tn, epi, s);
        if (kh == 1 && kw == 5) return raft_launch_conv_halo_1x5(a, th, tn, epi, s);
        return raft_launch_conv_halo_5x1(a, th, tn, epi, s);
    }
    return RAFT_E_UNSUPPORTED;
}

extern "C" int raft_conv2d_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                               const float *wp, const float *bias, int B, int H, int W, int kh, int kw, int npad,
                               int nvalid, int act, float scale, float *out, int ldo, void *stream) {
    RAFT_REQUIRE_PTR(a0);
    RAFT_REQUIRE_PTR(wp);
    RAFT_REQUIRE_PTR(bias);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(c1 == 0 || a1 != nullptr, RAFT_E_NULL);
    RAFT_REQUIRE(B > 0 && H > 0 && W > 0 && nvalid > 0 && nvalid <= npad && ldo >= nvalid, RAFT_E_SHAPE);
    RAFT_REQUIRE(lda0 >= c0 && (c1 == 0 || lda1 >= c1), RAFT_E_SHAPE);
    RAFT_REQUIRE(act == RAFT_ACT_NONE || act == RAFT_ACT_RELU, RAFT_E_UNSUPPORTED);
    ConvArgs a = {};
    a.a0 = a0; a.a1 = a1; a.lda0 = lda0; a.lda1 = lda1; a.c0 = c0; a.c1 = c1;
    a.wp = wp; a.bias = bias; a.B = B; a.H = H; a.W = W;
    a.npad = npad; a.nvalid = nvalid; a.hid = 0; a.scale = scale;
    a.o0 = out; a.ldo0 = ldo;
    return raft_launch_conv(a, kh, kw, act == RAFT_ACT_RELU ? EPI_RELU : EPI_LINEAR, (hipStream_t)stream);
}

extern "C" int raft_conv2d_winograd_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                                        const float *wp, const float *bias, int B, int H, int W, int npad, int nvalid,
                                        int act, float scale, float *out, int ldo, void *stream) {
    RAFT_REQUIRE_PTR(a0);
    RAFT_REQUIRE_PTR(wp);
    RAFT_REQUIRE_PTR(bias);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(c1 == 0 || a1 != nullptr, RAFT_E_NULL);
    RAFT_REQUIRE(B > 0 && H > 0 && W > 0 && nvalid > 0 && nvalid <= npad && ldo >= nvalid, RAFT_E_SHAPE);
    RAFT_REQUIRE(lda0 >= c0 && (c1 == 0 || lda1 >= c1), RAFT_E_SHAPE);
    RAFT_REQUIRE(act == RAFT_ACT_NONE || act == RAFT_ACT_RELU, RAFT_E_UNSUPPORTED);
    ConvArgs a = {};
    a.a0 = a0; a.a1 = a1; a.lda0 = lda0; a.lda1 = lda1; a.c0 = c0; a.c1 = c1;
    a.wp = wp; a.bias = bias; a.B = B; a.H = H; a.W = W;
    a.npad = npad; a.nvalid = nvalid; a.hid = 0; a.scale = scale;
    a.o0 = out; a.ldo0 = ldo;
    return raft_launch_conv_wino(a, act == RAFT_ACT_RELU ? EPI_RELU : EPI_LINEAR, (hipStream_t)stream);
}

extern "C" int raft_conv2d_winograd4_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                                         const float *wp, const float *bias, int B, int H, int W, int npad, int nvalid,
                                         int act, float scale, float *out, int ldo, void *stream) {
    RAFT_REQUIRE_PTR(a0);
    RAFT_REQUIRE_PTR(wp);
    RAFT_REQUIRE_PTR(bias);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(c1 == 0 || a1 != nullptr, RAFT_E_NULL);
    RAFT_REQUIRE(B > 0 && H > 0 && W > 0 && nvalid > 0 && nvalid <= npad && ldo >= nvalid, RAFT_E_SHAPE);
    RAFT_REQUIRE(lda0 >= c0 && (c1 == 0 || lda1 >= c1), RAFT_E_SHAPE);
    RAFT_REQUIRE(act == RAFT_ACT_NONE || act == RAFT_ACT_RELU, RAFT_E_UNSUPPORTED);
    ConvArgs a = {};
    a.a0 = a0; a.a1 = a1; a.lda0 = lda0; a.lda1 = lda1; a.c0 = c0; a.c1 = c1;
    a.wp = wp; a.bias = bias; a.B = B; a.H = H; a.W = W;
    a.npad = npad; a.nvalid = nvalid; a.hid = 0; a.scale = scale;
    a.o0 = out; a.ldo0 = ldo;
    return raft_launch_conv_wino4(a, act == RAFT_ACT_RELU ? EPI_RELU : EPI_LINEAR, (hipStream_t)stream);
}

static int conv1d_winograd(int mo, const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                           const float *wp, const float *bias, int B, int H, int W, int kh, int kw,
                           int npad, int nvalid, int act, float scale, float *out, int ldo, void *stream) {
    RAFT_REQUIRE_PTR(a0);
    RAFT_REQUIRE_PTR(wp);
    RAFT_REQUIRE_PTR(bias);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(c1 == 0 || a1 != nullptr, RAFT_E_NULL);
    RAFT_REQUIRE(B > 0 && H > 0 && W > 0 && nvalid > 0 && nvalid <= npad && ldo >= nvalid, RAFT_E_SHAPE);
    RAFT_REQUIRE(lda0 >= c0 && (c1 == 0 || lda1 >= c1), RAFT_E_SHAPE);
    RAFT_REQUIRE(act == RAFT_ACT_NONE || act == RAFT_ACT_RELU, RAFT_E_UNSUPPORTED);
    ConvArgs a = {};
    a.a0 = a0; a.a1 = a1; a.lda0 = lda0; a.lda1 = lda1; a.c0 = c0; a.c1 = c1;
    a.wp = wp; a.bias = bias; a.B = B; a.H = H; a.W = W;
    a.npad = npad; a.nvalid = nvalid; a.hid = 0; a.scale = scale;
    a.o0 = out; a.ldo0 = ldo;
    return raft_launch_conv_wino1d(a, kh, kw, act == RAFT_ACT_RELU ? EPI_RELU : EPI_LINEAR, (hipStream_t)stream, mo);
}

extern "C" int raft_conv1d_winograd_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                                        const float *wp, const float *bias, int B, int H, int W, int kh, int kw,
                                        int npad, int nvalid, int act, float scale, float *out, int ldo, void *stream) {
    return conv1d_winograd(2, a0, lda0, c0, a1, lda1, c1, wp, bias, B, H, W, kh, kw, npad, nvalid, act, scale, out, ldo, stream);
}

extern "C" int raft_conv1d_winograd4_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                                         const float *wp, const float *bias, int B, int H, int W, int kh, int kw,
                                         int npad, int nvalid, int act, float scale, float *out, int ldo, void *stream) {
    return conv1d_winograd(4, a0, lda0, c0, a1, lda1, c1, wp, bias, B, H, W, kh, kw, npad, nvalid, act, scale, out, ldo, stream);
}

// The per-iteration SepConvGRU convolutions: direct halo kernel or 1-D Winograd F(2, 5) / F(4, 5) (conv_wino1d.h).
// RAFT_GRU_WINO and RAFT_GRU_WINO4 are bit masks over {1: gru_zr1, 2: gru_q1, 4: gru_zr2, 8: gru_q2}; a layer runs
// F(4, 5) if its WINO4 bit is set and the 8-tap weights were supplied, else F(2, 5) if its WINO bit is set and the
// 6-tap weights were supplied, else the direct kernel.  Unset = the defaults below.
constexpr int RAFT_GRU_WINO_DEFAULT = 15;
constexpr int RAFT_GRU_WINO4_DEFAULT = 15;
static int launch_gru_conv(const raft_conv_weights &direct, const raft_conv_weights &wino, const raft_conv_weights &wino4,
                           int bit, ConvArgs a, int kh, int kw, int epi, hipStream_t s) {
    const int mask = raft_opt(RAFT_OPT_GRU_WINO, RAFT_GRU_WINO_DEFAULT);
    // F(4, 5) wins where the launch fills the chip; below ~2 x 3584 pixels (the reference's single 448 x 512 pair) every
    // kernel is one under-filled round of workgroups and the F(2, 5) kernel's smaller workgroups finish sooner
    // (B = 1: 8.69 -> 8.40 ms per forward, profiles/r05d_b1_probe.txt)
    const int mask4 = raft_opt(RAFT_OPT_GRU_WINO4, (int64_t)a.B * a.H * a.W * raft_concurrency() < 2 * 3584 ? 0 : RAFT_GRU_WINO4_DEFAULT);
    if ((mask4 & bit) && wino4.wp != nullptr && a.c0 % 32 == 0 && a.c1 % 32 == 0) {
        a.wp = wino4.wp;
        a.bias = wino4.bias;
        a.npad = wino4.npad;
        return raft_launch_conv_wino1d(a, kh, kw, epi, s, 4, 0);   // workgroup width by grid size (a forced 32- / 64-channel width for gru_q lost to it: profiles/r09q_gru_q_tnw.txt)
    }
    if ((mask & bit) && wino.wp != nullptr) {
        a.wp = wino.wp;
        a.bias = wino.bias;
        a.npad = wino.npad;
        return raft_launch_conv_wino1d(a, kh, kw, epi, s);
    }
    a.wp = direct.wp;
    a.bias = direct.bias;
    a.npad = direct.npad;
    return raft_launch_conv(a, kh, kw, epi, s);
}

// The 3x3 layers of the update block run either on the direct halo kernel or on the Winograd F(2x2, 3x3) kernel
// (conv_wino.h) when the caller supplied transformed weights.  RAFT_CONV_WINO is a bit mask over
// {1: convc2, 2: convf2, 4: conv, 8: fh1_mask0}; unset = RAFT_WINO_DEFAULT (the layers where it measured faster at
// B = 4, docs/NOTEBOOK.md section 4.4).  Read per call so that tests can switch it.
constexpr int RAFT_WINO_DEFAULT = 13;
constexpr int RAFT_SMALL_WINO_DEFAULT = 15;   // SmallRAFT: {1: conv, 2: gru_zr, 4: gru_q, 8: fh1}, switch RAFT_SMALL_WINO
// F(4x4, 3x3) (conv_wino4.h), switch RAFT_CONV_WINO4 = bit mask {1: convc2, 4: conv, 8: fh1_mask0 / fh1}.  Default (us alone,
// F(4x4) against F(2x2), profiles/r07i_wino4_bench.txt): from 4 pairs on the flow / mask head (63 vs 91 at 4 pairs, 128 vs 169 at
// 8) and convc2 (61 vs 76 with the K-split workgroups, 107 vs 132); conv (N = 128) from 8 pairs on (65 vs 114; at 4 pairs its
// 112 K-split workgroups lose to F(2x2): 59 vs 49).  A single pair nothing: a launch is then one round of workgroups whose
// duration is one workgroup's K loop, and the one-wave-per-SIMD F(4x4) workgroup is the longer one (single pair: 151 pairs/s
// without, 133 with -- same-box A/B with bench.py, profiles/r07p_bench_mask_ab.txt: 4 pairs 270 -> 282, 8 pairs 285 -> 296);
// at two pairs the flow / mask head alone gains (206 -> 221 pairs/s with mask 8 on two boxes; with convc2 as well 214 and one
// outlier of 235: profiles/r08k_round3_options.txt, r08z_b2_options.txt).
// Bit 2 = convf2 (3x3, 128 -> 64), with convc2 from 3 pairs on: alone its 56 K-split workgroups (4 pairs) are slower than the direct kernel's
// 224 (42 against 27 us), but they take a quarter of the CU-time and, with 108 KB of LDS each, settle on CUs of their own: in the
// three-stream loop convc2's 168 K-split workgroups + these 56 + the 32 of the background mask branch are exactly 256 -- the flow
// branch no longer competes with convc2, which can have its faster shape back (one process, profiles/r09i_b4_options3.txt:
// 303.1 pairs/s -> 325.9 at 4 pairs; with convc2 on 8-row workgroups 303.3; 8 pairs 345.0 -> 353.3).
static int wino4_default_mask(const ConvArgs &a) {
    const int64_t m = (int64_t)a.B * a.H * a.W * raft_concurrency();   // loops sharing the chip fill it like one loop of n times the batch
    return m < 2 * 3584 ? 0 : (8 | (m >= 3 * 3584 ? 1 | 2 : 0) | (m >= 8 * 3584 ? 4 : 0));   // three pairs: 250 -> 262 pairs/s with 11, two: 237 -> 231
}
static int launch_conv3x3(const raft_conv_weights &direct, const raft_conv_weights &wino, int bit, ConvArgs a, int epi,
                          hipStream_t s, bool small = false, const raft_conv_weights *wino44 = nullptr, int w4_ks_hint = 0) {
    const int mask = small ? raft_opt(RAFT_OPT_SMALL_WINO, RAFT_SMALL_WINO_DEFAULT) : raft_opt(RAFT_OPT_CONV_WINO, RAFT_WINO_DEFAULT);
    if (wino44 != nullptr && wino44->wp != nullptr && (raft_opt(RAFT_OPT_CONV_WINO4, wino4_default_mask(a)) & bit) &&
        (epi == EPI_LINEAR || epi == EPI_RELU || epi == EPI_RES)) {
        a.wp = wino44->wp;
        a.bias = wino44->bias;
        a.npad = wino44->npad;
        return raft_launch_conv_wino4(a, epi, s, 0, w4_ks_hint);
    }
    if ((mask & bit) && wino.wp != nullptr) {
        a.wp = wino.wp;
        a.bias = wino.bias;
        a.npad = wino.npad;
        return raft_launch_conv_wino(a, epi, s);
    }
    a.wp = direct.wp;
    a.bias = direct.bias;
    a.npad = direct.npad;
    return raft_launch_conv(a, 3, 3, epi, s);
}

// ------------------------------------------------------------------------------------------------
// convf1: 7x7, Cin = 2 (flow), relu -- a GEMM [pixels x 98] . [98 x COUT] on fp32 MFMA 16x16x4.  [reference update.py:93]
// A workgroup owns 4 rows x 16 columns of pixels; wave w takes row w (one MFMA row block) and all COUT channels.  The
// (4 + 6) x (16 + 6) x 2 flow halo tile lives in LDS (zeros outside the image = 'same' padding); a k-step is four
// consecutive k = (ky * 7 + kx) * 2 + c of the Keras-layout kernel [k][n], so the A operand of lane (pixel r, k-group g)
// is the halo value at (row + ky, r + kx, c) and the B operand the kernel row k, both from LDS.
// K = 98 is padded to 100 with zero kernel rows.  The first version of this layer held the 98 weights of a lane's channel in
// registers and walked the pixels with wave-uniform LDS reads on the VALU: 18.9 us per launch at B = 4 for 0.36 GFLOP
// (profiles/r05b); this one is bound by its launch.
// ------------------------------------------------------------------------------------------------
template <int COUT>
__global__ void __launch_bounds__(256) conv7x7_c2_kernel(const float *__restrict__ flow, const float *__restrict__ wk,
                                                         const float *__restrict__ bias, int B, int H, int W,
                                                         float *__restrict__ out, int ldo) {
    constexpr int TH = 4, TW = 16, HW = TW + 6, NJ = COUT / 16, KQ = 25, LDW = COUT + 16;
    static_assert(COUT == 64 || COUT == 128, "conv7x7_c2: COUT must be 64 or 128");
    __shared__ float sf[(TH + 6) * HW * 2];
    // the whole kernel [k][n], rows 98 and 99 zero; row stride COUT + 16: the four k-groups of a B fragment read rows
    // k .. k + 3, 16 banks apart.  Staged with ONE round of coalesced loads per workgroup: fragments fetched from global
    // memory inside the k loop serialise on their L2 round trips (one wave per SIMD, nothing to hide them behind)
    __shared__ __attribute__((aligned(16))) float sw[4 * KQ * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int xt = (W + TW - 1) / TW, yt = (H + TH - 1) / TH;
    const int x0 = (blockIdx.x % xt) * TW, y0 = ((blockIdx.x / xt) % yt) * TH, b = blockIdx.x / (xt * yt);
    {   // every load of the staging phase is issued before the first LDS write (unconditional, clamped addresses): a
        // load / wait / store loop costs one L2 round trip per iteration -- 13 of them were most of this kernel's 16 us
        constexpr int NW = (4 * KQ * (COUT / 4) + 255) / 256;
        f32x4 tw[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + 256 * u, k = min(i / (COUT / 4), 97), c4 = i % (COUT / 4);
            tw[u] = *(const f32x4 *)(wk + k * COUT + c4 * 4);
        }
        const int i0 = tid, i1 = min(tid + 256, (TH + 6) * HW - 1);
        const int ya = y0 - 3 + i0 / HW, xa = x0 - 3 + i0 % HW, yb = y0 - 3 + i1 / HW, xb = x0 - 3 + i1 % HW;
        const bool oka = (unsigned)ya < (unsigned)H && (unsigned)xa < (unsigned)W;
        const bool okb = (unsigned)yb < (unsigned)H && (unsigned)xb < (unsigned)W;
        const float2 fa = ((const float2 *)flow)[oka ? ((int64_t)b * H + ya) * W + xa : 0];
        const float2 fb = ((const float2 *)flow)[okb ? ((int64_t)b * H + yb) * W + xb : 0];
        static_assert((TH + 6) * HW <= 512, "two halo items per thread");
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = tid + 256 * u, k = i / (COUT / 4), c4 = i % (COUT / 4);
            if (k < 4 * KQ) *(f32x4 *)(sw + k * LDW + c4 * 4) = k < 98 ? tw[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (i0 < (TH + 6) * HW) ((float2 *)sf)[i0] = oka ? fa : make_float2(0.f, 0.f);
        if (tid + 256 < (TH + 6) * HW) ((float2 *)sf)[tid + 256] = okb ? fb : make_float2(0.f, 0.f);
    }
    __syncthreads();
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
    for (int q = 0; q < KQ; ++q) {
        const int k = 4 * q + g;
        const int kk = k < 98 ? k : 0, ky = kk / 14, rem = kk - ky * 14;        // rem = kx * 2 + c; rows 98, 99 of sw are zero
        const float av = sf[((wv + ky) * HW + r) * 2 + rem];                   // ((row + ky) * HW + r + kx) * 2 + c
        float bv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = sw[k * LDW + j * 16 + r];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
    }
    // D[row = pixel 4 g + e of the wave's row][col = channel 16 j + r]
    const int y = y0 + wv;
    if (y >= H) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float bj = bias[j * 16 + r];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int x = x0 + 4 * g + e;
            if (x < W) out[(((int64_t)b * H + y) * W + x) * ldo + j * 16 + r] = fmaxf(acc[j][e] + bj, 0.f);
        }
    }
}

extern "C" int raft_conv7x7_c2_f32(const float *flow, const float *kernel, const float *bias, int cout, int B, int H, int W,
                                   float *out, int ldo, void *stream) {
    RAFT_REQUIRE_PTR(flow);
    RAFT_REQUIRE_PTR(kernel);
    RAFT_REQUIRE_PTR(bias);
    RAFT_REQUIRE_PTR(out);
    RAFT_REQUIRE(B > 0 && H > 0 && W > 0 && ldo >= cout, RAFT_E_SHAPE);
    RAFT_REQUIRE(raft_aligned16(kernel), RAFT_E_ALIGN);
    const int grid = B * ((H + 3) / 4) * ((W + 15) / 16);
    if (cout == 128)
        conv7x7_c2_kernel<128><<<grid, 256, 0, (hipStream_t)stream>>>(flow, kernel, bias, B, H, W, out, ldo);
    else if (cout == 64)
        conv7x7_c2_kernel<64><<<grid, 256, 0, (hipStream_t)stream>>>(flow, kernel, bias, B, H, W, out, ldo);
    else
        return RAFT_E_UNSUPPORTED;
    return raft_launch_status();
}

// ------------------------------------------------------------------------------------------------
// flow_head.conv2: 3x3, Cout = 2, fused with the coordinate update of the loop
//   delta = conv(x) + b; coords1 += delta; flow = coords1 - coords0    [update.py:14, model.py:97-102]
// One wavefront per 2 rows x 4 columns of pixels: lanes split the CIN channels (float4 / float2 per lane, the 9 x V x 2
// weights of the lane's channels in registers, one or two 16-byte loads per tap), the 4 x 6 input pixels of the eight
// windows are loaded once (24 independent loads in flight), giving 16 per-lane partial sums (8 pixels x 2 outputs).
// They are reduced across the 64 lanes by a transpose-reduction: four halving steps (xor 32, 16, 8, 4: each lane keeps
// half of its values and adds the partner's copies of them) leave one value per lane, two butterfly steps finish it.
// Lanes 0, 4, ..., 60 then own one (pixel, component) each and apply the coordinate update.
// (The first version served 1 x 4 pixels per wave and fetched its 72 weights with scalar loads: 22 load instructions
// per pixel against 5 here; 13.6 us per launch at B = 4 for a layer that reads 14.7 MB.)
// ------------------------------------------------------------------------------------------------
template <int CIN>
__global__ void __launch_bounds__(256) flowhead2_kernel(const float *__restrict__ x, int ldx,
                                                        const float *__restrict__ wk,   // [9][CIN][2]
                                                        const float *__restrict__ bias, int B, int H, int W,
                                                        float *__restrict__ delta, float *__restrict__ coords1,
                                                        float *__restrict__ flow, float *__restrict__ flow2,
                                                        int ldf2, float *__restrict__ flow3 = nullptr) {
    constexpr int V = CIN / 64;   // channels per lane (4 or 2)
    static_assert(V == 4 || V == 2, "flowhead2: CIN must be 256 or 128");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int ngx = (W + 3) / 4, ngy = (H + 1) / 2;
    const int64_t g = (int64_t)blockIdx.x * 4 + wid;          // group of 2 x 4 pixels
    if (g >= (int64_t)B * ngy * ngx) return;                   // wave-uniform
    const int gx = (int)(g % ngx), gy = (int)((g / ngx) % ngy), b = (int)(g / ((int64_t)ngx * ngy));
    const int x0 = gx * 4, y0 = gy * 2;
    // per-lane weights: 9 taps x V channels x 2 outputs, contiguous as [v][o] at (t * CIN + lane * V) * 2
    float w0[9][V], w1[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float *src = wk + ((int64_t)t * CIN + lane * V) * 2;
#pragma unroll
        for (int h = 0; h < V / 2; ++h) {
            const f32x4 q = *(const f32x4 *)(src + 4 * h);
            w0[t][2 * h] = q[0]; w1[t][2 * h] = q[1]; w0[t][2 * h + 1] = q[2]; w1[t][2 * h + 1] = q[3];
        }
    }
    // the 4 x 6 input pixels (zero outside the image; the conditions are wave-uniform)
    float in[4][6][V];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int yy = y0 + r - 1, xx = x0 + c - 1;
#pragma unroll
            for (int v = 0; v < V; ++v) in[r][c][v] = 0.f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float *src = x + (((int64_t)b * H + yy) * W + xx) * ldx + lane * V;
                if (V == 4) {
                    const f32x4 q = *(const f32x4 *)src;
                    in[r][c][0] = q[0]; in[r][c][1] = q[1]; in[r][c][2] = q[2]; in[r][c][3] = q[3];
                } else {
                    const float2 q = *(const float2 *)src;
                    in[r][c][0] = q.x; in[r][c][1] = q.y;
                }
            }
        }
    float acc[16];   // index = (row * 4 + pixel) * 2 + component
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float xv = in[r + t / 3][p + t % 3][v];
                    acc[(r * 4 + p) * 2] = fmaf(xv, w0[t][v], acc[(r * 4 + p) * 2]);
                    acc[(r * 4 + p) * 2 + 1] = fmaf(xv, w1[t][v], acc[(r * 4 + p) * 2 + 1]);
                }
    // transpose-reduction over the 64 lanes
    float a8[8], a4[4], a2[2], a1;
    {
        const bool hi = lane & 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float keep = hi ? acc[8 + i] : acc[i], send = hi ? acc[i] : acc[8 + i];
            a8[i] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool hi = lane & 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = hi ? a8[4 + i] : a8[i], send = hi ? a8[i] : a8[4 + i];
            a4[i] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool hi = lane & 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = hi ? a4[2 + i] : a4[i], send = hi ? a4[i] : a4[2 + i];
            a2[i] = keep + __shfl_xor(send, 8, 64);
        }
    }
    {
        const bool hi = lane & 4;
        const float keep = hi ? a2[1] : a2[0], send = hi ? a2[0] : a2[1];
        a1 = keep + __shfl_xor(send, 4, 64);
    }
    a1 += __shfl_xor(a1, 2, 64);
    a1 += __shfl_xor(a1, 1, 64);
    if ((lane & 3) == 0) {
        const int idx = lane >> 2;            // = (row * 4 + pixel) * 2 + component
        const int py = y0 + (idx >> 3), px = x0 + ((idx >> 1) & 3), comp = idx & 1;
        if (px < W && py < H) {
            const int64_t m = ((int64_t)b * H + py) * W + px;
            const float d = a1 + bias[comp];
            const float c = coords1[2 * m + comp] + d;
            coords1[2 * m + comp] = c;
            delta[2 * m + comp] = d;
            const float f = c - (float)(comp ? py : px);     // coords0 = (x, y) grid
            flow[2 * m + comp] = f;
            if (flow2) flow2[m * ldf2 + comp] = f;
            if (flow3) flow3[2 * m + comp] = f;       // the copy the (concurrent) mask branch of this iteration reads
        }
    }
}

// ------------------------------------------------------------------------------------------------
// state preparation  [model.py:84-89]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prepare_state_kernel(const float *__restrict__ cnet, int B, int h, int w,
                                                            int hdim, int cdim, float *__restrict__ net,
                                                            float *__restrict__ x, int ldx, int flow_slot,
                                                            float *__restrict__ corr, int ldc, int corr_used,
                                                            float *__restrict__ coords1, float *__restrict__ flow) {
    const int64_t M = (int64_t)B * h * w;
    const int per = hdim + cdim;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * per) return;
    const int64_t m = i / per;
    const int c = (int)(i - m * per);
    const float v = cnet[i];
    if (c < hdim)
        net[m * hdim + c] = tanhf(v);
    else
        x[m * ldx + (c - hdim)] = fmaxf(v, 0.f);
    if (c == 0) {
        const int px = (int)(m % w), py = (int)((m / w) % h);
        ((float2 *)coords1)[m] = make_float2((float)px, (float)py);
        ((float2 *)flow)[m] = make_float2(0.f, 0.f);
    }
    // GRU input tail [flow | zero pad]: the flow slot starts at 0 and is rewritten every iteration
    if (c < ldx - flow_slot) x[m * ldx + flow_slot + c] = 0.f;
    // zero pad channels of the lookup output (never written by the lookup, read by convc1)
    if (c < ldc - corr_used) corr[m * ldc + corr_used + c] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// BasicUpdateBlock
// workspace (floats per pixel): cor1 256 | corflo 256 [cor2 192 | flo2 64] | flo1 128 | z 128 | rh 128 | fm 512
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int WS_COR1 = 0, WS_CORFLO = 256, WS_FLO1 = 512, WS_Z = 640, WS_RH = 768, WS_FM = 896;
// second [flow_head.conv1 | mask.0] buffer and two copies of the flow for the three-stream loop with the fused mask + upsampling
// kernel: iteration i uses buffer i & 1, so the mask branch of iteration i - 1 is never overwritten by the main chain of i
constexpr int WS_FM2 = 1408, WS_FLOWM = 1920, WS_PER_PIX = 1924;
constexpr int HDIM = 128, XDIM = 256, CORR_LD = 352, CORR_USED = 324;
constexpr int CDIM = 128;               // inp channels = x[:, 0:CDIM]; x[:, CDIM:XDIM] = [motion 126 | flow 2]
constexpr int CTX_LD = 6 * HDIM;        // [z1 | r1 | q1 | z2 | r2 | q2] context terms per pixel
}   // namespace

extern "C" int64_t raft_update_workspace_floats(int B, int h, int w) {
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    return (int64_t)B * h * w * WS_PER_PIX;
}

static int check_state(const raft_state *st) {
    RAFT_REQUIRE_PTR(st);
    RAFT_REQUIRE_PTR(st->net);
    RAFT_REQUIRE_PTR(st->x);
    RAFT_REQUIRE_PTR(st->corr);
    RAFT_REQUIRE_PTR(st->coords1);
    RAFT_REQUIRE_PTR(st->flow);
    RAFT_REQUIRE_PTR(st->delta);
    RAFT_REQUIRE_PTR(st->mask);
    RAFT_REQUIRE_PTR(st->ws);
    RAFT_REQUIRE_PTR(st->ctx);
    return RAFT_OK;
}

extern "C" int raft_prepare_state_f32(const float *cnet, int B, int h, int w, const raft_state *st, void *stream) {
    RAFT_REQUIRE_PTR(cnet);
    int rc = check_state(st);
    if (rc) return rc;
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    const int64_t total = (int64_t)B * h * w * (HDIM + 128);
    prepare_state_kernel<<<raft_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
        cnet, B, h, w, HDIM, 128, st->net, st->x, XDIM, XDIM - 2, st->corr, CORR_LD, CORR_USED, st->coords1, st->flow);
    return raft_launch_status();
}

static ConvArgs conv_args(const raft_conv_weights &wt, const float *a0, int lda0, int c0, const float *a1, int lda1,
                          int c1, int B, int h, int w, int nvalid, float *o0, int ldo0) {
    ConvArgs a = {};
    a.a0 = a0; a.lda0 = lda0; a.c0 = c0; a.a1 = a1; a.lda1 = lda1; a.c1 = c1;
    a.wp = wt.wp; a.bias = wt.bias; a.npad = wt.npad; a.nvalid = nvalid;
    a.B = B; a.H = h; a.W = w; a.scale = 1.0f; a.o0 = o0; a.ldo0 = ldo0;
    return a;
}

// Loop-invariant part of the SepConvGRU.  hx = [h | inp | motion | flow] and [r*h | inp | motion | flow]
// (update.py:53, 58, 63): `inp` never changes inside the prediction loop (model.py:86, 91-106), so the
// inp rows of convz / convr / convq contribute the same pre-activation term in every iteration.  It is
// computed here once per forward -- one 1x5 and one 5x1 convolution 128 -> [z | r | q] (the biases ride
// along) -- and the per-iteration GRU convolutions start their accumulators from it and walk only the
// h / motion / flow rows (K = 5 * 256 instead of 5 * 384).
extern "C" int raft_gru_context_f32(const raft_basic_update_weights *wts, int B, int h, int w,
                                    const raft_state *st, void *stream) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_TRY(check_state(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    for (int pass = 0; pass < 2; ++pass) {
        const raft_conv_weights &wc = pass == 0 ? wts->gru_ctx1 : wts->gru_ctx2;
        const raft_conv_weights &w4 = pass == 0 ? wts->gru_ctx1_w4 : wts->gru_ctx2_w4;
        const int kh = pass == 0 ? 1 : 5, kw = pass == 0 ? 5 : 1;
        // F(4, 5) like the per-iteration GRU convolutions (same switch: bit 1 / 4 of RAFT_GRU_WINO4 = the pass)
        const bool wino = w4.wp != nullptr && (raft_opt(RAFT_OPT_GRU_WINO4, RAFT_GRU_WINO4_DEFAULT) & (pass == 0 ? 1 : 4));
        ConvArgs a = conv_args(wino ? w4 : wc, st->x, XDIM, CDIM, nullptr, 0, 0, B, h, w, 3 * HDIM, st->ctx + pass * 3 * HDIM, CTX_LD);
        if (wino)
            RAFT_TRY(raft_launch_conv_wino1d(a, kh, kw, EPI_LINEAR, (hipStream_t)stream, 4));
        else
            RAFT_TRY(raft_launch_conv(a, kh, kw, EPI_LINEAR, (hipStream_t)stream));
    }
    return RAFT_OK;
}

// Optional per-stage HIP-event recorder (profiling entry point only; see raft_iterate_basic_timed_f32).
struct StageTimer {
    hipEvent_t *ev;
    int n, cap;
    void mark(hipStream_t s) {
        if (n < cap) (void)hipEventRecord(ev[n++], s);
    }
};
#define RAFT_MARK()                 \
    do {                            \
        if (tm) tm->mark(s);        \
    } while (0)

// Where the loop's correlation features come from: the stored pyramid, or fmap1 + the pooled fmap2 pyramid (on demand).
struct LookupSource {
    const float *pyr;
    const int64_t *level_offsets;
    const float *fmap1, *fmap2_pyr;
    int C;
};
// Optional three-stream schedule of one loop iteration (raft_iterate_basic_overlap_f32):
//   main  lookup, convc1, convc2, [join flow branch] conv, GRU, [join previous upsample] fh1_mask0, fh2
//   s1    convf1, convf2                 (needs only the previous iteration's flow)
//   s2    mask2, upsample                (feed nothing inside the loop; must drain before the next fh1_mask0
//                                         overwrites their inputs)
// so the small / short / one-workgroup-per-CU kernels run in the shadows of the big ones.
struct Overlap {
    hipStream_t s1, s2;
    hipEvent_t e_fh, e_f, e_fm, e_up;   // after fh2, after convf2, after fh1_mask0, after upsample
    bool have_up;                       // e_up has been recorded (false in the first iteration)
    // Rotating buffers (all-predictions loop with the fused mask + upsampling kernel).  Every event operation on the MAIN
    // stream costs the dependent chain 6 - 11 us of idle time (the next kernel is not dispatched under the previous one's
    // tail: profiles/r07v_loop_gaps_b4.txt), and two of the four per iteration only protected buffers: the wait for the
    // previous mask branch before fh1_mask0 / fh2 overwrite what it reads, and the record that let mask.2 start before fh2.
    // With rot set, iteration i writes [fh1 | mask.0] and the mask branch's copy of the flow into buffer i & 1 and records
    // e_rot[i & 1] after its mask + upsampling kernel; the FLOW branch of iteration i + 2 waits for that event on its own
    // stream, and the main chain already waits for the flow branch before `conv` -- so the buffer is free before fh1_mask0
    // of i + 2 rewrites it, with no event operation added to the main stream (two per iteration are left: the flow-branch
    // join and the record after fh2).
    bool rot;
    int iter;
    hipEvent_t e_rot[2];
    // Background mask branch (rot only): every iteration but the last launches the mask + upsampling kernel with at most this
    // many workgroups (0 = one per tile).  The chain's kernels have 7 * 2^k workgroups at 448 x 512 and leave 32 CUs idle; 32
    // long-lived mask workgroups settle there (their 95 KB of LDS keep chain workgroups off those CUs) instead of competing
    // with the chain for all of them.  The last iteration's launch is a full one: nothing is left to hide behind.
    int mask_bg_wgs;
};
#define RAFT_HIP(expr)                       \
    do {                                     \
        hipError_t e__ = (expr);             \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// with_mask = false (final-only prediction, every iteration but the last): the mask branch -- mask.0 (the second half of
// fh1_mask0) and mask.2 -- is skipped; flow_head.conv1 alone runs from wts->fh1_w.
// The loops run the lookup fused into convc1 (raft_lookup_convc1_f32) when they read a stored volume, the repacked
// kernel was supplied and RAFT_LOOKUP_FUSED is not switched off.
static bool lookup_is_fused(const raft_basic_update_weights *wts, const LookupSource *src) {
    return src && src->pyr && wts->convc1_f.wp != nullptr && raft_opt(RAFT_OPT_LOOKUP_FUSED, 1) != 0;
}

// fused_src != NULL: st->corr is NOT read; cor1 comes from the volume through the fused kernel
static int update_basic_impl(const raft_basic_update_weights *wts, int B, int h, int w, const raft_state *st,
                             void *stream, StageTimer *tm, Overlap *ov = nullptr, bool with_mask = true,
                             const LookupSource *fused_src = nullptr, float *flow_up_fused = nullptr) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_TRY(check_state(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    hipStream_t sf = ov ? ov->s1 : s;   // flow branch
    hipStream_t sm = ov ? ov->s2 : s;   // mask branch
    const int64_t M = (int64_t)B * h * w;
    float *ws = st->ws;
    float *cor1 = ws + M * WS_COR1, *corflo = ws + M * WS_CORFLO, *flo1 = ws + M * WS_FLO1;
    const bool rot = ov && ov->rot && with_mask && flow_up_fused != nullptr;
    float *zb = ws + M * WS_Z, *rh = ws + M * WS_RH, *fm = ws + M * ((rot && (ov->iter & 1)) ? WS_FM2 : WS_FM);
    float *flowm = rot ? ws + M * WS_FLOWM + (ov->iter & 1) * 2 * M : nullptr;

    // ---- BasicMotionEncoder (update.py:97-106)
    if (fused_src) {   // cor = relu(convc1(retrieve(coords1)))   lookup + 1x1, 324 -> 256, one kernel
        RAFT_TRY(raft_lookup_convc1_f32(fused_src->pyr, fused_src->level_offsets, st->coords1, B, h, w, wts->convc1_f.wp,
                                        wts->convc1_f.bias, wts->convc1_f.npad, 256, cor1, 256, stream));
        RAFT_MARK();
    } else {   // cor = relu(convc1(corr))            1x1, 324(+28 zero pad) -> 256
        ConvArgs a = conv_args(wts->convc1, st->corr, CORR_LD, CORR_LD, nullptr, 0, 0, B, h, w, 256, cor1, 256);
        RAFT_TRY(raft_launch_conv(a, 1, 1, EPI_RELU, s));
        RAFT_MARK();
    }
    {   // cor = relu(convc2(cor))             3x3, 256 -> 192   -> corflo[:, 0:192]
        ConvArgs a = conv_args(wts->convc2, cor1, 256, 256, nullptr, 0, 0, B, h, w, 192, corflo, 256);
        // F(4x4) workgroup shape: the launcher's grid rule (K-split 4-row workgroups while 8-row ones would be fewer than 128: 168
        // instead of 84 at 4 pairs).  While the flow branch ran the direct convf2 (224 workgroups competing for the same CUs) the
        // 8-row shape was the better one in the loop (288.7 -> 293.7 pairs/s: less CU-time, room for the side branches,
        // profiles/r08n_b4_options.txt); with convf2 on its own 56 CUs (wino4_default_mask) the K-split shape wins by 7 %.
        // RAFT_CONVC2_KS = 1 / 2 forces either; every loop uses the same shape (the loops stay bit-identical to each other).
        RAFT_TRY(launch_conv3x3(wts->convc2, wts->convc2_w, 1, a, EPI_RELU, s, false, &wts->convc2_w44, raft_opt(RAFT_OPT_CONVC2_KS, 0)));
        RAFT_MARK();
    }
    if (ov) RAFT_HIP(hipStreamWaitEvent(sf, ov->e_fh, 0));   // flow of the previous iteration is final
    if (rot && ov->iter >= 2) RAFT_HIP(hipStreamWaitEvent(sf, ov->e_rot[ov->iter & 1], 0));   // mask branch of iteration - 2: its buffers are free
    {   // flo = relu(convf1(flow))            7x7, 2 -> 128
        conv7x7_c2_kernel<128><<<B * ((h + 3) / 4) * ((w + 15) / 16), 256, 0, sf>>>(st->flow, wts->convf1.wp, wts->convf1.bias, B, h, w, flo1, 128);
        RAFT_TRY(raft_launch_status());
        RAFT_MARK();
    }
    {   // flo = relu(convf2(flo))             3x3, 128 -> 64    -> corflo[:, 192:256]
        ConvArgs a = conv_args(wts->convf2, flo1, 128, 128, nullptr, 0, 0, B, h, w, 64, corflo + 192, 256);
        // F(4x4) shape: K-split workgroups up to 4 pairs (28 eight-row workgroups -> 56), eight-row ones from 56 on (8 pairs:
        // 56 of them beside convc2's 168 and the mask branch's 32: 353.4 -> 356.1 pairs/s, profiles/r09k_b8_options.txt)
        const int f2_grid1 = B * ((h + 7) / 8) * ((w + 63) / 64);
        RAFT_TRY(launch_conv3x3(wts->convf2, wts->convf2_w, 2, a, EPI_RELU, sf, false, &wts->convf2_w44,
                                raft_opt(RAFT_OPT_CONVF2_KS, f2_grid1 * raft_concurrency() >= 56 ? 1 : 0)));
        RAFT_MARK();
    }
    if (ov) {
        RAFT_HIP(hipEventRecord(ov->e_f, sf));
        RAFT_HIP(hipStreamWaitEvent(s, ov->e_f, 0));
    }
    {   // out = relu(conv(cat[cor, flo]))     3x3, 256 -> 126   -> x[:, 128:254]; x[:, 254:256] = flow (kept by flowhead2)
        ConvArgs a = conv_args(wts->conv, corflo, 256, 256, nullptr, 0, 0, B, h, w, 126, st->x + 128, XDIM);
        RAFT_TRY(launch_conv3x3(wts->conv, wts->conv_w, 4, a, EPI_RELU, s, false, &wts->conv_w44));
        RAFT_MARK();
    }
    // ---- SepConvGRU (update.py:51-67): hx = [h | x]; [r*h | x]
    for (int pass = 0; pass < 2; ++pass) {
        const raft_conv_weights &wzr = pass == 0 ? wts->gru_zr1 : wts->gru_zr2;
        const raft_conv_weights &wq = pass == 0 ? wts->gru_q1 : wts->gru_q2;
        const raft_conv_weights &wzr_w = pass == 0 ? wts->gru_zr1_w : wts->gru_zr2_w;
        const raft_conv_weights &wq_w = pass == 0 ? wts->gru_q1_w : wts->gru_q2_w;
        const raft_conv_weights &wzr_w4 = pass == 0 ? wts->gru_zr1_w4 : wts->gru_zr2_w4;
        const raft_conv_weights &wq_w4 = pass == 0 ? wts->gru_q1_w4 : wts->gru_q2_w4;
        const int kh = pass == 0 ? 1 : 5, kw = pass == 0 ? 5 : 1;
        const float *xm = st->x + CDIM;                      // [motion | flow]; the inp rows live in st->ctx
        const float *ctx = st->ctx + pass * 3 * HDIM;        // [z | r | q] context of this pass
        {
            ConvArgs a = conv_args(wzr, st->net, HDIM, HDIM, xm, XDIM, XDIM - CDIM, B, h, w, 2 * HDIM, zb, HDIM);
            a.hid = HDIM; a.o1 = rh; a.ldo1 = HDIM; a.e0 = st->net; a.lde0 = HDIM;
            a.init = ctx; a.ldi = CTX_LD;
            RAFT_TRY(launch_gru_conv(wzr, wzr_w, wzr_w4, pass == 0 ? 1 : 4, a, kh, kw, EPI_GRU_ZR, s));
            RAFT_MARK();
        }
        {
            ConvArgs a = conv_args(wq, rh, HDIM, HDIM, xm, XDIM, XDIM - CDIM, B, h, w, HDIM, st->net, HDIM);
            a.e0 = st->net; a.lde0 = HDIM; a.e1 = zb; a.lde1 = HDIM;
            a.init = ctx + 2 * HDIM; a.ldi = CTX_LD;
            RAFT_TRY(launch_gru_conv(wq, wq_w, wq_w4, pass == 0 ? 2 : 8, a, kh, kw, EPI_GRU_Q, s));
            RAFT_MARK();
        }
    }
    if (ov && ov->have_up && !rot) RAFT_HIP(hipStreamWaitEvent(s, ov->e_up, 0));   // mask2 / upsample of the previous iteration
    if (with_mask) {   // relu(flow_head.conv1(net)) | relu(mask.0(net))   3x3, 128 -> 256 + 256
        ConvArgs a = conv_args(wts->fh1_mask0, st->net, HDIM, HDIM, nullptr, 0, 0, B, h, w, 512, fm, 512);
        RAFT_TRY(launch_conv3x3(wts->fh1_mask0, wts->fh1_mask0_w, 8, a, EPI_RELU, s, false, &wts->fh1_mask0_w44));
        RAFT_MARK();
    } else {           // relu(flow_head.conv1(net)) only            3x3, 128 -> 256        -> fm[:, 0:256]
        const bool w44 = wts->fh1_w44.wp != nullptr && (raft_opt(RAFT_OPT_CONV_WINO4, (int64_t)B * h * w * raft_concurrency() < 2 * 3584 ? 0 : 8) & 8);
        ConvArgs a = conv_args(w44 ? wts->fh1_w44 : wts->fh1_w, st->net, HDIM, HDIM, nullptr, 0, 0, B, h, w, 256, fm, 512);
        RAFT_TRY(w44 ? raft_launch_conv_wino4(a, EPI_RELU, s, wts->fh1_mask0_w44.npad) : raft_launch_conv_wino(a, EPI_RELU, s, wts->fh1_mask0_w.npad));
    }
    if (ov && with_mask && flow_up_fused == nullptr) {   // two-kernel mask branch: mask.2 may start before fh2
        RAFT_HIP(hipEventRecord(ov->e_fm, s));
        RAFT_HIP(hipStreamWaitEvent(sm, ov->e_fm, 0));
    }
    {   // delta = flow_head.conv2(.), coords1 += delta, flow = coords1 - coords0
        flowhead2_kernel<256><<<raft_ceil_div((int64_t)B * ((h + 1) / 2) * ((w + 3) / 4), 4), 256, 0, s>>>(fm, 512, wts->fh2.wp, wts->fh2.bias, B, h, w,
                                                                   st->delta, st->coords1, st->flow, st->x + 254, XDIM, flowm);
        RAFT_TRY(raft_launch_status());
        RAFT_MARK();
    }
    if (ov) RAFT_HIP(hipEventRecord(ov->e_fh, s));
    if (with_mask && flow_up_fused != nullptr) {
        // mask.2 and the convex upsampling as ONE kernel (mask_upsample.hip): the mask is never written.  Besides fm (the mask
        // branch already waits for fh1_mask0) it needs the flow fh2 has just written.
        if (ov) RAFT_HIP(hipStreamWaitEvent(sm, ov->e_fh, 0));
        RAFT_TRY(raft_launch_mask_upsample(fm + 256, 512, wts->mask2.wp, wts->mask2.bias, wts->mask2.npad, rot ? flowm : st->flow, B, h, w,
                                           0.25f, flow_up_fused, sm, rot ? ov->mask_bg_wgs : 0));
        RAFT_MARK();
    } else if (with_mask) {   // mask = 0.25 * mask.2(.)             1x1, 256 -> 576
        ConvArgs a = conv_args(wts->mask2, fm + 256, 512, 256, nullptr, 0, 0, B, h, w, 576, st->mask, 576);
        a.scale = 0.25f;
        RAFT_TRY(raft_launch_conv(a, 1, 1, EPI_LINEAR, sm));
        RAFT_MARK();
    }
    return RAFT_OK;
}

extern "C" int raft_update_basic_f32(const raft_basic_update_weights *wts, int B, int h, int w,
                                     const raft_state *st, void *stream) {
    return update_basic_impl(wts, B, h, w, st, stream, nullptr);
}

extern "C" int raft_iterate_basic_f32(const raft_basic_update_weights *wts, const float *pyr,
                                      const int64_t *level_offsets, int B, int h, int w, int iters,
                                      const raft_state *st, float *flow_up, void *stream) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(flow_up);
    RAFT_TRY(check_state(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0 && iters > 0, RAFT_E_SHAPE);
    const int64_t up = (int64_t)B * 64 * h * w * 2;
    const LookupSource src = {pyr, level_offsets, nullptr, nullptr, 0};
    const bool fused = lookup_is_fused(wts, &src);
    for (int i = 0; i < iters; ++i) {
        if (!fused) RAFT_TRY(raft_corr_lookup_f32(pyr, level_offsets, st->coords1, B, h, w, 4, 4, st->corr, CORR_LD, stream));
        const bool mf = mask_is_fused(wts, (int64_t)B * h * w);
        RAFT_TRY(update_basic_impl(wts, B, h, w, st, stream, nullptr, nullptr, true, fused ? &src : nullptr, mf ? flow_up + i * up : nullptr));
        if (!mf) RAFT_TRY(raft_upsample_convex_f32(st->flow, st->mask, B, h, w, flow_up + i * up, stream));
    }
    return RAFT_OK;
}

static int loop_lookup(const LookupSource &src, const raft_state *st, int B, int h, int w, void *stream) {
    if (src.pyr) return raft_corr_lookup_f32(src.pyr, src.level_offsets, st->coords1, B, h, w, 4, 4, st->corr, CORR_LD, stream);
    return raft_corr_lookup_ondemand_f32(src.fmap1, src.fmap2_pyr, st->coords1, B, h, w, src.C, 4, 4, st->corr, CORR_LD, stream);
}

// Caller-owned loop context (include/raft_hip.h): the four cross-stream events of the three-stream schedule, created
// ONCE by raft_loop_ctx_create (the only allocating entry point), plus a small cache of instantiated hipGraphs of whole
// prediction loops keyed by every argument of the call.
struct LoopKey {
    raft_basic_update_weights wts;
    LookupSource src;
    raft_state st;
    int64_t offs[RAFT_MAX_LEVELS + 1];   // VALUES of level_offsets (the pointer is host memory of the caller)
    const float *flow_up;
    void *stream, *aux0, *aux1;
    int B, h, w, iters, final_only, opt_stamp;
};
struct LoopGraph {
    LoopKey key;
    hipGraphExec_t exec;
    uint64_t last_use;
};
struct raft_loop_ctx {
    hipEvent_t ev[4];
    LoopGraph graphs[4];
    int n_graphs;
    uint64_t clock;
    int device;
};

static int iterate_basic_overlap_impl(const raft_basic_update_weights *wts, const LookupSource &src, int B, int h, int w,
                                      int iters, const raft_state *st, float *flow_up, void *stream, void *aux0, void *aux1,
                                      raft_loop_ctx *ctx, bool final_only = false);

extern "C" int raft_loop_ctx_create(raft_loop_ctx **out) {
    RAFT_REQUIRE_PTR(out);
    raft_loop_ctx *c = (raft_loop_ctx *)calloc(1, sizeof(raft_loop_ctx));
    if (!c) return (int)hipErrorOutOfMemory;
    int rc = (int)hipGetDevice(&c->device);
    int made = 0;
    // Default HIP events (system-scope release / acquire when they complete).  The events only order streams of ONE device and
    // kernel boundaries release / acquire at agent scope anyway, so RAFT_EVENT_FENCE=0 creates them with
    // hipEventDisableSystemFence: +0.4 .. 0.9 % on the three-stream loop (329.8 against 326.3 - 327.0 pairs/s at 4 pairs, A/B/A in
    // one process, profiles/r10c_event_fence.txt), validated by the bitwise three-stream tests only -- since round 6 an opt-in:
    // the throughput schedule (several single-stream loops in flight) has no event inside the loop, so the default costs it nothing.
    // Read once, when the context is created.
    const unsigned flags = hipEventDisableTiming | (raft_opt(RAFT_OPT_EVENT_FENCE, 1) ? 0u : (unsigned)hipEventDisableSystemFence);
    for (; made < 4 && rc == RAFT_OK; ++made) rc = (int)hipEventCreateWithFlags(&c->ev[made], flags);
    if (rc != RAFT_OK) {
        for (int k = 0; k < made - 1; ++k) (void)hipEventDestroy(c->ev[k]);
        free(c);
        return rc;
    }
    *out = c;
    return RAFT_OK;
}

extern "C" int raft_loop_ctx_destroy(raft_loop_ctx *c) {
    if (!c) return RAFT_OK;
    for (int k = 0; k < c->n_graphs; ++k) (void)hipGraphExecDestroy(c->graphs[k].exec);
    for (int k = 0; k < 4; ++k) (void)hipEventDestroy(c->ev[k]);
    free(c);
    return RAFT_OK;
}

// raft_iterate_basic_f32 on three streams (see struct Overlap).  aux0 / aux1 are caller-owned streams
// distinct from `stream`; all work is joined back into `stream` before returning.
extern "C" int raft_iterate_basic_overlap_f32(const raft_basic_update_weights *wts, const float *pyr,
                                              const int64_t *level_offsets, int B, int h, int w, int iters,
                                              const raft_state *st, float *flow_up, void *stream, void *aux0,
                                              void *aux1, raft_loop_ctx *ctx) {
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    const LookupSource src = {pyr, level_offsets, nullptr, nullptr, 0};
    return iterate_basic_overlap_impl(wts, src, B, h, w, iters, st, flow_up, stream, aux0, aux1, ctx);
}

// The same three-stream loop with the volume-free ("alternate") correlation: every iteration's lookup computes its
// footprint correlations from fmap1 and the pooled fmap2 pyramid (raft_fmap_pyramid_f32).  BASELINE config 4.
extern "C" int raft_iterate_basic_ondemand_f32(const raft_basic_update_weights *wts, const float *fmap1,
                                               const float *fmap2_pyr, int C, int B, int h, int w, int iters,
                                               const raft_state *st, float *flow_up, void *stream, void *aux0,
                                               void *aux1, raft_loop_ctx *ctx) {
    RAFT_REQUIRE_PTR(fmap1);
    RAFT_REQUIRE_PTR(fmap2_pyr);
    const LookupSource src = {nullptr, nullptr, fmap1, fmap2_pyr, C};
    return iterate_basic_overlap_impl(wts, src, B, h, w, iters, st, flow_up, stream, aux0, aux1, ctx);
}

// The prediction loop for callers that only want flow_predictions[-1] (reference model.py:160-166, predict_step): the
// mask head and the convex upsampling run in the LAST iteration only; flow_up_last: (B, 8h, 8w, 2).  The recurrence
// (lookup, motion encoder, GRU, flow head) is launch for launch the one of raft_iterate_basic_overlap_f32, so the
// result equals its last prediction.  Needs the Winograd copy of flow_head.conv1 (wts->fh1_w).
extern "C" int raft_iterate_basic_final_f32(const raft_basic_update_weights *wts, const float *pyr,
                                            const int64_t *level_offsets, int B, int h, int w, int iters,
                                            const raft_state *st, float *flow_up_last, void *stream, void *aux0,
                                            void *aux1, raft_loop_ctx *ctx) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE(wts->fh1_w.wp != nullptr, RAFT_E_NULL);
    const LookupSource src = {pyr, level_offsets, nullptr, nullptr, 0};
    return iterate_basic_overlap_impl(wts, src, B, h, w, iters, st, flow_up_last, stream, aux0, aux1, ctx, true);
}

// Enqueue the whole loop on `stream` + the two side streams (also the body of a stream capture).
static int enqueue_loop(const raft_basic_update_weights *wts, const LookupSource &src, int B, int h, int w, int iters,
                        const raft_state *st, float *flow_up, void *stream, void *aux0, void *aux1, raft_loop_ctx *ctx,
                        bool final_only) {
    hipStream_t s = (hipStream_t)stream;
    if (aux0 == stream) {
        // single-stream schedule (aux0 == aux1 == stream): the launches of raft_iterate_basic_f32, for every lookup source and
        // for the final-only loop -- what several concurrent loops (one per lane of the pipelined forward) run
        const int64_t up1 = (int64_t)B * 64 * h * w * 2;
        for (int i = 0; i < iters; ++i) {
            const bool with_mask = !final_only || i == iters - 1;
            const bool fused = lookup_is_fused(wts, &src);
            if (!fused) RAFT_TRY(loop_lookup(src, st, B, h, w, stream));
            float *up_i = flow_up + (final_only ? 0 : i * up1);
            const bool mf = with_mask && mask_is_fused(wts, (int64_t)B * h * w);
            RAFT_TRY(update_basic_impl(wts, B, h, w, st, stream, nullptr, nullptr, with_mask, fused ? &src : nullptr, mf ? up_i : nullptr));
            if (with_mask && !mf) RAFT_TRY(raft_upsample_convex_f32(st->flow, st->mask, B, h, w, up_i, stream));
        }
        return RAFT_OK;
    }
    Overlap ov = {};
    ov.s1 = (hipStream_t)aux0;
    ov.s2 = (hipStream_t)aux1;
    ov.e_fh = ctx->ev[0];
    ov.e_f = ctx->ev[1];
    ov.e_fm = ctx->ev[2];
    ov.e_up = ctx->ev[3];
    ov.e_rot[0] = ctx->ev[2];   // e_fm is not used in that mode
    ov.e_rot[1] = ctx->ev[3];
    ov.rot = !final_only && mask_is_fused(wts, (int64_t)B * h * w);
    const int64_t up = (int64_t)B * 64 * h * w * 2;
    int rc = (int)hipEventRecord(ov.e_fh, s);   // state prepared on `stream`: the flow branch may start
    for (int i = 0; i < iters && rc == RAFT_OK; ++i) {
        const bool with_mask = !final_only || i == iters - 1;
        const bool fused = lookup_is_fused(wts, &src);
        rc = fused ? RAFT_OK : loop_lookup(src, st, B, h, w, stream);
        float *up_i = flow_up + (final_only ? 0 : i * up);
        const bool mf = with_mask && mask_is_fused(wts, (int64_t)B * h * w);
        ov.iter = i;
        // default 32, except where the chain's launches cover the chip exactly (the flow / mask head's F(4x4) grid a multiple of
        // 256: a single 1024 x 1024 pair loses 6 % to a background branch); one process, profiles/r09d_mask_bg_shapes.txt:
        // 448 x 512 at 4 / 5 / 6 / 8 / 12 / 16 pairs +2.7 / +7.9 / +4.5 / +6.1 / +2.5 / +1.6 %, 16 or 40+ workgroups lose
        const int head_grid = B * ((h + 7) / 8) * ((w + 63) / 64) * 8;
        ov.mask_bg_wgs = (ov.rot && i + 1 < iters) ? (head_grid % 256 ? 32 : 0) : 0;
        if (rc == RAFT_OK) rc = update_basic_impl(wts, B, h, w, st, stream, nullptr, &ov, with_mask, fused ? &src : nullptr, mf ? up_i : nullptr);
        if (!with_mask) continue;
        if (!mf) {
            // upsample on the mask branch: needs mask2 (same stream) and the flow written by fh2
            if (rc == RAFT_OK) rc = (int)hipStreamWaitEvent(ov.s2, ov.e_fh, 0);
            if (rc == RAFT_OK) rc = raft_upsample_convex_f32(st->flow, st->mask, B, h, w, up_i, ov.s2);
        }
        if (ov.rot) ov.e_up = ov.e_rot[i & 1];
        if (rc == RAFT_OK) rc = (int)hipEventRecord(ov.e_up, ov.s2);
        ov.have_up = true;
    }
    if (rc == RAFT_OK && ov.have_up) rc = (int)hipStreamWaitEvent(s, ov.e_up, 0);   // join
    return rc;
}

// Padding-free copy of the weight table (it is an array of raft_conv_weights): the key is compared with memcmp.
static void copy_weights_clean(raft_basic_update_weights *dst, const raft_basic_update_weights *src) {
    static_assert(sizeof(raft_basic_update_weights) % sizeof(raft_conv_weights) == 0, "weight table = array of raft_conv_weights");
    memset(dst, 0, sizeof(*dst));
    const raft_conv_weights *sw = (const raft_conv_weights *)src;
    raft_conv_weights *dw = (raft_conv_weights *)dst;
    for (size_t i = 0; i < sizeof(*src) / sizeof(raft_conv_weights); ++i) {
        dw[i].wp = sw[i].wp;
        dw[i].bias = sw[i].bias;
        dw[i].npad = sw[i].npad;
    }
}

// RAFT_LOOP_GRAPH (raft_set_option): 1 = replay the loop as ONE hipGraph launch (captured from the same enqueue code the
// first time a given set of arguments is seen, then cached in the caller's raft_loop_ctx), 0 = enqueue the ~350 kernels
// and ~100 event operations from the host every call.  Default 0: measured on MI355X / ROCm 7.2 the replay is SLOWER
// than the stream launches at every batch size (B = 1: 8.44 vs 7.94 ms, B = 4: 15.9 vs 15.4 ms,
// profiles/r05a_batch_sweep.txt) -- the host is not the limiter of the small-batch loop, the dependent chain of short
// kernels on the GPU is (docs/NOTEBOOK.md section 4.2).  Kept as a switch: bit-identical results, one launch per forward.
static bool use_loop_graph(int, int, int) { return raft_opt(RAFT_OPT_LOOP_GRAPH, 0) != 0; }

static int iterate_basic_overlap_impl(const raft_basic_update_weights *wts, const LookupSource &src, int B, int h, int w,
                                      int iters, const raft_state *st, float *flow_up, void *stream, void *aux0, void *aux1,
                                      raft_loop_ctx *ctx, bool final_only) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_REQUIRE_PTR(flow_up);
    RAFT_REQUIRE_PTR(aux0);
    RAFT_REQUIRE_PTR(aux1);
    RAFT_REQUIRE_PTR(ctx);
    RAFT_TRY(check_state(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0 && iters > 0, RAFT_E_SHAPE);
    // three distinct streams, or all three the same one (the single-stream schedule)
    RAFT_REQUIRE((aux0 != stream && aux1 != stream && aux0 != aux1) || (aux0 == stream && aux1 == stream), RAFT_E_UNSUPPORTED);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (!use_loop_graph(B, h, w) || s == nullptr) {   // the legacy NULL stream cannot be captured
        rc = enqueue_loop(wts, src, B, h, w, iters, st, flow_up, stream, aux0, aux1, ctx, final_only);
        if (rc != RAFT_OK) {   // never leave side streams running behind an error return
            (void)hipStreamSynchronize((hipStream_t)aux0);
            (void)hipStreamSynchronize((hipStream_t)aux1);
        }
        return rc;
    }
    LoopKey key;
    memset(&key, 0, sizeof(key));   // padding bytes take part in the memcmp below
    copy_weights_clean(&key.wts, wts);
    key.src.pyr = src.pyr;
    key.src.fmap1 = src.fmap1;
    key.src.fmap2_pyr = src.fmap2_pyr;
    key.src.C = src.C;
    if (src.level_offsets)
        for (int l = 0; l <= RAFT_MAX_LEVELS; ++l) key.offs[l] = src.level_offsets[l];
    key.st = *st;   // nine pointers, no padding
    key.flow_up = flow_up;
    key.stream = stream;
    key.aux0 = aux0;
    key.aux1 = aux1;
    key.B = B; key.h = h; key.w = w; key.iters = iters;
    key.final_only = final_only ? 1 : 0;
    key.opt_stamp = raft_opt_generation();   // any raft_set_option call may change which kernels the loop launches
    ++ctx->clock;
    for (int k = 0; k < ctx->n_graphs; ++k)
        if (memcmp(&ctx->graphs[k].key, &key, sizeof(key)) == 0) {
            ctx->graphs[k].last_use = ctx->clock;
            return (int)hipGraphLaunch(ctx->graphs[k].exec, s);
        }
    // capture the enqueue code: the event record / wait pairs pull the two side streams into the capture
    rc = (int)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (rc != RAFT_OK) return rc;
    rc = enqueue_loop(wts, src, B, h, w, iters, st, flow_up, stream, aux0, aux1, ctx, final_only);
    hipGraph_t graph = nullptr;
    const int rc_end = (int)hipStreamEndCapture(s, &graph);
    if (rc == RAFT_OK) rc = rc_end;
    hipGraphExec_t exec = nullptr;
    if (rc == RAFT_OK) rc = (int)hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc != RAFT_OK) return rc;
    int slot = ctx->n_graphs;
    if (slot == 4) {   // evict the least recently used graph
        slot = 0;
        for (int k = 1; k < 4; ++k)
            if (ctx->graphs[k].last_use < ctx->graphs[slot].last_use) slot = k;
        (void)hipGraphExecDestroy(ctx->graphs[slot].exec);
    } else {
        ++ctx->n_graphs;
    }
    ctx->graphs[slot].key = key;
    ctx->graphs[slot].exec = exec;
    ctx->graphs[slot].last_use = ctx->clock;
    return (int)hipGraphLaunch(exec, s);
}

// Profiling twin of raft_iterate_basic_f32: identical launches, plus a HIP event after every kernel
// on `stream`; synchronises and accumulates per-stage milliseconds into stage_ms[RAFT_BASIC_STAGES]
// (host array).  Used by bench.py for the live roofline numbers -- never on the product path.
extern "C" int raft_iterate_basic_timed_f32(const raft_basic_update_weights *wts, const float *pyr,
                                            const int64_t *level_offsets, int B, int h, int w, int iters,
                                            const raft_state *st, float *flow_up, void *stream,
                                            float *stage_ms) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(flow_up);
    RAFT_REQUIRE_PTR(stage_ms);
    RAFT_TRY(check_state(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0 && iters > 0 && iters <= 64, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    const int per_iter = RAFT_BASIC_STAGES;
    const int nev = iters * per_iter + 1;
    hipEvent_t *ev = (hipEvent_t *)malloc(sizeof(hipEvent_t) * nev);
    if (!ev) return (int)hipErrorOutOfMemory;
    for (int i = 0; i < nev; ++i) (void)hipEventCreate(&ev[i]);
    StageTimer tm = {ev, 0, nev};
    const int64_t up = (int64_t)B * 64 * h * w * 2;
    int rc = RAFT_OK;
    const LookupSource src = {pyr, level_offsets, nullptr, nullptr, 0};
    const bool fused = lookup_is_fused(wts, &src);
    tm.mark(s);
    for (int i = 0; i < iters && rc == RAFT_OK; ++i) {
        // RAFT_LOOKUP_FUSED as in the product loops: fused, the lookup stage is empty and the convc1 stage is the fused kernel
        if (!fused) rc = raft_corr_lookup_f32(pyr, level_offsets, st->coords1, B, h, w, 4, 4, st->corr, CORR_LD, stream);
        tm.mark(s);
        // RAFT_MASK_FUSED likewise: fused, the mask2 stage is the fused kernel and the upsampling stage is empty
        const bool mf = mask_is_fused(wts, (int64_t)B * h * w);
        if (rc == RAFT_OK) rc = update_basic_impl(wts, B, h, w, st, stream, &tm, nullptr, true, fused ? &src : nullptr, mf ? flow_up + i * up : nullptr);
        if (rc == RAFT_OK && !mf) rc = raft_upsample_convex_f32(st->flow, st->mask, B, h, w, flow_up + i * up, stream);
        tm.mark(s);
    }
    if (rc == RAFT_OK) rc = (int)hipStreamSynchronize(s);
    if (rc == RAFT_OK && tm.n == nev) {
        for (int k = 0; k < per_iter; ++k) stage_ms[k] = 0.f;
        for (int i = 0; i < iters; ++i)
            for (int k = 0; k < per_iter; ++k) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, ev[i * per_iter + k], ev[i * per_iter + k + 1]);
                stage_ms[k] += ms;
            }
    }
    for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
    free(ev);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// SmallUpdateBlock  (reference update.py:70-85, 17-35, 109-125; model.py:190-226)
//   net (M,96); x (M,160) = [inp 64 | motion 80 | flow 2 | 14 zero pad]; corr (M,224) = 196 + 28 pad
// workspace (floats per pixel): corflo 128 [cor 96 | flo2 32] | flo1 64 | z 96 | rh 96 | fh 128
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int SW_CORFLO = 0, SW_FLO1 = 128, SW_Z = 192, SW_RH = 288, SW_FH = 384, SW_PER_PIX = 512;
constexpr int S_HDIM = 96, S_CDIM = 64, S_XLD = 160, S_FLOW_SLOT = 144, S_CORR_LD = 224, S_CORR_USED = 196;
}   // namespace

extern "C" int64_t raft_small_update_workspace_floats(int B, int h, int w) {
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    return (int64_t)B * h * w * SW_PER_PIX;
}

static int check_state_small(const raft_state *st) {
    RAFT_REQUIRE_PTR(st);
    RAFT_REQUIRE_PTR(st->net);
    RAFT_REQUIRE_PTR(st->x);
    RAFT_REQUIRE_PTR(st->corr);
    RAFT_REQUIRE_PTR(st->coords1);
    RAFT_REQUIRE_PTR(st->flow);
    RAFT_REQUIRE_PTR(st->delta);
    RAFT_REQUIRE_PTR(st->ws);
    return RAFT_OK;
}

extern "C" int raft_prepare_state_small_f32(const float *cnet, int B, int h, int w, const raft_state *st,
                                            void *stream) {
    RAFT_REQUIRE_PTR(cnet);
    RAFT_TRY(check_state_small(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    const int64_t total = (int64_t)B * h * w * (S_HDIM + S_CDIM);
    prepare_state_kernel<<<raft_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
        cnet, B, h, w, S_HDIM, S_CDIM, st->net, st->x, S_XLD, S_FLOW_SLOT, st->corr, S_CORR_LD, S_CORR_USED,
        st->coords1, st->flow);
    return raft_launch_status();
}

extern "C" int raft_update_small_f32(const raft_small_update_weights *wts, int B, int h, int w,
                                     const raft_state *st, void *stream) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_TRY(check_state_small(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0, RAFT_E_SHAPE);
    hipStream_t s = (hipStream_t)stream;
    const int64_t M = (int64_t)B * h * w;
    float *ws = st->ws;
    float *corflo = ws + M * SW_CORFLO, *flo1 = ws + M * SW_FLO1, *zb = ws + M * SW_Z, *rh = ws + M * SW_RH;
    float *fh = ws + M * SW_FH;
    {   // cor = relu(convc1(corr))      1x1, 196(+28) -> 96     -> corflo[:, 0:96]
        ConvArgs a = conv_args(wts->convc1, st->corr, S_CORR_LD, S_CORR_LD, nullptr, 0, 0, B, h, w, 96, corflo, 128);
        RAFT_TRY(raft_launch_conv(a, 1, 1, EPI_RELU, s));
    }
    {   // flo = relu(convf1(flow))      7x7, 2 -> 64
        conv7x7_c2_kernel<64><<<B * ((h + 3) / 4) * ((w + 15) / 16), 256, 0, s>>>(st->flow, wts->convf1.wp, wts->convf1.bias, B, h, w, flo1, 64);
        RAFT_TRY(raft_launch_status());
    }
    {   // flo = relu(convf2(flo))       3x3, 64 -> 32           -> corflo[:, 96:128]
        ConvArgs a = conv_args(wts->convf2, flo1, 64, 64, nullptr, 0, 0, B, h, w, 32, corflo + 96, 128);
        RAFT_TRY(raft_launch_conv(a, 3, 3, EPI_RELU, s));
    }
    {   // out = relu(conv(cat[cor, flo])) 3x3, 128 -> 80        -> x[:, 64:144]
        ConvArgs a = conv_args(wts->conv, corflo, 128, 128, nullptr, 0, 0, B, h, w, 80, st->x + 64, S_XLD);
        RAFT_TRY(launch_conv3x3(wts->conv, wts->conv_w, 1, a, EPI_RELU, s, true));
    }
    {   // ConvGRU (update.py:26-35), 3x3: z | r
        ConvArgs a = conv_args(wts->gru_zr, st->net, S_HDIM, S_HDIM, st->x, S_XLD, S_XLD, B, h, w, 2 * S_HDIM, zb,
                               S_HDIM);
        a.hid = S_HDIM; a.o1 = rh; a.ldo1 = S_HDIM; a.e0 = st->net; a.lde0 = S_HDIM;
        RAFT_TRY(launch_conv3x3(wts->gru_zr, wts->gru_zr_w, 2, a, EPI_GRU_ZR, s, true));
    }
    {
        ConvArgs a = conv_args(wts->gru_q, rh, S_HDIM, S_HDIM, st->x, S_XLD, S_XLD, B, h, w, S_HDIM, st->net, S_HDIM);
        a.e0 = st->net; a.lde0 = S_HDIM; a.e1 = zb; a.lde1 = S_HDIM;
        RAFT_TRY(launch_conv3x3(wts->gru_q, wts->gru_q_w, 4, a, EPI_GRU_Q, s, true));
    }
    {   // relu(flow_head.conv1(net))    3x3, 96 -> 128
        ConvArgs a = conv_args(wts->fh1, st->net, S_HDIM, S_HDIM, nullptr, 0, 0, B, h, w, 128, fh, 128);
        RAFT_TRY(launch_conv3x3(wts->fh1, wts->fh1_w, 8, a, EPI_RELU, s, true));
    }
    {   // delta = flow_head.conv2(.), coords1 += delta, flow = coords1 - coords0
        flowhead2_kernel<128><<<raft_ceil_div((int64_t)B * ((h + 1) / 2) * ((w + 3) / 4), 4), 256, 0, s>>>(fh, 128, wts->fh2.wp, wts->fh2.bias, B, h, w,
                                                                   st->delta, st->coords1, st->flow,
                                                                   st->x + S_FLOW_SLOT, S_XLD);
        RAFT_TRY(raft_launch_status());
    }
    return RAFT_OK;
}

extern "C" int raft_iterate_small_f32(const raft_small_update_weights *wts, const float *pyr,
                                      const int64_t *level_offsets, int B, int h, int w, int iters,
                                      const raft_state *st, float *flow_up, void *stream) {
    RAFT_REQUIRE_PTR(wts);
    RAFT_REQUIRE_PTR(pyr);
    RAFT_REQUIRE_PTR(level_offsets);
    RAFT_REQUIRE_PTR(flow_up);
    RAFT_TRY(check_state_small(st));
    RAFT_REQUIRE(B > 0 && h > 0 && w > 0 && iters > 0, RAFT_E_SHAPE);
    const int64_t up = (int64_t)B * 64 * h * w * 2;
    for (int i = 0; i < iters; ++i) {
        RAFT_TRY(raft_corr_lookup_f32(pyr, level_offsets, st->coords1, B, h, w, 4, 3, st->corr, S_CORR_LD, stream));
        RAFT_TRY(raft_update_small_f32(wts, B, h, w, st, stream));
        RAFT_TRY(raft_upflow8_f32(st->flow, B, h, w, flow_up + i * up, stream));
    }
    return RAFT_OK;
}

// ------------------------------------------------------------------------------------------------
// misc
// ------------------------------------------------------------------------------------------------
extern "C" int raft_version(void) { return RAFT_HIP_VERSION; }

extern "C" const char *raft_error_string(int rc) {
    switch (rc) {
        case RAFT_OK: return "ok";
        case RAFT_E_NULL: return "raft: required pointer is NULL";
        case RAFT_E_SHAPE: return "raft: invalid or inconsistent dimension";
        case RAFT_E_UNSUPPORTED: return "raft: configuration not instantiated (radius / channels / kernel size)";
        case RAFT_E_ALIGN: return "raft: pointer or leading dimension not 16-byte aligned";
    }
    if (rc > 0) return hipGetErrorString((hipError_t)rc);
    return "raft: unknown error";
}
