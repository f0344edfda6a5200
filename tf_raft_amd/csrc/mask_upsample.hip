// mask.2 (1x1 convolution 256 -> 576, x 0.25) fused with RAFT.upsample_flow for gfx950   [reference update.py:137-141, 152 and
// model.py:39-66]: the (B, h, w, 576) mask never reaches HBM -- 33 MB written by the convolution and 33 MB read back by the
// upsampling per iteration at 4 pairs, and one kernel boundary, are gone.
//
// The two halves are the two kernels they replace, instruction for instruction where it decides the value:
//   * GEMM [64 pixels x 256] . [256 x 576] on fp32 MFMA 16x16x4 with the K order of conv_halo_kernel<1, 1, 4, ...> (32-channel
//     chunks, per chunk two rounds kk, lane group G takes k-quad 4 kk + G, four MFMAs per quad), weights in the packed layout of
//     the direct kernels straight from L2, v = (acc + bias) * 0.25: the logits are bit-identical to mask2's;
//   * softmax over the 9 taps of every sub-pixel and the blend of the 3x3 neighbourhood of 8 * flow exactly as
//     upsample_convex_kernel evaluates them (max chain, exp2((v - max) * log2 e), one reciprocal, products in tap order).
// A workgroup owns a 4 x 16 tile of coarse pixels and all 576 channels: 8 waves, wave = (two of the four tile rows, 9 of the 36
// column blocks) = 18 accumulator tiles; 144 channels are exactly 16 sub-pixels x 9 taps.  After the K loop the logits of half
// the tile (32 pixels x 576, pixel stride 592 floats: conflict-free stride-9 reads) go to LDS, all 512 threads turn them into
// 32 x 64 output pixels (lanes = consecutive sub-columns: 64-byte runs per store), then the other half.
#include "conv_mfma.h"

namespace {
struct MaskUpArgs {
    const float *a;       // relu(mask.0(net)): (M, lda) floats, 256 channels used
    int lda;
    const float *wp;      // mask.2 packed (1, 64, 576, 4)
    const float *bias;    // 576
    const float *flow;    // (M, 2) low-resolution flow, already updated by the flow head
    float *out;           // (B, 8h, 8w, 2)
    int B, h, w;
    float scale;          // 0.25
};

constexpr int MU_TH = 4, MU_TW = 16, MU_LDA = 40, MU_ABUF = 64 * MU_LDA, MU_PXS = 592, MU_N = 576, MU_K = 256;

__global__ void __launch_bounds__(512) mask_upsample_kernel(MaskUpArgs p) {
    __shared__ __attribute__((aligned(16))) float sa[2 * MU_ABUF];       // A tile, double-buffered (20 KB)
    __shared__ __attribute__((aligned(16))) float sm[32 * MU_PXS];       // logits of half a tile (74 KB)
    __shared__ float2 sf[6 * 18];                                        // 8 * flow, rows y0-1..y0+4, columns x0-1..x0+16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = lane >> 4, LR = lane & 15;
    const int rbp = wv & 1, cg = wv >> 1;                                // tile rows {2 rbp, 2 rbp + 1}; column blocks 9 cg .. 9 cg + 8
    const int tiles_x = (p.w + MU_TW - 1) / MU_TW, tiles_y = (p.h + MU_TH - 1) / MU_TH;
    const int ntiles = p.B * tiles_x * tiles_y;
    // A launch may have FEWER workgroups than tiles (background mode, see raft_launch_mask_upsample): workgroup g walks the tiles
    // g, g + gridDim.x, ...
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;   // XCD-aware remap (see conv_halo.h)
    {
        const int nwg = ntiles, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = bid / (tiles_x * tiles_y);
    const int y0 = ty * MU_TH, x0 = tx * MU_TW;
    const int M = p.B * p.h * p.w;

    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.a, 0, (int)((((long)M - 1) * p.lda + MU_K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)p.wp, 0, MU_K * MU_N * 4, 0x00020000);

    // ---- A staging: one 16-byte item per thread per chunk: (tile pixel tid >> 3, channel quad tid & 7)
    const int spx = tid >> 3, sc4 = tid & 7;
    const int syy = y0 + (spx >> 4), sxx = x0 + (spx & 15);
    const bool sok = (syy < p.h) & (sxx < p.w);
    const unsigned soff = sok ? (unsigned)((((b * p.h + syy) * p.w + sxx) * p.lda + sc4 * 4) * 4) : RAFT_OOB;
    f32x4 ra;
    auto gload = [&](int c) { ra = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsa, (int)soff, c * 128, 0)); };
    auto lstore = [&](int buf) { *(f32x4 *)(sa + buf * MU_ABUF + spx * MU_LDA + sc4 * 4) = ra; };
    // ---- fragments
    const int a_lane = ((2 * rbp) * 16 + LR) * MU_LDA + G * 4;
    f32x4 fa[2][2], fb[2][9];
    auto frag_a = [&](int buf, int kk, f32x4 *f) {
#pragma unroll
        for (int i = 0; i < 2; ++i) f[i] = *(const f32x4 *)(sa + buf * MU_ABUF + a_lane + i * 16 * MU_LDA + kk * 16);
    };
    const unsigned b_lane = (unsigned)(((G * MU_N) + cg * 144 + LR) * 16);   // bytes
    auto frag_b = [&](int c, int kk, f32x4 *f) {
        const unsigned row = (unsigned)((c * 8 + kk * 4) * MU_N) * 16u;      // wave-uniform bytes
#pragma unroll
        for (int j = 0; j < 9; ++j)
            f[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(b_lane + j * 256), (int)row, 0));
    };
    f32x4 acc[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- K loop: 8 chunks of 32 channels, two rounds each; fragments one round ahead, next tile staged under the current one
    constexpr int NCH = MU_K / 32;
    gload(0);
    frag_b(0, 0, fb[0]);
    if (tid < 108) {   // 8 * flow of the tile's 6 x 18 neighbourhood (zero outside the image: 'SAME' patches, model.py:55)
        const int yy = y0 - 1 + tid / 18, xx = x0 - 1 + tid % 18;
        float2 f = make_float2(0.f, 0.f);
        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
            f = *(const float2 *)(p.flow + ((int64_t)(b * p.h + yy) * p.w + xx) * 2);
            f.x *= 8.f;
            f.y *= 8.f;
        }
        sf[tid] = f;
    }
    lstore(0);
    raft_barrier_lds();
    gload(1);
    frag_a(0, 0, fa[0]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int cur = kk;
            if (kk == 0) {
                frag_a(buf, 1, fa[1]);
                frag_b(c, 1, fb[1]);
            } else if (c + 1 < NCH) {
                // the next chunk's tile: written to the other buffer and made visible before its first fragment is read
                lstore(buf ^ 1);
                if (c + 2 < NCH) gload(c + 2);
                raft_barrier_lds();
                frag_a(buf ^ 1, 0, fa[0]);
                frag_b(c + 1, 0, fb[0]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 9; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][r], fb[cur][j][r], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: half a tile (two rows = 32 pixels) at a time through LDS
    float bias[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) bias[j] = p.bias[cg * 144 + j * 16 + LR];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        raft_barrier_lds();                                   // the previous half's logits have been consumed
        if (rbp == half) {
            // lane owns channel n = 144 cg + 16 j + LR; register r of accumulator (i, j) is pixel (row 2 half + i, column 4 G + r)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 9; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sm[(i * 16 + 4 * G + r) * MU_PXS + cg * 144 + j * 16 + LR] = (acc[i][j][r] + bias[j]) * p.scale;
        }
        raft_barrier_lds();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int item = tid + 512 * t;
            const int px = item >> 6, sub = item & 63;       // pixel of the half tile, sub-pixel i * 8 + j
            const int row = 2 * half + (px >> 4), col = px & 15;
            const int y = y0 + row, x = x0 + col;
            if (y >= p.h || x >= p.w) continue;
            const float *tl = sm + px * MU_PXS + sub * 9;
            float v[9], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                v[k] = tl[k];
                mx = fmaxf(mx, v[k]);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                v[k] = __builtin_amdgcn_exp2f((v[k] - mx) * 1.44269504088896341f);
                s += v[k];
            }
            const float inv = 1.0f / s;
            float ox = 0.f, oy = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float2 nb = sf[(row + k / 3) * 18 + col + k % 3];   // patch depth order (ky, kx, ch)
                const float wk = v[k] * inv;
                ox += wk * nb.x;
                oy += wk * nb.y;
            }
            float2 *o = (float2 *)p.out + ((int64_t)b * (8 * p.h) + (8 * y + (sub >> 3))) * (int64_t)(8 * p.w) + (8 * x + (sub & 7));
            *o = make_float2(ox, oy);
        }
    }
    raft_barrier_lds();   // the next tile's staging rewrites sa / sf, its epilogue sm
  }
}
}   // namespace

// mask = 0.25 * mask.2(a) (1x1, 256 -> 576) and flow_up = RAFT.upsample_flow(flow, mask) in one kernel; `a` = (B*h*w, lda) with
// the 256 input channels first, `wp` / `bias` = mask.2 packed for the direct kernels (npad 576).  Internal (conv.hip loops).
// max_wgs > 0: BACKGROUND mode -- at most that many workgroups, each walking several tiles.  At 448 x 512 every kernel of the
// dependent chain has 7 * 2^k workgroups (56 feature rows), i.e. it leaves 32 of the 256 CUs idle; a mask branch of 32 long-lived
// workgroups (95 KB of LDS each: no chain workgroup fits beside one) settles on 32 CUs and the chain takes the other 224.
int raft_launch_mask_upsample(const float *a, int lda, const float *wp, const float *bias, int npad, const float *flow, int B,
                              int h, int w, float scale, float *out, hipStream_t s, int max_wgs) {
    if (a == nullptr || wp == nullptr || bias == nullptr || flow == nullptr || out == nullptr) return RAFT_E_NULL;
    if (B <= 0 || h <= 0 || w <= 0) return RAFT_E_SHAPE;
    if (npad != MU_N || lda < MU_K || lda % 4) return RAFT_E_UNSUPPORTED;
    if (!raft_aligned16(a) || !raft_aligned16(wp)) return RAFT_E_ALIGN;
    if ((int64_t)B * h * w * lda * 4 >= ((int64_t)1 << 31)) return RAFT_E_UNSUPPORTED;
    MaskUpArgs p = {a, lda, wp, bias, flow, out, B, h, w, scale};
    int grid = B * ((h + MU_TH - 1) / MU_TH) * ((w + MU_TW - 1) / MU_TW);
    if (max_wgs > 0 && grid > max_wgs) grid = max_wgs;
    mask_upsample_kernel<<<grid, 512, 0, s>>>(p);
    return raft_launch_status();
}
