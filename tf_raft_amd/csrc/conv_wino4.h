// Winograd F(4x4, 3x3) stride-1 'same' convolution on fp32 MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A with the interpolation points {0, +-5/8, +-3/2, inf} (tap order [0, +a, -a, +b, -b,
// inf]; matrices and the reason for these points: tf_raft_amd/packing.py, tools/study/wino_error.py).  A 4x4 output tile
// costs 36 multiplies per (input channel, output channel) instead of 144: 4x fewer MACs than the direct kernel and 1.78x
// fewer than F(2x2, 3x3) (conv_wino.h).  fp32 throughout; U = G g G^T is evaluated on the host in float64 and rounded once.
// Measured deviation from the float64 convolution: ~3x that of F(2x2, 3x3), ~5x that of the direct fp32 kernel
// (tests/test_gpu_kernels.py reports all three); the F(2x2) and direct kernels stay selectable (RAFT_CONV_WINO4 = 0).
//
// Work decomposition (why it differs from conv_wino.h): 36 independent tap-GEMMs keep 36 accumulator tiles per (row block,
// column block) alive until the output transform -- 144 registers per 16 tiles x 16 channels.  Two column blocks per wave
// (so that one transformed input feeds two MFMAs) are 288 accumulator registers: ONE wave per SIMD with the 512-register
// budget, accumulators in AGPRs.  Nothing hides a stall for a lone wave, so the K loop is a hand-placed software pipeline:
//
//   * MFMA row = one 4x4 output tile, row block = 16 tiles along x (4 rows x 64 columns).  Workgroup = 4 waves = 2 row
//     blocks (8 x 64 pixels) x 2 pairs of column blocks (64 channels); wave w: row block w & 1, channel pair w >> 1.
//   * K is walked in 16-channel chunks; the 10 x 66 halo tile of a chunk sits in LDS (double-buffered, ONE barrier per
//     chunk).  Pixel x of a halo row is at x * 16 + (x >> 2) * 4 floats and its 16 channels are stored as [half][quad][2]:
//     lane (tile LR, k-quad G) reads channels (4G + 2h, 4G + 2h + 1) of its patch as ds_read_b64 at lane stride 68 LR +
//     2 G floats -- conflict-free (tools/bank_check.py wino4).  A chunk is processed as two halves h (k-steps 2h, 2h + 1 of
//     every tap) so that the transform works on 2 channels per lane: W and V are 12 registers a row instead of 24.
//   * A half is three PHASES of two tap rows each -- (+a, -a), (+b, -b), (0, inf): rows of a pair share their even / odd
//     parts -- and a phase is 12 tap SLOTS of four MFMAs (2 k-steps x 2 column blocks): 72 slots per chunk.  Patch rows 1..4 of
//     a half are read ONCE into 48 registers (every phase needs them: under the last phase of the previous half), rows 0 / 5
//     when the (0, inf) rows are formed -- 72 ds_read_b64 per chunk.  Stage 1 (B^T over the patch rows -> W[2][2][6]) of phase
//     k + 1 is spread over the slots of phase k (LDS reads in even slots, arithmetic in odd slots); stage 2 (W -> V) runs one
//     tap ahead; the weight fragments arrive RAFT_WINO4_PF (7) slots ahead, straight from L2: the stream is packed in
//     consumption order, so a slot's fragments for both column blocks are one 16-byte load at ONE running scalar offset.
//     The next chunk's halo tile is fetched item by item in slots 0..NA-1 (scalar channel offset: no address arithmetic) and
//     written to the other LDS buffer RAFT_WINO4_LAG (12) slots later; the chunk's barrier sits in front of its last phase,
//     whose stage-1 reads already belong to the next chunk.  Two chunks per loop trip (one per LDS buffer: every LDS access is
//     base register + immediate); an odd chunk count runs one ghost chunk on zeros.
//   * 72 accumulator tiles are 288 registers: MFMAs are inline asm, taps 0..30 accumulate in AGPRs, 31..35 in VGPRs (hipcc
//     picks one register class for all builtin MFMAs of a function and shuffles the overflow through copies otherwise).
//   * epilogue: A^T . A over the 36 accumulators of a column block (lane-local), bias, relu / residual, stores with one lane
//     base per tensor + wave-uniform element offsets (as conv_wino.h).
#pragma once
#include "conv_mfma.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#include <type_traits>
#include <utility>

// two channels of a lane in the transforms.  Default: a 2-vector (hipcc emits v_pk_fma_f32); RAFT_WINO4_SCALAR (tools/ablate,
// together with -fno-slp-vectorize): a pair of scalars (v_fma_f32), to price packed against scalar VALU beside the MFMAs
#ifdef RAFT_WINO4_SCALAR
struct alignas(8) w4v2 {
    float x, y;
    __device__ __forceinline__ float operator[](int i) const { return i ? y : x; }
};
__device__ __forceinline__ w4v2 operator+(w4v2 a, w4v2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ w4v2 operator-(w4v2 a, w4v2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ w4v2 operator*(float k, w4v2 a) { return {k * a.x, k * a.y}; }
#else
typedef f32x2 w4v2;
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  The accumulator
// of a tap is selected in the FRONT END (if constexpr over named variables): one array of 72 tiles is never promoted to
// registers by hipcc (too many uses of one alloca), and a switch over 72 cases x 288 call sites is not unrolled.
template <class F, int... I>
__device__ __forceinline__ void w4_static_for_impl(F &f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void w4_static_for(F &&f) {
    w4_static_for_impl(f, std::make_integer_sequence<int, N>{});
}
#define W4_TAPS(X)                                                                                                       \
    X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21)   \
    X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35)

#ifndef RAFT_WINO4_PF
#define RAFT_WINO4_PF 7      // weight fragments are fetched this many tap slots (4 MFMAs each) ahead (in the loop the weights are
                             // cold: 4 slots measured 70 us for convc2 / fh1_mask0 at 4 pairs, 6 - 7 slots 61 - 64, profiles/r07k)
#endif
#ifdef RAFT_WINO4_NOSB        // tools/ablate: leave the order of the slot bodies to the compiler
#define W4_SB()
#else
#define W4_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef RAFT_WINO4_NOSNOP      // tools/ablate: MFMA statements without their leading wait states (timing only: may miscompute)
#define W4_PAD ""
#else
#define W4_PAD "s_nop 1\n\t"
#endif
// the four MFMAs of a tap slot: %0 / %1 = the tap's accumulator tiles (column block 0 / 1), %2 %3 = V (k-steps 2h, 2h + 1),
// %4 %5 = weights of column block 0, %6 %7 = of column block 1
#define W4_MFMA4                                                                                                          \
    W4_PAD "v_mfma_f32_16x16x4_f32 %0, %2, %4, %0\n\tv_mfma_f32_16x16x4_f32 %1, %2, %6, %1\n\t"                          \
           "v_mfma_f32_16x16x4_f32 %0, %3, %5, %0\n\tv_mfma_f32_16x16x4_f32 %1, %3, %7, %1"
#ifndef RAFT_WINO4_ATAPS
#define RAFT_WINO4_ATAPS 31   // taps (of 36; two accumulator tiles each) whose accumulators are kept in AGPRs
#endif
#ifndef RAFT_WINO4_ABL
#define RAFT_WINO4_ABL 0     // tools/ablate: 1 no weight loads in the loop, 2 no stage 1, 4 no halo staging, 8 no stores, 16 no stage 2
#endif

namespace wino4 {
constexpr float PA = 0.625f, PB = 1.5f;
constexpr float A2 = PA * PA, B2 = PB * PB, SS = A2 + B2, PP = A2 * B2, A3 = A2 * PA, B3 = B2 * PB;
constexpr int TW = 64, HWP = TW + 2;            // 16 tiles of 4 columns; halo row of 66 pixels
constexpr int RS = 1124;                       // floats per halo row: 66 * 16 + 17 * 4
constexpr int NTHR = 256;
constexpr int PF = RAFT_WINO4_PF, NR = 8;
constexpr int W4_ATAPS = RAFT_WINO4_ATAPS;
#ifndef RAFT_WINO4_LAG
#define RAFT_WINO4_LAG 12
#endif
constexpr int LOAD_SLOT0 = 0, STORE_LAG = RAFT_WINO4_LAG;   // halo item i: global load in slot LOAD_SLOT0 + i, LDS write STORE_LAG later
static_assert(PF >= 1 && PF < NR, "prefetch distance must fit the fragment ring");
// tap row of (phase, row-in-phase)
__device__ __forceinline__ constexpr int tap_row(int ph, int row) { return ph == 0 ? 1 + row : (ph == 1 ? 3 + row : 5 * row); }
}   // namespace wino4

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-lambda-capture"
// KS = 1: workgroup = 2 row blocks (8 x 64 pixels) x 64 channels, wave = (row block, pair of column blocks).
// KS = 2: workgroup = 1 row block (4 x 64 pixels) x 64 channels, wave = (pair of column blocks, K half): wave set ks walks the
// 16-channel chunks ks, ks + 2, ... (a stage = two chunks, staged side by side) and the two partial results meet in LDS after the
// output transform -- for layers whose 2-row-block workgroups are too few to cover the chip (N <= 256 at 4 pairs).
// PRE / STATS serve the instance-norm encoder as in conv_wino.h (KS = 1, one source): PRE applies relu(x * scale[b][c] + shift[b][c])
// to every in-image element while the halo tile is written to LDS (the normalisation + relu of the producer: the normalised
// tensor is never materialised; padding stays zero), STATS writes per-(workgroup, row block) (sum, sum of squares) of the raw
// output per channel to p.stats[2 * pixel tile + rb][npad][2].
template <int EPI, int KS = 1, int PRE = 0, int STATS = 0>
__global__ void __launch_bounds__(256, 1) conv_wino4_kernel(ConvArgs p) {
    using namespace wino4;
    static_assert(EPI == EPI_LINEAR || EPI == EPI_RELU || EPI == EPI_RES, "winograd F(4x4) kernel: linear / relu / residual epilogues");
    static_assert(KS == 1 || KS == 2, "K split: 1 or 2");
    static_assert((!PRE && !STATS) || KS == 1, "encoder variants: two row blocks per workgroup");
    constexpr int TH = KS == 1 ? 8 : 4, HH = TH + 2, HP = HH * HWP;   // halo rows x 66 pixels
    constexpr int A_BUF = HH * RS + 16;                               // + one dummy pixel for the padding items of the staging loop
    constexpr int NA = (HP * 4 * KS + NTHR - 1) / NTHR;               // 16-byte items per thread per stage (11 / 13)
    static_assert(LOAD_SLOT0 + NA - 1 + STORE_LAG < 60, "halo writes must precede the chunk's barrier (slot 60)");
    __shared__ __attribute__((aligned(16))) float smem[2 * KS * A_BUF];   // [stage buffer][K half]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it (and what derives from it) scalar
    const int G = lane >> 4, LR = lane & 15;
    const int rb = KS == 1 ? (w & 1) : 0, cbp = KS == 1 ? (w >> 1) : (w & 1), ks = KS == 1 ? 0 : (w >> 1);
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntn = p.npad / 64;
    const int M = p.B * p.H * p.W;

    int bid = blockIdx.x;   // XCD-aware remap (see conv_halo.h)
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int mt = bid / ntn, nt = bid - mt * ntn;
#ifdef RAFT_WINO4_XCD
    // experiment: one channel group (or a set of them) per XCD, so that an XCD's L2 holds 1/8 of the transformed weights and
    // every pixel tile of that group runs on it (hardware places workgroup b on XCD b % 8)
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        if ((nwg & 7) == 0 && (ntn & 7) == 0) {
            const int per = ntn >> 3;
            nt = xcd + 8 * (idx % per);
            mt = idx / per;
        } else if ((nwg & 7) == 0 && 8 % ntn == 0) {
            const int g = 8 / ntn;
            nt = xcd % ntn;
            mt = idx * g + xcd / ntn;
        }
    }
#endif
    const int tx0 = mt % tiles_x, ty0 = (mt / tiles_x) % tiles_y, b = mt / (tiles_x * tiles_y);
    const int y0 = ty0 * TH, x0 = tx0 * TW;
    const int n0 = nt * 64;
    const int cin = p.c0 + p.c1;
    const int nch = cin >> 4;        // 16-channel chunks
    const int nst = nch / KS;        // stages (KS = 2: nch is even and so is c0 / 16 -- the launcher checks)

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.a0, 0, (int)((((long)M - 1) * p.lda0 + p.c0) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.wp, 0, (int)((long)36 * cin * p.npad * 4), 0x00020000);

    // ---- halo staging: item = (halo pixel, 16-byte channel quad of the chunk).  The byte offset of an item inside its source
    // tensor is kept per thread (out-of-image items: an offset beyond any extent, the bounds check returns 0); the chunk's
    // channel offset goes into the instruction's scalar offset, so a load is one instruction and no address arithmetic.
    unsigned pixoff[NA];
    int lds_off[NA], pixi[NA];
    unsigned okbits = 0;                                               // PRE: bit i = item i lies inside the image
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int item = tid + NTHR * i;
        const int hq = item >> 2, c4 = item & 3;
        const int s2 = hq / HP, hp = hq - s2 * HP;                 // chunk of the stage (KS = 2: 0 / 1), halo pixel
        const int hy = hp / HWP, hx = hp - hy * HWP;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        const bool ok = (s2 < KS) & ((unsigned)yy < (unsigned)p.H) & ((unsigned)xx < (unsigned)p.W);
        pixi[i] = ok ? (b * p.H + yy) * p.W + xx : -1;
        okbits |= ok ? 1u << i : 0u;
        pixoff[i] = ok ? (unsigned)((pixi[i] * p.lda0 + s2 * 16 + c4 * 4) * 4) : RAFT_OOB;
        lds_off[i] = s2 < KS ? s2 * A_BUF + hy * RS + hx * 16 + (hx >> 2) * 4 + 2 * c4 : HH * RS;   // floats; second half at + 8
    }
    f32x4 ra[NA];
    f32x2 pre_sc[2], pre_sh[2];                                        // PRE: scale / shift of the NEXT chunk's channel quad (tid & 3)
    // source of a chunk (channels [0, c0) come from a0, the rest from a1): descriptor and scalar channel offset of the NEXT
    // chunk are set once per chunk; at the switch from a0 to a1 the per-thread offsets are re-derived with a1's pixel stride
    __amdgpu_buffer_rsrc_t rsn = rs0;
    int soff_n = 0;
    bool on_first = true;
    auto next_source = [&](int st) {   // st = stage index (KS chunks of 16 channels each, all from one source)
        const int ch = st * 16 * KS;
        const bool first = ch < p.c0, live = st < nst && !(RAFT_WINO4_ABL & 4);
        const float *base = first ? p.a0 : (p.c1 ? p.a1 : p.a0);
        const int ext = first ? (int)((((long)M - 1) * p.lda0 + p.c0) * 4) : (p.c1 ? (int)((((long)M - 1) * p.lda1 + p.c1) * 4) : 0);
        rsn = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, live ? ext : 0, 0x00020000);   // past the last chunk: empty
        soff_n = (first ? ch : ch - p.c0) * 4;
        if constexpr (PRE) {        // one source (the launcher checks): channel ch + 4 (tid & 3) of image b; past the last chunk: unused
            const int cq = (st < nst ? ch : 0) + (tid & 3) * 4;
            const f32x4 sc = *(const f32x4 *)(p.pre_scale + (long)b * cin + cq), sh = *(const f32x4 *)(p.pre_shift + (long)b * cin + cq);
            pre_sc[0] = f32x2{sc[0], sc[1]};
            pre_sc[1] = f32x2{sc[2], sc[3]};
            pre_sh[0] = f32x2{sh[0], sh[1]};
            pre_sh[1] = f32x2{sh[2], sh[3]};
            return;
        }
        if (on_first && !first) {   // wave-uniform, taken once
            on_first = false;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                pixoff[i] = pixi[i] >= 0 ? (unsigned)((pixi[i] * p.lda1 + (((tid + NTHR * i) >> 2) / HP) * 16 + (tid & 3) * 4) * 4) : RAFT_OOB;
        }
    };
    auto gload_item = [&](int i) {
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsn, (int)pixoff[i], soff_n, 0));
    };
    auto lstore_item = [&](int i, int buf) {
        float *dst = smem + buf * (KS * A_BUF) + lds_off[i];
        f32x2 lo = f32x2{ra[i][0], ra[i][1]}, hi = f32x2{ra[i][2], ra[i][3]};
        if constexpr (PRE) {
            const bool ok = (okbits >> i) & 1u;
            const f32x2 zero = f32x2{0.f, 0.f};
            lo = __builtin_elementwise_max(__builtin_elementwise_fma(lo, pre_sc[0], pre_sh[0]), zero);
            hi = __builtin_elementwise_max(__builtin_elementwise_fma(hi, pre_sc[1], pre_sh[1]), zero);
            lo = ok ? lo : zero;
            hi = ok ? hi : zero;
        }
        *(f32x2 *)dst = lo;
        *(f32x2 *)(dst + 8) = hi;
    };

    // ---- fragments
    const int a_lane = ks * A_BUF + (4 * rb) * RS + 68 * LR + 2 * G;   // floats: patch origin of tile LR in row block rb, quad G
    auto rd = [&](int buf, int r, int j, int h) -> w4v2 {   // buf, r, j, h are compile-time: one base register + an immediate
        return *(const w4v2 *)(smem + a_lane + (buf * (KS * A_BUF) + r * RS + 16 * j + 4 * (j >> 2) + 8 * h));
    };
    // weights: packed in the order the loop consumes them, [chunk][slot q][k-quad G][n / 32][n % 16][(n / 16) % 2][2]
    // (packing.py pack_conv_winograd4): the fragments of slot q for lane (G, n % 16) and BOTH column blocks of the wave are
    // one 16-byte load, and the stream advances by one constant stride per slot
    const unsigned b_lane = (unsigned)(((G * (p.npad >> 5) + (n0 >> 5) + cbp) * 16 + LR) * 16);   // bytes
    const unsigned qstride = (unsigned)(4 * p.npad) * 8u;                                     // bytes per slot
    unsigned wrow = (unsigned)ks * 72u * qstride;                                             // wave-uniform: first own chunk
    f32x4 fb[NR];   // [slot] = {cb 0: k 2h, 2h + 1; cb 1: k 2h, 2h + 1}
    // called once per slot, in order: fetches the fragments of slot q (mod 72) of the wave's stream; `skip` marks the first slot
    // of the wave's NEXT chunk (KS = 2: the other K half's chunk lies in between)
    auto frag_b = [&](int q, bool skip = false) {
        if (KS == 2 && skip) wrow += 72u * qstride;
        fb[q & (NR - 1)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)b_lane, (int)wrow, 0));
        wrow += qstride;
        asm volatile("" : "+s"(wrow));   // keep it ONE running scalar: hipcc otherwise keeps 72 loop-invariant products in (spilled) SGPRs
    };

    // 72 accumulator tiles = 288 registers: the accumulator file holds 256, and hipcc picks ONE register class for every
    // MFMA of a function (with the builtin it keeps all 72 tiles in AGPRs and shuffles the overflow through copies: 1584
    // v_accvgpr moves per K chunk and scratch spills).  So the MFMAs are inline asm: most tiles accumulate in AGPRs ("+a"),
    // the last ones in VGPRs ("+v").  hipcc pads no hazards of an asm statement (cdna_hip_programming.md 5.7): the two wait
    // states between a VALU write of an operand and the MFMA are inside the string of the FIRST MFMA of a slot (the
    // operands of a slot are computed in the slot before), the MFMA -> VALU read distance of the epilogue is the nop
    // statement after the loop; accumulate chains (D = C) need none.
    // taps 0..W4_ATAPS-1 in AGPRs, the rest in VGPRs (a few AGPRs are left to the register allocator: with all 256 taken it
    // spills accumulators to scratch around the epilogue)
#define X(t) f32x4 accA_##t = f32x4{0.f, 0.f, 0.f, 0.f}, accB_##t = f32x4{0.f, 0.f, 0.f, 0.f};
    W4_TAPS(X)
#undef X
    int acc_end_ = 0;   // closes the explicit capture lists (implicit capture does not reach into discarded if-constexpr branches)
    // the four MFMAs of a tap slot (k-steps 2h, 2h + 1 x the wave's two column blocks) are ONE asm statement on the tap's two
    // accumulator tiles: hipcc pads every asm boundary that a VALU instruction follows
#define X(t) &accA_##t, &accB_##t,
    auto mma4 = [W4_TAPS(X) & acc_end_](auto tap, float a0, float a1, f32x4 bw) {
#undef X
        constexpr int T = decltype(tap)::value;
#define X(t)                                                                                                                 \
    if constexpr (T == t) {                                                                                                 \
        if constexpr (t < W4_ATAPS)                                                                                         \
            asm volatile(W4_MFMA4 : "+a"(accA_##t), "+a"(accB_##t) : "v"(a0), "v"(a1), "v"(bw[0]), "v"(bw[1]), "v"(bw[2]), "v"(bw[3])); \
        else                                                                                                                \
            asm volatile(W4_MFMA4 : "+v"(accA_##t), "+v"(accB_##t) : "v"(a0), "v"(a1), "v"(bw[0]), "v"(bw[1]), "v"(bw[2]), "v"(bw[3])); \
    }
        W4_TAPS(X)
#undef X
    };
#define X(t) &accA_##t, &accB_##t,
    auto acc_of = [W4_TAPS(X) & acc_end_](auto tile) -> f32x4 {   // tile = tap * 2 + column block
#undef X
        constexpr int IDX = decltype(tile)::value;
#define X(t)                                       \
    if constexpr (IDX == 2 * t) return accA_##t;   \
    if constexpr (IDX == 2 * t + 1) return accB_##t;
        W4_TAPS(X)
#undef X
    };

    // ---- stage 1 (B^T over the patch rows).  R[r - 1][j] = patch rows 1..4 of the current half (every phase needs them: read
    // once); rows 0 and 5 are read when phase C's rows are formed.  W[parity][row in phase][patch column].
    w4v2 R[4][6], W[2][2][6], e0, e5;
    auto s1_pair = [&](int par, int ph, int j) {   // phases A (+-a) and B (+-b) from R
        if (RAFT_WINO4_ABL & 2) {
            W[par][0][j] = W[par][1][j] = w4v2{1.f, 1.f};
            return;
        }
        const float c2 = ph == 0 ? B2 : A2, c1 = ph == 0 ? PA : PB;
        const w4v2 t1 = R[3][j] - c2 * R[1][j], t2 = R[2][j] - c2 * R[0][j];
        W[par][0][j] = t1 + c1 * t2;
        W[par][1][j] = t1 - c1 * t2;
    };
    auto s1_edge = [&](int par, int j) {           // phase C (0, inf) from R and the freshly read rows 0 / 5
        if (RAFT_WINO4_ABL & 2) {
            W[par][0][j] = W[par][1][j] = w4v2{1.f, 1.f};
            return;
        }
        W[par][0][j] = (R[3][j] - SS * R[1][j]) + PP * e0;
        W[par][1][j] = (e5 - SS * R[2][j]) + PP * R[0][j];
    };
    // ---- stage 2: V[tx] of a row from its W[0..5]; computed one tap ahead (pairs together)
    w4v2 V[6];
    auto s2_calc = [&](const w4v2 *wr, int tx) {   // fills V[tx] (and V[tx + 1] for the first tap of a pair)
        if (RAFT_WINO4_ABL & 16) {
            V[tx] = wr[tx];
            return;
        }
        if (tx == 0) {
            V[0] = (wr[4] - SS * wr[2]) + PP * wr[0];
        } else if (tx == 1) {
            const w4v2 t1 = wr[4] - B2 * wr[2], t2 = wr[3] - B2 * wr[1];
            V[1] = t1 + PA * t2;
            V[2] = t1 - PA * t2;
        } else if (tx == 3) {
            const w4v2 t1 = wr[4] - A2 * wr[2], t2 = wr[3] - A2 * wr[1];
            V[3] = t1 + PB * t2;
            V[4] = t1 - PB * t2;
        } else if (tx == 5) {
            V[5] = (wr[5] - SS * wr[3]) + PP * wr[1];
        }
    };

    // ---- prologue: chunk 0 into buffer 0, rows 1..4 + stage 1 of (chunk 0, half 0, phase A), the first PF weight fragments
    next_source(0);
#pragma unroll
    for (int i = 0; i < NA; ++i) gload_item(i);
#pragma unroll
    for (int q = 0; q < PF; ++q) frag_b(q);
#pragma unroll
    for (int i = 0; i < NA; ++i) lstore_item(i, 0);
    raft_barrier_lds();
#pragma unroll
    for (int j = 0; j < 6; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) R[r][j] = (RAFT_WINO4_ABL & 2) ? w4v2{1.f, 1.f} : rd(0, r + 1, j, 0);
        s1_pair(0, 0, j);
    }
    s2_calc(W[0][0], 0);

    // one 16-channel chunk; BUF = its LDS buffer (compile-time: every LDS access is base register + immediate)
    auto chunk = [&](auto buf_c, int c) {
        constexpr int BUF = decltype(buf_c)::value;
        next_source(c + 1);
        w4_static_for<6>([&](auto inst_c) {
            constexpr int inst = decltype(inst_c)::value;
            constexpr int h = inst / 3, ph = inst % 3, par = inst & 1;
            // under this instance's MFMAs: phase A -> rows (+-b) of the same half from R; phase B -> rows 0 / 5 are read and
            // rows (0, inf) formed; phase C -> R is dead: rows 1..4 of the NEXT half are read and its rows (+-a) formed
            constexpr int n_h = ph == 2 ? 1 - h : h;
            constexpr int n_buf = (ph == 2 && h == 1) ? 1 - BUF : BUF;
            if (inst == 5) raft_barrier_lds();   // chunk c + 1 is in the other buffer; every read of this one has been consumed
            w4_static_for<12>([&](auto u_c) {
                constexpr int u = decltype(u_c)::value;
                constexpr int row = u / 6, tx = u % 6, j = u >> 1;
                constexpr int q = inst * 12 + u;
                constexpr int t = tap_row(ph, row) * 6 + tx;
                // weight fragments PF slots ahead (runs into the next chunk; past the last one: out of range, 0)
                if (!(RAFT_WINO4_ABL & 1)) frag_b((q + PF) % 72, q + PF == 72);
                // stage 1 of the next instance (unconditional: under the last chunk's last phase it transforms stale data that
                // nobody uses -- a branch here lets the compiler sink the reads next to the arithmetic)
                if (!(RAFT_WINO4_ABL & 2)) {
                    if constexpr (ph == 1 && (u & 1) == 0) {
                        e0 = rd(BUF, 0, j, h);
                        e5 = rd(BUF, 5, j, h);
                    }
                    if constexpr (ph == 2 && (u & 1) == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) R[r][j] = rd(n_buf, r + 1, j, n_h);
                    }
                }
                if constexpr ((u & 1) == 1) {
                    if constexpr (ph == 0) s1_pair(par ^ 1, 1, j);
                    if constexpr (ph == 1) s1_edge(par ^ 1, j);
                    if constexpr (ph == 2) s1_pair(par ^ 1, 0, j);
                }
                // the next chunk's halo tile, one item per slot
                {
                    constexpr int li = q - LOAD_SLOT0, si = q - LOAD_SLOT0 - STORE_LAG;
                    if constexpr (li >= 0 && li < NA) gload_item(li);
                    if constexpr (si >= 0 && si < NA) lstore_item(si, 1 - BUF);
                }
                W4_SB();
                mma4(std::integral_constant<int, t>{}, V[tx][0], V[tx][1], fb[q & (NR - 1)]);
                W4_SB();
                // stage 2 one tap ahead (the next row after the last tap of a row; after the last tap of an instance V[0] of the
                // next instance's first row, whose W was completed in this slot)
                if constexpr (u < 11) s2_calc(W[par][(u + 1) / 6], (u + 1) % 6);
                if constexpr (u == 11) s2_calc(W[par ^ 1][0], 0);
            });
        });
    };
    // two chunks per trip (one per LDS buffer).  An odd chunk count runs one ghost chunk: its halo tile was fetched through an
    // empty descriptor (zeros) and its weights lie beyond the stream (zeros), so it adds nothing -- a branch around it would put
    // 72 accumulator tiles through phi copies
    for (int st = 0; st < nst; st += 2) {
        chunk(std::integral_constant<int, 0>{}, st);
        chunk(std::integral_constant<int, 1>{}, st + 1);
    }

    // the last MFMAs' results must have left the matrix pipe before a VALU reads them (8-pass MFMA: 12 wait states)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane owns channel n; register r of an accumulator is tile 4G + r of the row block
    constexpr bool HAS_E0 = EPI == EPI_RES;
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.o0, 0, (int)((((long)M - 1) * p.ldo0 + p.nvalid) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t re0 = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(HAS_E0 ? (const void *)p.e0 : (const void *)p.o0), 0,
        HAS_E0 ? (int)((((long)M - 1) * p.lde0 + p.nvalid) * 4) : 0, 0x00020000);
    auto bstore = [](float v, __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, soff, 0);
    };
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
    };
    const bool interior = (y0 + TH <= p.H) & (x0 + TW <= p.W);   // wave-uniform
    // the epilogue's lane indices are re-derived from an opaque copy of the thread id: otherwise hipcc evaluates the
    // epilogue's addresses before the K loop and carries them through it in registers the loop does not have (spills)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int G_e = (tid_e >> 4) & 3, LR_e = tid_e & 15, rb_e = (tid_e >> 6) & 1, cbp_e = tid_e >> 7;
    const int xb = x0 + 16 * G_e;                                // the lane's tiles 4G + r: pixels xb + 4 r + jx
    const int ks_e = KS == 1 ? 0 : (tid_e >> 7), cb_e = KS == 1 ? cbp_e : ((tid_e >> 6) & 1), w_e = tid_e >> 6;
    // one output row `yy` (image row) of the lane's four tiles for column block j: bias, activation / residual, stores.
    // Element (r, jx) = pixel xb + 4 r + jx: wave-uniform byte offset from the lane base; tiles cut by the image border
    // (`interior` false, wave-uniform) add the per-element out-of-range bit.
    // lane bases of the (up to four) output rows this wave finishes: KS = 1 rows 4 rb + i, KS = 2 rows 2 ks + i (i < 2)
    constexpr int NROW = KS == 1 ? 4 : 2;
    unsigned bo[NROW], be[NROW];
    bool rowok[NROW];
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        const int yy = y0 + (KS == 1 ? 4 * rb_e : 2 * ks_e) + i;
        const unsigned pix0 = (unsigned)((b * p.H + yy) * p.W + xb);
        bo[i] = pix0 * p.ldo0 * 4u;
        be[i] = HAS_E0 ? pix0 * p.lde0 * 4u : 0u;
        rowok[i] = yy < p.H;
    }
    float bias2[2];
    unsigned nofs[2];
    bool nok2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (cb_e * 2 + j) * 16 + LR_e;
        nok2[j] = n < p.nvalid;                                            // channel beyond nvalid: every access out of range
        nofs[j] = (unsigned)n * 4u;
        bias2[j] = p.bias[n];                                              // bias has npad entries
    }
    // rows of two tiles (registers 2 rh, 2 rh + 1 of the accumulators): bias, activation / residual, stores.  Element (r, jx) =
    // pixel xb + 4 r + jx: wave-uniform byte offset from the lane base; tiles cut by the image border (`interior` false,
    // wave-uniform) add the per-element out-of-range bit.  i = index into bo / be / rowok (compile-time at every call site).
    float st1[2] = {0.f, 0.f}, st2[2] = {0.f, 0.f};                     // STATS: (sum, sum of squares) of the lane's outputs per column block
    auto emit = [&](int j, int i, int rh, const f32x2 *Y) {
        const float bias = bias2[j];
        const unsigned vo = nok2[j] ? bo[i] + nofs[j] : RAFT_OOB, ve = (HAS_E0 && nok2[j]) ? be[i] + nofs[j] : RAFT_OOB;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * rh + rr;
            float xv[4];
            if (HAS_E0) {
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const unsigned dead = (interior | (rowok[i] & (xb + 4 * r + jx < p.W))) ? 0u : RAFT_OOB;
                    xv[jx] = bload(re0, ve | dead, (4 * r + jx) * p.lde0 * 4);
                }
            }
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                float v = Y[jx][rr] + bias;
                if (EPI == EPI_RES) {
                    v = fmaxf(xv[jx] + fmaxf(v, 0.f), 0.f);
                } else {
                    if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                    v *= p.scale;
                }
                if ((RAFT_WINO4_ABL & 8) && v != 12345.678f) continue;
                const unsigned dead = (interior | (rowok[i] & (xb + 4 * r + jx < p.W))) ? 0u : RAFT_OOB;
                if constexpr (STATS) {
                    // masked as DATA (0x80000000 -> all ones): reusing the lane condition behind `dead` keeps 32 of them alive in
                    // SGPR pairs across the column block (61 spilled SGPRs)
                    unsigned dd = dead;
                    asm volatile("" : "+v"(dd));
                    const float vs = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & ~(unsigned)((int)dd >> 31));
                    st1[j] += vs;
                    st2[j] = fmaf(vs, vs, st2[j]);
                }
                bstore(v, ro0, vo | dead, (4 * r + jx) * p.ldo0 * 4);
            }
        }
    };
    // KS = 2: the two K halves of a (row block, column block pair) meet after the (linear) output transform: wave set ks
    // finishes output rows 2 ks, 2 ks + 1 of every tile and hands the other two rows of its partial result to its partner
    // (wave w ^ 2) through LDS -- [j][rh][destination wave][row][jx][lane] float2 in the halo buffers (32 KB)
    f32x2 *xch = (f32x2 *)smem;
    if (KS == 2) raft_barrier_lds();   // every wave is done reading its last halo tile
    // one column block j and one PAIR of tiles (accumulator registers 2 rh, 2 rh + 1) at a time: A^T over the tap rows for
    // each tap column (T[i][tx]), then A over the tap columns -- 48 registers of T instead of 96 for whole accumulators
    w4_static_for<4>([&](auto jr_c) {
        constexpr int j = decltype(jr_c)::value >> 1, rh = decltype(jr_c)::value & 1;
        f32x2 T[4][6], keep[2][4];
        w4_static_for<6>([&](auto tx_c) {
            constexpr int tx = decltype(tx_c)::value;
            auto half = [&](f32x4 a) { return f32x2{a[2 * rh], a[2 * rh + 1]}; };
            const f32x2 m0 = half(acc_of(std::integral_constant<int, 2 * tx + j>{})),
                        m1 = half(acc_of(std::integral_constant<int, 2 * (6 + tx) + j>{})),
                        m2 = half(acc_of(std::integral_constant<int, 2 * (12 + tx) + j>{})),
                        m3 = half(acc_of(std::integral_constant<int, 2 * (18 + tx) + j>{})),
                        m4 = half(acc_of(std::integral_constant<int, 2 * (24 + tx) + j>{})),
                        m5 = half(acc_of(std::integral_constant<int, 2 * (30 + tx) + j>{}));
            const f32x2 sa = m1 + m2, da = m1 - m2, sb = m3 + m4, db = m3 - m4;
            T[0][tx] = m0 + (sa + sb);
            T[1][tx] = PA * da + PB * db;
            T[2][tx] = A2 * sa + B2 * sb;
            T[3][tx] = (A3 * da + B3 * db) + m5;
            __builtin_amdgcn_sched_barrier(0);   // keeps the accumulator reads of later columns from being hoisted (registers)
        });
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 sa = T[i][1] + T[i][2], da = T[i][1] - T[i][2], sb = T[i][3] + T[i][4], db = T[i][3] - T[i][4];
            f32x2 Y[4];
            Y[0] = T[i][0] + (sa + sb);
            Y[1] = PA * da + PB * db;
            Y[2] = A2 * sa + B2 * sb;
            Y[3] = (A3 * da + B3 * db) + T[i][5];
            if (KS == 1) {
                emit(j, i, rh, Y);
            } else {
                const bool mine = (i >> 1) == ks_e;   // wave-uniform
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    if (mine)
                        keep[i & 1][jx] = Y[jx];
                    else
                        xch[(((((j * 2 + rh) * 4 + (w_e ^ 2)) * 2 + (i & 1)) * 4 + jx) << 6) + (tid_e & 63)] = Y[jx];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KS == 2) {   // one exchange per (column block, tile pair), each in its own LDS region: no barrier before the next one's writes
            raft_barrier_lds();
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
                f32x2 Y[4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx)
                    Y[jx] = keep[i2][jx] + xch[(((((j * 2 + rh) * 4 + w_e) * 2 + i2) * 4 + jx) << 6) + (tid_e & 63)];
                emit(j, i2, rh, Y);
            }
        }
    });
    if constexpr (STATS) {   // lanes LR, LR + 16, LR + 32, LR + 48 hold the same channel; entry = (pixel tile, row block)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s1 = st1[j], s2 = st2[j];
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int n = n0 + (cb_e * 2 + j) * 16 + LR_e;
            if (G_e == 0) *(float2 *)(p.stats + ((long)(2 * mt + rb_e) * p.npad + n) * 2) = make_float2(s1, s2);
        }
    }
}

#pragma clang diagnostic pop

// launcher (conv_wino4.hip); `a.wp` holds the F(4x4, 3x3)-transformed weights in consumption order
// (Cin/16, 72, 4, npad/32, 16, 2, 2) -- tf_raft_amd/packing.py pack_conv_winograd4.  epi: EPI_LINEAR / EPI_RELU / EPI_RES;
// a.stats != NULL (with EPI_LINEAR) selects STATS, a.pre_scale != NULL PRE on top of it (the instance-norm encoder's
// combinations; one source, two-row-block workgroups).  Stats entries per image: 2 * ceil(H/8) * ceil(W/64).
// ks_hint = 1 / 2: the caller's choice of the workgroup shape where the launcher would decide by grid size (an explicit
// RAFT_WINO4_KS still wins).
int raft_launch_conv_wino4(const ConvArgs &a, int epi, hipStream_t s, int decide_npad = 0, int ks_hint = 0);
