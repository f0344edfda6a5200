/*
 * raft_hip.h -- C ABI of libraft_hip.so: the MI355X (gfx950) RAFT forward-prediction hot path.
 *
 * Every entry point replaces one piece of the reference's Python/TensorFlow path
 * (daigo0927/tf-raft); the reference interface each one stands in for is cited as
 * file:line into the reference tree.  The reference has no FFI of its own (it is pure
 * Python on TensorFlow ops), so "what its FFI for this path would bind" is exactly the set
 * of op groups listed here; INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes, no torch / C++ types;
 *   - all tensors are fp32, NHWC (channels contiguous), exactly the reference's layouts;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only
 *     enqueue work, they never synchronise and never allocate.  Two documented exceptions:
 *     raft_loop_ctx_create / _destroy (the caller-owned context of the three-stream loops: 4 HIP
 *     events + a cache of instantiated hipGraphs) and the bench-only raft_iterate_basic_timed_f32;
 *   - return value: RAFT_OK (0), a negative RAFT_E_* argument error, or a positive
 *     hipError_t from the launch; no exception or abort crosses the ABI;
 *   - the library keeps no per-call state and is re-entrant (ordering only through `stream`).  Its
 *     only process-global state is the table of tuning switches below (raft_set_option).
 *
 * Tuning switches (tests, A/B timing, ablation -- never needed for correct results; every setting
 * computes the same function within the per-kernel tolerances).  Each is an integer with a built-in
 * default; the table is initialised ONCE from the environment variables of the same names when the
 * library is loaded and is changed afterwards only through raft_set_option -- the launch path reads
 * an atomic, it never calls getenv:
 *   RAFT_CONV_TILE      "<code>" or "<npad>:<taps>:<code>,...": tile of the direct convolution kernels
 *                       (100 + 10*TH + TN: TH x 16-pixel x 64*TN-channel halo tiles, TH in {4,7,8}, TN in {1,2})
 *   RAFT_CONV_WINO      bit mask {1 convc2, 2 convf2, 4 conv, 8 fh1_mask0}: layers on the F(2x2,3x3) kernel (13)
 *   RAFT_CONV_WINO4     the same mask (bit 2 = convf2) for the F(4x4,3x3) kernel, preferred where its bit is set and the 6x6-tap weights
 *                       were supplied        (default by launch size: 0 for a single 448x512 pair, 8 = fh1_mask0 from 2 pairs, + 1 | 2 = convc2, convf2 from 3, + 4 = conv from 8)
 *   RAFT_WINO4_KS       1/2  F(4x4,3x3) kernel: 8 x 64-pixel workgroups / 4 x 64-pixel workgroups with K split between two
 *                            wave sets                                                        (default: by grid size)
 *   RAFT_SMALL_WINO     bit mask {1 conv, 2 gru_zr, 4 gru_q, 8 fh1} of the SmallUpdateBlock      (default 15)
 *   RAFT_GRU_WINO       bit mask {1 zr1, 2 q1, 4 zr2, 8 q2}: SepConvGRU layers on F(2,5)         (default 15)
 *   RAFT_GRU_WINO4      the same mask for F(4,5), preferred where both bits are set              (default 15)
 *   RAFT_WINO_TNW       1/2  32- or 64-channel workgroups of the Winograd kernels                (default: by grid size)
 *   RAFT_WINO_SB        0/1  pinned weight prefetch of the F(2x2,3x3) kernel                      (default: by grid size)
 *   RAFT_WINO1D_TM      1/2  half- / full-height F(2,5) tiles                                    (default: by grid size)
 *   RAFT_WINO_CK        1/2  16 or 32 channels per barrier (4: 64, split-K kernel only)          (default: by grid size)
 *   RAFT_WINO_KS        1/2  F(2x2,3x3) kernel: K split between two wave sets of a 512-thread workgroup (default: 2 for
 *                            launches of fewer wave-tasks than SIMDs, i.e. single pairs)
 *   RAFT_ENC_WINO       0/1  encoder ResBlock 3x3 layers on the F(2x2,3x3) kernel                 (default 1)
 *   RAFT_ENC_WINO4      bit mask {1 layer1, 2 layer2, 4 layer3}: stride-1 3x3 layers of those encoder stages on the
 *                       F(4x4,3x3) kernel where a transformed copy is present (block_w44)      (default: every layer whose launch has more than 256 workgroups)
 *   RAFT_LOOKUP_FUSED   0/1  prediction loops: lookup + convc1 as two kernels / fused (raft_lookup_convc1_f32)  (default 1)
 *   RAFT_MASK_FUSED     0/1  prediction loops: mask.2 + convex upsampling as two kernels / one (the mask is never stored) (default: 1 from 2 pairs at 448x512 on)
 *   RAFT_CONVC2_KS      1/2  convc2 on the F(4x4) kernel in the loops: 8-row workgroups / K-split 4-row workgroups   (default: by grid size)
 *   RAFT_CONVF2_KS      1/2  the same for convf2                             (default: K-split below 56 eight-row workgroups)
 *   RAFT_EVENT_FENCE    0/1  cross-stream events of a raft_loop_ctx without / with the system-scope fence of a default HIP event
 *                            (read when the context is created)                                    (default 1 since round 6; 0: the
 *                            events only order streams of one device, +0.4 .. 0.9 % on the three-stream loop, profiles/r10c_event_fence.txt)
 *   RAFT_CORR_XCD       0/1/n  volume build: plain (n, m, batch) tile grid / one region of the tile plane per XCD, walked in strips
 *                            of 2 (n >= 2: n) column tiles                                          (default 1)
 *   RAFT_CORR_POOL      0/1  volume build: pyramid level 1 as extra GEMM columns against the pooled fmap2 / pooled from the level-0
 *                            accumulators in the epilogue (2x2 averages by two DPP adds; 18 % fewer workgroups; the reference's own
 *                            summation order, corr.py:106-114).  1 needs even map sizes and an even number of 4x8 tiles per
 *                            row and column (448x512 and 1024x1024 frames have them), else 0 is used         (default 1)
 *   RAFT_ONDEMAND_BLOCK 0/1  on-demand lookup: wave per query / 4x8 query blocks on MFMA         (default 1)
 *   RAFT_LOOP_GRAPH     0/1  three-stream loops replayed as one hipGraph launch                  (default 0: measured
 *                       slower than stream launches on ROCm 7.2 at every batch size)
 */
#ifndef RAFT_HIP_H_
#define RAFT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAFT_HIP_VERSION 219          /* 0.2.0: ABI stamp, checked by the Python binding -- bump on ANY struct / signature change */
#define RAFT_MAX_LEVELS 4

enum {
    RAFT_OK = 0,
    RAFT_E_NULL = -1,        /* a required pointer is NULL */
    RAFT_E_SHAPE = -2,       /* a dimension is non-positive or inconsistent */
    RAFT_E_UNSUPPORTED = -3, /* radius / channel count / kernel size not instantiated */
    RAFT_E_ALIGN = -4        /* a pointer or leading dimension is not 16-byte aligned */
};

int raft_version(void);
/* Human-readable message for a return code (RAFT_E_* or hipError_t). */
const char *raft_error_string(int rc);

/* Tuning switches (table above).  value: decimal text (RAFT_CONV_TILE: its rule list); "" = unset (built-in default);
 * NULL = back to the load-time (environment) state.  RAFT_E_UNSUPPORTED for an unknown name.  Thread-safe; takes effect
 * for launches enqueued after the call. */
int raft_set_option(const char *name, const char *value);
/* Launch-shape hint of the CALLING THREAD (thread-local, not process-global): its subsequent launches share the device with
 * n - 1 other independent launch sequences of about the same size -- the lanes of tf_raft_amd/model.py's pipelined forward,
 * several recurrent loops in flight on streams of their own.  The launchers' "does this grid fill the chip?" rules (Winograd
 * variant per layer, K-split / channel width of a workgroup, fused mask head) then count a grid n times, i.e. choose the
 * shapes of an n-times larger batch: less CU-time per launch instead of less latency.  Every shape computes the same function
 * within the per-kernel tolerances (as with the switches above).  n <= 1 restores the default.  Returns the previous value. */
int raft_set_thread_concurrency(int n);
/* Current value as text into buf ("" while unset). */
int raft_get_option(const char *name, char *buf, size_t len);

/* CRC-32C (Castagnoli, reflected 0x1EDC6F41) of `n` bytes, continuing from `crc` (0 to start).
 * Host-only.  The checksum of the TensorFlow tensor-bundle checkpoint files the reference
 * stores its weights in (README.md:66-96); used by tf_raft_amd/checkpoint.py. */
uint32_t raft_crc32c(uint32_t crc, const void *data, size_t n);

/* ------------------------------------------------------------------ correlation volume */

/* Pyramid geometry.  Level l holds, for each of the B*h*w query pixels, an (lh[l], lw[l]) map;
 * lh/lw follow tf.nn.avg_pool2d(2, 2, 'VALID') (floor).  A map is stored as 4x8 tiles of 32
 * floats (one 128-byte line each), tiles row-major, padded to whole tiles:
 *   map_floats(l) = ceil(lh/4) * ceil(lw/8) * 32,
 *   element (y, x) at ((y/4) * ceil(lw/8) + x/8) * 32 + (y%4) * 8 + x%8     (padding is zero),
 * so that the (2r+2)^2 lookup footprint pulls ~7 lines of a large map instead of ~13.
 * level_offsets[l] is the float offset of level l inside one allocation (level l = B*h*w maps
 * back to back), level_offsets[levels] the total float count.  Host-only helper (no GPU work).
 * reference corr.py:106-114. */
int raft_corr_pyramid_layout(int B, int h, int w, int levels,
                             int64_t *level_offsets /* [levels+1] */,
                             int *lh /* [levels] */, int *lw /* [levels] */);

/* Float count of the fmap2 feature-pyramid workspace raft_corr_build_f32 needs. */
int64_t raft_corr_build_workspace_floats(int B, int h, int w, int C, int levels);

/* All-pairs correlation volume + 4-level pyramid.
 * Replaces CorrBlock.__init__ / CorrBlock.correlation (reference corr.py:100-114, 154-162):
 *   corr[b, q, t] = <fmap1[b, q, :], fmap2[b, t, :]> / sqrt(C), then 3x avg_pool2d over t.
 * fmap1, fmap2: (B, h, w, C).  pyr: float[level_offsets[levels]]; level l is laid out
 * (B*h*w, map_floats(l)) in the tiled map layout above == the reference's corr_pyramid[l]
 * (B*h*w, lh[l], lw[l], 1) after un-tiling.
 * Levels > 0 are computed as <fmap1, avgpool_l(fmap2)> / sqrt(C) (average pooling over the
 * target dims commutes with the dot product); `workspace` holds the pooled fmap2 pyramid. */
int raft_corr_build_f32(const float *fmap1, const float *fmap2, int B, int h, int w, int C,
                        int levels, float *pyr, const int64_t *level_offsets,
                        float *workspace, void *stream);

/* Windowed bilinear pyramid lookup.  Replaces CorrBlock.retrieve + bilinear_sampler
 * (reference corr.py:116-152, 28-69), including its exact semantics: clamp, ceil/floor
 * weights (integer or out-of-range coordinate => 0), window axis 0 offsets x.
 * coords: (B, h, w, 2) xy.  out: (B, h, w, ld_out) with channel = lvl*(2r+1)^2 + a*(2r+1) + b
 * in the first levels*(2r+1)^2 channels; channels beyond that are left untouched.
 * radius 3 and 4 are instantiated. */
int raft_corr_lookup_f32(const float *pyr, const int64_t *level_offsets, const float *coords,
                         int B, int h, int w, int levels, int radius,
                         float *out, int ld_out, void *stream);

/* Same lookup without a stored volume ("alternate" correlation; no reference code --
 * the reference README.md:109 notes its absence): correlations of the (2r+2)^2 footprint are
 * computed on demand from fmap1 and the pooled fmap2 pyramid held in `fmap2_pyr`
 * (the workspace layout of raft_corr_build_f32, filled by raft_fmap_pyramid_f32). */
int raft_fmap_pyramid_f32(const float *fmap2, int B, int h, int w, int C, int levels,
                          float *fmap2_pyr, void *stream);
int raft_corr_lookup_ondemand_f32(const float *fmap1, const float *fmap2_pyr,
                                  const float *coords, int B, int h, int w, int C,
                                  int levels, int radius, float *out, int ld_out,
                                  void *stream);

/* bilinear_sampler (reference corr.py:28-69) as a standalone op.
 * image: (n, h, w[, 1]); coords: (n, kh, kw, 2) xy; out: (n, kh, kw[, 1]). */
int raft_bilinear_sampler_f32(const float *image, const float *coords, int64_t n, int h, int w,
                              int kh, int kw, float *out, void *stream);

/* coords_grid (reference corr.py:72-90): coords[b, y, x] = (x, y). */
int raft_coords_grid_f32(float *coords, int B, int h, int w, void *stream);

/* ------------------------------------------------------------------ upsampling */

/* RAFT.upsample_flow (reference model.py:39-66): softmax over the 9 taps of
 * mask[b, y, x, (i*8 + j)*9 + k], 3x3 zero-padded neighbourhood of 8*flow, depth_to_space(8).
 * flow: (B, h, w, 2); mask: (B, h, w, 576); out: (B, 8h, 8w, 2). */
int raft_upsample_convex_f32(const float *flow, const float *mask, int B, int h, int w,
                             float *out, void *stream);

/* upflow8 (reference corr.py:93-96): 8 * tf.image.resize(flow, (8h, 8w), 'bilinear')
 * with TF2 half-pixel centres.  flow: (B, h, w, 2); out: (B, 8h, 8w, 2). */
int raft_upflow8_f32(const float *flow, int B, int h, int w, float *out, void *stream);

/* Measurement utility (no reference counterpart): 16-byte-per-lane streaming copy of n floats
 * (n % 4 == 0, 16-byte aligned pointers).  bench.py times it to state the HBM copy bandwidth of
 * the box it runs on -- the "measured roofline" the lookup / build / upsample kernels are quoted
 * against next to the 8 TB/s datasheet figure. */
int raft_stream_copy_f32(const float *src, float *dst, int64_t n, void *stream);

/* Measurement utility (no reference counterpart): `blocks` workgroups of 256 threads, every wave issuing `iters` x 8
 * independent v_mfma_f32_16x16x4_f32 (nothing else in the loop; non-zero lane-varying operands); out: blocks * 256
 * floats (written so that the loop cannot be optimised away).  FLOPs = blocks * 4 * iters * 8 * 2048.  bench.py times it
 * to state the fp32-MFMA rate the box SUSTAINS (power-limited clock) next to the 157.3 TFLOP/s datasheet figure every
 * `roofline.frac` is quoted against. */
int raft_mfma_probe_f32(float *out, int blocks, int iters, void *stream);

/* ------------------------------------------------------------------ convolutions */

/* Packed weight layout of the implicit-GEMM convolution: for a Keras kernel (kh, kw, Cin, Cout)
 *   wp[((t * (Kpad/4) + k/4) * npad + n) * 4 + k%4] = kernel[t / kw, t % kw, k, n]
 * with Kpad = Cin rounded up per input source to a multiple of 32 and npad = Cout rounded up
 * to a multiple of 64; padding is zero.  bias is float[npad].  The host mirror
 * (tf_raft_amd/packing.py) produces these once per model. */

enum { RAFT_ACT_NONE = 0, RAFT_ACT_RELU = 1 };

/* One stride-1 'same' Keras Conv2D (+ optional relu, + scale) on fp32 MFMA.
 * Replaces a layers.Conv2D call (reference update.py:10-11, 91-95, 138-140).
 * The input is the channel concatenation of up to two NHWC sources: channels [0, c0) of a0
 * (pixel stride lda0 floats) followed by [0, c1) of a1 (c1 may be 0, a1 NULL); c0 and c1 must be
 * multiples of 32 (pad with zero channels).  out[pixel * ldo + n] for n < nvalid.
 * (kh, kw) in {(1,1), (3,3), (1,5), (5,1)}. */
int raft_conv2d_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                    const float *wp, const float *bias, int B, int H, int W, int kh, int kw,
                    int npad, int nvalid, int act, float scale, float *out, int ldo,
                    void *stream);

/* The same convolution for 3x3 kernels by Winograd F(2x2, 3x3) (fp32 arithmetic, 2.25x fewer multiplies; what
 * cuDNN runs under the reference's Conv2D on a GPU; not bit-identical to the direct kernel).  `wp` = the
 * transformed kernel G g G^T packed as a 4x4-tap kernel; c0, c1 multiples of 16, npad a multiple of 32. */
int raft_conv2d_winograd_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                             const float *wp, const float *bias, int B, int H, int W, int npad,
                             int nvalid, int act, float scale, float *out, int ldo, void *stream);

/* The same by Winograd F(4x4, 3x3) (36 multiplies per 16 outputs: 4x fewer than the direct kernel, 1.78x fewer than
 * F(2x2, 3x3); fp32, points {0, +-5/8, +-3/2, inf}; deviation from the float64 convolution ~3x that of F(2x2, 3x3)).
 * `wp` = the transformed kernel in the consumption order of the kernel, (Cin/16, 72, 4, npad/32, 16, 2, 2) -- packing.py
 * pack_conv_winograd4; c0, c1 multiples of 16, npad a multiple of 64. */
int raft_conv2d_winograd4_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                              const float *wp, const float *bias, int B, int H, int W, int npad,
                              int nvalid, int act, float scale, float *out, int ldo, void *stream);

/* 1x5 (kh = 1, kw = 5) or 5x1 convolution by 1-D Winograd F(2, 5) (6 multiplies per output pair instead of 10;
 * fp32, points {0, +-1, +-1/2, inf}).  `wp` = the transformed kernel packed with 6 taps; c0, c1 multiples of 16. */
int raft_conv1d_winograd_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                             const float *wp, const float *bias, int B, int H, int W, int kh, int kw,
                             int npad, int nvalid, int act, float scale, float *out, int ldo, void *stream);

/* The same by 1-D Winograd F(4, 5) (8 multiplies per 4 outputs; points {0, +-1, +-1/2, +-2, inf}).  `wp` = the
 * transformed kernel packed with 8 taps (packing.py pack_conv_winograd1d(..., m=4)); c0, c1 multiples of 32. */
int raft_conv1d_winograd4_f32(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                              const float *wp, const float *bias, int B, int H, int W, int kh, int kw,
                              int npad, int nvalid, int act, float scale, float *out, int ldo, void *stream);

/* cor1 = relu(convc1(CorrBlock.retrieve(coords))) in ONE kernel (reference corr.py:116-152 + update.py:91, 98): the
 * window values are those of raft_corr_lookup_f32 bit for bit, but they live only in LDS as the A operand of the 1x1
 * convolution -- the (B, h, w, 324) lookup output is neither written nor re-read.  levels = 4, radius = 4, npad = 256.
 * wp / bias: packing.py pack_convc1_fused.  out: (B*h*w, ldo), first nvalid channels written. */
int raft_lookup_convc1_f32(const float *pyr, const int64_t *level_offsets, const float *coords, int B, int h, int w,
                           const float *wp, const float *bias, int npad, int nvalid, float *out, int ldo, void *stream);

/* ------------------------------------------------------------------ update block */

typedef struct raft_conv_weights {
    const float *wp;     /* packed kernel (see above) or a layer-specific layout */
    const float *bias;
    int npad;
} raft_conv_weights;

/* BasicUpdateBlock weights (reference update.py:128-153).  z and r convolutions of each GRU
 * half are fused along N (npad 256 = [z | r]); flow_head.conv1 and mask[0] are fused along N
 * (npad 512 = [flow_head.conv1 | mask.0]).
 * convf1: (7,7,2,128) kept in Keras layout [t][c][n] (98 x 128 floats).
 * fh2: flow_head.conv2 (3,3,256,2) kept in Keras layout [t][c][2].
 * SepConvGRU (update.py:38-67), input order hx = [h(128) | inp(128) | motion(126) | flow(2)]: the rows of
 * convz / convr / convq that multiply `inp` are split off into gru_ctx1 (1x5) / gru_ctx2 (5x1), each
 * (kh,kw,128,384) = [z | r | q] WITH the three biases; gru_zr{1,2} / gru_q{1,2} keep the h rows and the
 * [motion | flow] rows (K = 256 per tap) and carry ZERO bias.  `inp` is constant over the prediction
 * loop, so its contribution is evaluated once per forward (raft_gru_context_f32). */
typedef struct raft_basic_update_weights {
    raft_conv_weights convc1, convc2, convf1, convf2, conv;
    raft_conv_weights gru_zr1, gru_q1, gru_zr2, gru_q2;
    raft_conv_weights fh1_mask0, fh2, mask2;
    raft_conv_weights gru_ctx1, gru_ctx2;
    /* optional (wp == NULL: not supplied): Winograd F(2x2, 3x3) transformed copies of the 3x3 layers, U = G g G^T
     * packed as a 4x4-tap kernel (16, Cin/4, npad, 4) -- tf_raft_amd/packing.py pack_conv_winograd */
    raft_conv_weights convc2_w, convf2_w, conv_w, fh1_mask0_w;
    /* optional: 1-D Winograd F(2, 5) transformed copies of gru_zr{1,2} / gru_q{1,2}: U = G' g packed as a 6-tap
     * kernel (6, Cin/4, npad, 4) -- tf_raft_amd/packing.py pack_conv_winograd1d */
    raft_conv_weights gru_zr1_w, gru_q1_w, gru_zr2_w, gru_q2_w;
    /* optional: Winograd F(2x2, 3x3) copy of flow_head.conv1 ALONE (3,3,128,256), for raft_iterate_basic_final_f32 */
    raft_conv_weights fh1_w;
    /* optional: 1-D Winograd F(4, 5) transformed copies of gru_zr{1,2} / gru_q{1,2}, packed as 8-tap kernels
     * (8, Cin/4, npad, 4) -- pack_conv_winograd1d(..., m=4); preferred over the F(2, 5) copies when supplied */
    raft_conv_weights gru_zr1_w4, gru_q1_w4, gru_zr2_w4, gru_q2_w4;
    /* optional: convc1 repacked for raft_lookup_convc1_f32 (K = 4 levels x 84: each level's 81 channels + 3 zero rows;
     * (84, 256, 4) -- packing.py pack_convc1_fused).  When supplied, the raft_iterate_basic_* loops on a STORED volume
     * run the lookup fused into convc1 (RAFT_LOOKUP_FUSED = 0 keeps the two kernels). */
    raft_conv_weights convc1_f;
    /* optional: 1-D Winograd F(4, 5) transformed copies of gru_ctx1 / gru_ctx2 (8-tap kernels, with the biases) */
    raft_conv_weights gru_ctx1_w4, gru_ctx2_w4;
    /* optional: Winograd F(4x4, 3x3) transformed copies of convc2 / conv / fh1_mask0 and of flow_head.conv1 alone
     * (packing.py pack_conv_winograd4); preferred over the F(2x2, 3x3) copies where RAFT_CONV_WINO4 has the layer's bit */
    raft_conv_weights convc2_w44, conv_w44, fh1_mask0_w44, fh1_w44;
    /* optional: the same for convf2 (RAFT_CONV_WINO4 bit 2) */
    raft_conv_weights convf2_w44;
} raft_basic_update_weights;

/* Device state of the recurrent loop (all caller-owned, (B*h*w) pixels, NHWC):
 *   net    (M,128)  hidden state h, updated in place
 *   x      (M,256)  GRU input [inp(128) | motion(126) | flow(2)]; inp is written once by
 *                   raft_prepare_state_f32, the rest every iteration
 *   corr   (M,352)  lookup output, 324 used + 28 zero pad channels
 *   coords1(M,2), flow (M,2), delta (M,2), mask (M,576)
 *   ws              scratch, raft_update_workspace_floats() floats
 *   ctx    (M,768)  loop-invariant GRU pre-activation terms [z1 | r1 | q1 | z2 | r2 | q2] of `inp`,
 *                   written by raft_gru_context_f32 (BasicUpdateBlock only; SmallRAFT ignores it) */
typedef struct raft_state {
    float *net, *x, *corr, *coords1, *flow, *delta, *mask, *ws, *ctx;
} raft_state;

int64_t raft_update_workspace_floats(int B, int h, int w);

/* model.py:84-89 -- net = tanh(cnet[..., :128]); inp = relu(cnet[..., 128:]); coords1 = grid;
 * flow = 0; zero the pad channels of corr.  cnet: (B, h, w, 256). */
int raft_prepare_state_f32(const float *cnet, int B, int h, int w, const raft_state *st,
                           void *stream);

/* conv{z,r,q}{1,2} restricted to the `inp` rows of the GRU input, plus their biases -> st->ctx.
 * Must run after raft_prepare_state_f32 (or any other write of x[:, 0:128]) and before
 * raft_update_basic_f32 / raft_iterate_basic_*; reference update.py:51-67 evaluates these rows inside
 * every SepConvGRU call, model.py:86 shows `inp` is fixed for the whole loop. */
int raft_gru_context_f32(const raft_basic_update_weights *wts, int B, int h, int w,
                         const raft_state *st, void *stream);

/* One BasicUpdateBlock call + the coordinate update (reference update.py:143-153 and
 * model.py:97-102): reads st->corr, st->flow, st->net, st->x, st->ctx; writes st->net, st->mask
 * (already scaled by 0.25), st->delta, st->coords1 (+= delta), st->flow (coords1 - coords0). */
int raft_update_basic_f32(const raft_basic_update_weights *wts, int B, int h, int w,
                          const raft_state *st, void *stream);

/* The whole prediction loop of RAFT.call (reference model.py:91-109), `iters` times:
 * lookup -> update -> coords1 += delta -> convex upsample.  flow_up: (iters, B, 8h, 8w, 2);
 * prediction i is written to flow_up + i * B*8h*8w*2. */
int raft_iterate_basic_f32(const raft_basic_update_weights *wts, const float *pyr,
                           const int64_t *level_offsets, int B, int h, int w, int iters,
                           const raft_state *st, float *flow_up, void *stream);

/* Caller-owned context of the three-stream loops below: the four cross-stream events of the schedule and a cache of up
 * to four instantiated hipGraphs of whole prediction loops.  Create once per model / thread on the device the loops run
 * on (the only entry point that allocates), pass to every raft_iterate_basic_{overlap,ondemand,final}_f32 call, destroy
 * after the last loop has drained.  Not thread-safe: one context per launching thread. */
typedef struct raft_loop_ctx raft_loop_ctx;
int raft_loop_ctx_create(raft_loop_ctx **ctx);
int raft_loop_ctx_destroy(raft_loop_ctx *ctx);

/* The same loop scheduled on three streams: the flow branch (convf1, convf2) and the mask branch (mask2,
 * convex upsample) run on the caller-owned side streams aux0 / aux1 next to the main chain on `stream`,
 * ordered by events; everything is joined back into `stream` before the call returns (it still only
 * enqueues).  Results are identical to raft_iterate_basic_f32.
 * aux0 == aux1 == stream selects the SINGLE-STREAM schedule (the launches of raft_iterate_basic_f32, no events) for
 * this and the two entry points below: what a caller that keeps several loops in flight on streams of their own
 * (tf_raft_amd/model.py, lanes of the pipelined forward) runs on each.  Any other coincidence of the three streams is
 * RAFT_E_UNSUPPORTED.
 * With RAFT_LOOP_GRAPH on (off by default) the first call with a given set of arguments (pointers, sizes,
 * streams) captures these launches into a hipGraph kept in `ctx`; later calls with the same arguments replay it with ONE
 * hipGraphLaunch on `stream` -- the reference's canonical (1,448,512,3) call is bound by the host's ~350 launches + ~100
 * event operations otherwise.  A non-NULL `stream` is required for that (the legacy default stream cannot be captured;
 * the plain launches are used on it). */
int raft_iterate_basic_overlap_f32(const raft_basic_update_weights *wts, const float *pyr,
                                   const int64_t *level_offsets, int B, int h, int w, int iters,
                                   const raft_state *st, float *flow_up, void *stream, void *aux0, void *aux1,
                                   raft_loop_ctx *ctx);

/* The three-stream loop with the volume-free correlation (BASELINE config 4, no reference code: README.md:109):
 * fmap1 (B, h, w, C) and fmap2_pyr (raft_fmap_pyramid_f32) replace the stored pyramid. */
int raft_iterate_basic_ondemand_f32(const raft_basic_update_weights *wts, const float *fmap1,
                                    const float *fmap2_pyr, int C, int B, int h, int w, int iters,
                                    const raft_state *st, float *flow_up, void *stream, void *aux0, void *aux1,
                                    raft_loop_ctx *ctx);

/* The three-stream loop for callers that want flow_predictions[-1] only (reference model.py:160-166, predict_step):
 * mask head + convex upsampling run in the last iteration only; flow_up_last: (B, 8h, 8w, 2).  The recurrence is launch
 * for launch that of raft_iterate_basic_overlap_f32, so the result equals its last prediction.  Needs wts->fh1_w. */
int raft_iterate_basic_final_f32(const raft_basic_update_weights *wts, const float *pyr,
                                 const int64_t *level_offsets, int B, int h, int w, int iters,
                                 const raft_state *st, float *flow_up_last, void *stream, void *aux0, void *aux1,
                                 raft_loop_ctx *ctx);

/* Profiling twin of raft_iterate_basic_f32 (bench.py only): the same launches with a HIP event
 * recorded on `stream` after every kernel; synchronises the stream and accumulates the elapsed
 * milliseconds of each stage over all iterations into the HOST array stage_ms.  Stage order:
 * lookup, convc1, convc2, convf1, convf2, conv, gru_zr1, gru_q1, gru_zr2, gru_q2, fh1_mask0, fh2,
 * mask2, upsample. */
#define RAFT_BASIC_STAGES 14
int raft_iterate_basic_timed_f32(const raft_basic_update_weights *wts, const float *pyr,
                                 const int64_t *level_offsets, int B, int h, int w, int iters,
                                 const raft_state *st, float *flow_up, void *stream,
                                 float *stage_ms /* host, [RAFT_BASIC_STAGES] */);

/* ------------------------------------------------------------------ encoders */

/* BasicEncoder / SmallEncoder forward (reference extractor.py:88-130, 133-175; ResBlock 19-49;
 * Normalization 6-16), inference mode.  Channel widths: stem c0, layer1..3 c1..c3 (multiples of 32),
 * output cout.  norm:
 *   RAFT_NORM_NONE      no normalisation (SmallRAFT cnet)
 *   RAFT_NORM_INSTANCE  tfa InstanceNormalization, eps 1e-3, biased variance, affine in_gamma / in_beta
 *   RAFT_NORM_FOLDED    Keras BatchNormalization in inference mode, folded into kernels and biases by the
 *                       host packer (tf_raft_amd/packing.py): the device sees plain convolutions
 * Convolution kernels use the packed layout above; the 7x7 stem is packed as 7 K-chunks (one per kernel
 * row) of k = kx * 4 + ch over the 4-channel padded image.  block[i][2] (down-sampling 1x1) has wp == NULL
 * for stride-1 blocks.  in_gamma / in_beta index: 0 = stem norm1, 1 + 3*i + {0, 1, 2} = block i norm1,
 * norm2, downsample norm. */
enum { RAFT_NORM_NONE = 0, RAFT_NORM_INSTANCE = 1, RAFT_NORM_FOLDED = 2 };

typedef struct raft_encoder_weights {
    int c0, c1, c2, c3, cout, norm;
    raft_conv_weights conv1;
    raft_conv_weights block[6][3];
    raft_conv_weights conv2;
    const float *in_gamma[19];
    const float *in_beta[19];
    /* optional (wp == NULL: absent): Winograd F(2x2, 3x3) transformed copies of block[i][0] (when it has
     * stride 1) and block[i][1], packed as 4x4-tap kernels (see raft_conv2d_winograd_f32) */
    raft_conv_weights block_w[6][2];
    /* optional: Winograd F(4x4, 3x3) transformed copies of the same layers in the consumption order of
     * raft_conv2d_winograd4_f32 (used for the stages RAFT_ENC_WINO4 selects) */
    raft_conv_weights block_w44[6][2];
} raft_encoder_weights;

int64_t raft_encoder_workspace_floats(const raft_encoder_weights *w, int n, int H, int W);

/* images: (n, H, W, 3); out: (n, ceil(H/8), ceil(W/8), cout).  input_affine != 0 applies the model's
 * 2 * (image / 255) - 1 (reference model.py:70-71) while the image is staged. */
int raft_encoder_f32(const raft_encoder_weights *w, const float *images, int n, int H, int W,
                     int input_affine, float *out, float *workspace, void *stream);
/* The feature encoder's call on [image1, image2] (reference extractor.py:114-116 concatenates along the batch, 128-130 splits
 * again): images_a / images_b: (n_each, H, W, 3) each, staged from where they lie; out: (2 * n_each, ...), the first n_each
 * maps belong to images_a.  Workspace = raft_encoder_workspace_floats(w, 2 * n_each, H, W). */
int raft_encoder_pair_f32(const raft_encoder_weights *w, const float *images_a, const float *images_b, int n_each, int H, int W,
                          int input_affine, float *out, float *workspace, void *stream);

/* ------------------------------------------------------------------ SmallRAFT update block */

/* SmallUpdateBlock weights (reference update.py:109-125): z and r of the 3x3 ConvGRU fused along N
 * (npad 192 = [z 96 | r 96]); convf1 (7,7,2,64) and fh2 = flow_head.conv2 (3,3,128,2) in Keras layout.
 * State: net (M,96); x (M,160) = [inp 64 | motion 80 | flow 2 | 14 zero pad]; corr (M,224) = 196 used
 * + 28 zero pad; mask is unused (SmallUpdateBlock returns None). */
typedef struct raft_small_update_weights {
    raft_conv_weights convc1, convf1, convf2, conv, gru_zr, gru_q, fh1, fh2;
    /* optional: Winograd F(2x2, 3x3) transformed copies of the 3x3 layers (see raft_basic_update_weights) */
    raft_conv_weights conv_w, gru_zr_w, gru_q_w, fh1_w;
} raft_small_update_weights;

int64_t raft_small_update_workspace_floats(int B, int h, int w);
/* cnet: (B, h, w, 160) -> net = tanh(cnet[..., :96]), inp = relu(cnet[..., 96:]).  model.py:206-211 */
int raft_prepare_state_small_f32(const float *cnet, int B, int h, int w, const raft_state *st,
                                 void *stream);
/* One SmallUpdateBlock call + coordinate update (update.py:118-125, model.py:216-220). */
int raft_update_small_f32(const raft_small_update_weights *wts, int B, int h, int w,
                          const raft_state *st, void *stream);
/* The prediction loop of SmallRAFT.call (model.py:213-226): lookup (radius 3) -> update -> upflow8. */
int raft_iterate_small_f32(const raft_small_update_weights *wts, const float *pyr,
                           const int64_t *level_offsets, int B, int h, int w, int iters,
                           const raft_state *st, float *flow_up, void *stream);

/* ------------------------------------------------------------------ evaluation metrics / loss */

/* Reductions over flow predictions (reference tf_raft/losses/losses.py; the caller side of the forward pass:
 * model.py:146-158 test_step, 126-144 train_step).  flow_gt / predictions: (npix, 2) fp32 with npix = B*H*W;
 * valid: npix bytes (0 = invalid).  valid' = valid & (|flow_gt|_2 < max_flow).  Deterministic (no atomics);
 * `workspace` = raft_metrics_workspace_doubles() doubles, 8-byte aligned; results are written to device memory. */
int64_t raft_metrics_workspace_doubles(void);
/* end_point_error (losses.py:24-43): out5 = {mean EPE over valid', rate EPE < 1, rate EPE < 3, rate EPE < 5,
 * number of valid' pixels}; the means are NaN when no pixel is valid' (as tf.reduce_mean of an empty tensor). */
int raft_flow_metrics_f32(const float *flow_gt, const unsigned char *valid, const float *flow_pred,
                          int64_t npix, float max_flow, float *out5, double *workspace, void *stream);
/* sequence_loss (losses.py:4-21): loss = sum_i gamma^(n-i-1) * mean over ALL 2*npix elements of valid' * |pred_i - gt|.
 * Prediction i starts at preds + i * pred_stride floats (the flow_up buffer of raft_iterate_*: stride 2*npix);
 * n_predictions <= 64. */
int raft_sequence_loss_f32(const float *flow_gt, const unsigned char *valid, const float *preds,
                           int64_t pred_stride, int n_predictions, int64_t npix, double gamma, float max_flow,
                           float *loss_out, double *workspace, void *stream);

/* ------------------------------------------------------------------ training step, first slice (backward kernels) */

/* The reference's train_step (model.py:126-144) differentiates the forward pass with tf.GradientTape.  These entry points
 * are the backward of the path-specific ops, deterministic (no atomics).  BASELINE config 5; see DESIGN.md section 9 for
 * what is and is not built yet. */

/* d sequence_loss / d prediction_i for all n predictions (losses.py:4-21):
 *   d_preds[i][e] = upstream * gamma^(n-i-1) * valid'[pixel(e)] * sign(pred_i[e] - flow_gt[e]) / (2 * npix),  sign(0) = 0.
 * d_preds has the layout of preds (prediction i at d_preds + i * pred_stride). */
int raft_sequence_loss_grad_f32(const float *flow_gt, const unsigned char *valid, const float *preds,
                                int64_t pred_stride, int n_predictions, int64_t npix, double gamma, float max_flow,
                                float upstream, float *d_preds, void *stream);

/* Backward of raft_corr_lookup_f32 (CorrBlock.retrieve + bilinear_sampler, corr.py:116-152, 28-69), with TensorFlow's
 * gradient conventions: floor / ceil contribute nothing, clip_by_value passes the gradient on [0, size - 1].
 * d_out: (B, h, w, ld_out) upstream gradient of the lookup output (first levels*(2r+1)^2 channels used).
 * d_coords: (B, h, w, 2), overwritten.  d_pyr: gradient w.r.t. the correlation pyramid in the layout of `pyr`
 * (raft_corr_pyramid_layout), ACCUMULATED (+=) so that the iterations of the loop add up -- zero it before the first
 * call; NULL skips it. */
int raft_corr_lookup_backward_f32(const float *pyr, const int64_t *level_offsets, const float *coords,
                                  const float *d_out, int ld_out, int B, int h, int w, int levels, int radius,
                                  float *d_coords, float *d_pyr, void *stream);

/* dx = dy where y > 0 else 0 (the relu between the convolutions of the update block), n elements. */
int raft_relu_backward_f32(const float *y, const float *dy, float *dx, int64_t n, void *stream);

/* Training-time weight packing on the device (train_step's master weights change every step: reference model.py:134-136), one
 * launch per (kernel, use).  kernel: (kh, kw, cin, cout) Keras layout.  dgrad != 0 packs the kernel of the INPUT gradient
 * K'[ky][kx][co][ci] = K[kh-1-ky][kw-1-kx][ci][co] instead.  mode 0: the kh * kw taps as they are; mode 1 (3x3): the 16 taps
 * G g G^T of F(2x2, 3x3), g = (4, 3) row-major doubles; mode 2 (1x5 / 5x1): the gt taps G' g of F(gt - 4, 5), g = (gt, 5) doubles;
 * float64 accumulation, one rounding.  wp: [taps][kpad / 4][npad][4] as raft_conv2d_f32 / raft_conv2d_winograd_f32 /
 * raft_conv1d_winograd{,4}_f32 consume it (zero beyond the kernel's channels); bias_out: npad floats (bias, or zeros when bias is
 * NULL or dgrad is set). */
int raft_pack_train_conv_f32(const float *kernel, const float *bias, int kh, int kw, int cin, int cout, int dgrad, int mode,
                             const double *g, int gt, int kpad, int npad, float *wp, float *bias_out, void *stream);

/* Keras Conv2D (stride 1, 'same'; kh x kw in {1x1, 3x3, 1x5, 5x1}) backward w.r.t. kernel and bias:
 *   d_kernel[ky][kx][ci][co] = sum_pixels x[b, y + ky - (kh-1)/2, x + kx - (kw-1)/2, ci] * dy[b, y, x, co]   (Keras layout)
 *   d_bias[co] = sum_pixels dy[b, y, x, co]                                                       (NULL skips it)
 * x: (B, H, W, ldx) with cin channels used, dy: (B, H, W, ldy) with cout channels used; cin, cout, ldx, ldy multiples
 * of 4.  fp32 MFMA over pixel slices + an ordered second-stage sum (deterministic); workspace:
 * raft_conv2d_wgrad_workspace_floats() floats.  The gradient w.r.t. the INPUT is raft_conv2d_f32 of dy with the flipped,
 * transposed kernel (tf_raft_amd/packing.py pack_conv_dgrad). */
int64_t raft_conv2d_wgrad_workspace_floats(int cin, int cout, int B, int H, int W, int kh, int kw);
int raft_conv2d_wgrad_f32(const float *x, int ldx, int cin, const float *dy, int ldy, int cout, int B, int H, int W,
                          int kh, int kw, float *d_kernel, float *d_bias, float *workspace, void *stream);
/* The same gradient summed over nseg (1..32) pairs (xs[i], dys[i]) of identical geometry in ONE pixel reduction; xs / dys are
 * HOST arrays of device pointers (copied into the kernel arguments).  The iterations of a training step share the update
 * block's weights (reference model.py:91-109): one launch over all of them replaces nseg launches, nseg second-stage sums and
 * nseg - 1 accumulations per layer.  Workspace: raft_conv2d_wgrad_workspace_floats(cin, cout, B * nseg, H, W, kh, kw). */
int raft_conv2d_wgrad_multi_f32(const float *const *xs, const float *const *dys, int nseg, int ldx, int cin, int ldy, int cout,
                                int B, int H, int W, int kh, int kw, float *d_kernel, float *d_bias, float *workspace, void *stream);

/* ---- second slice: everything one BasicUpdateBlock call needs (tf_raft_amd/grad.py basic_update_block_backward) */

/* relu(Conv2D(7, 7, Cin = 2 -> cout)) of the flow (update.py:93, 76): kernel (98, cout) = Keras (7,7,2,cout) flattened,
 * cout in {64, 128}; out (B, H, W, ldo).  The inference loop runs this kernel inside raft_update_*_f32. */
int raft_conv7x7_c2_f32(const float *flow, const float *kernel, const float *bias, int cout, int B, int H, int W,
                        float *out, int ldo, void *stream);
/* Its backward for an upstream gradient dy that already carries the relu mask: d_flow (B, H, W, 2), d_kernel (98, cout),
 * d_bias (cout).  Deterministic; workspace raft_conv7x7_c2_wgrad_workspace_floats(cout) floats. */
int64_t raft_conv7x7_c2_wgrad_workspace_floats(int cout);
int raft_conv7x7_c2_backward_f32(const float *flow, const float *dy, int ldy, const float *kernel, int cout, int B, int H,
                                 int W, float *d_flow, float *d_kernel, float *d_bias, float *workspace, void *stream);

/* SepConvGRU gates (update.py:51-67) as separate kernels, so that the training forward keeps z, r, q:
 *   zr: a_zr (M, 2C) = [convz | convr] pre-activations -> z = sigmoid, r = sigmoid, rh = r * h        (all (M, C))
 *   q : a_q (n) -> q = tanh(a_q), h_new = (1 - z) h + z q
 * and their backward:
 *   q_backward: d_h_new -> dz_pre = d (q - h) z (1 - z), dq_pre = d z (1 - q^2), dh = d (1 - z)      (dh overwritten)
 *   r_backward: d_rh -> dr_pre = d h r (1 - r), dh += d r                                           (dh accumulated) */
int raft_gru_gate_zr_f32(const float *a_zr, const float *h, int C, int64_t M, float *z, float *r, float *rh, void *stream);
int raft_gru_gate_q_f32(const float *a_q, const float *z, const float *h, int64_t n, float *q, float *h_new, void *stream);
int raft_gru_gate_q_backward_f32(const float *d_h_new, const float *z, const float *q, const float *h, int64_t n,
                                 float *dz_pre, float *dq_pre, float *dh, void *stream);
int raft_gru_gate_r_backward_f32(const float *d_rh, const float *r, const float *h, int64_t n, float *dr_pre, float *dh,
                                 void *stream);
/* out = alpha * a + beta * b (b may be NULL): gradient accumulation where two branches meet, the 0.25 of the mask head. */
int raft_axpby_f32(float alpha, const float *a, float beta, const float *b, float *out, int64_t n, void *stream);

/* Backward of raft_upsample_convex_f32 (RAFT.upsample_flow, model.py:39-66): d_up (B, 8h, 8w, 2) -> d_flow (B, h, w, 2)
 * and d_mask (B, h, w, 576), both overwritten.  Deterministic (the scatter into the 3x3 neighbours is a gather over a
 * per-pixel, per-tap scratch); workspace raft_upsample_convex_backward_workspace_floats() floats. */
int64_t raft_upsample_convex_backward_workspace_floats(int B, int h, int w);
int raft_upsample_convex_backward_f32(const float *flow, const float *mask, const float *d_up, int B, int h, int w,
                                      float *d_flow, float *d_mask, float *workspace, void *stream);

/* ---- optimizer side of train_step (model.py:131-136): tf.clip_by_global_norm + tfa.optimizers.AdamW (tfa 0.11.1) */

/* *out (+)= sum of x[i]^2 in float64, deterministic; accumulate != 0 adds to the value already in *out (the global norm
 * runs over every gradient tensor).  workspace: raft_sumsq_workspace_doubles() doubles. */
int64_t raft_sumsq_workspace_doubles(void);
int raft_sumsq_f32(const float *x, int64_t n, int accumulate, double *out, double *workspace, void *stream);
/* One AdamW step on one tensor, tensorflow-addons 0.11.1 semantics: var -= weight_decay * var (decoupled, not scaled by
 * the learning rate); g = grad * clip_norm / max(sqrt(*global_norm_sq), clip_norm) (global_norm_sq NULL: no clipping);
 * m, v Adam moments; var -= lr_t * m / (sqrt(v) + epsilon) with lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) from the caller. */
int raft_adamw_step_f32(float *var, const float *grad, float *m, float *v, int64_t n, float lr_t, float beta1, float beta2,
                        float epsilon, float weight_decay, const double *global_norm_sq, float clip_norm, void *stream);
/* The same two steps over MANY tensors per launch (host arrays of device pointers and element counts, copied into the kernel
 * arguments in chunks of 64): a RAFT model has 154 trainable tensors, i.e. 154 x 3 launches per training step otherwise.
 * raft_sumsq_multi_f32: *out = sum over all tensors of the sum of squares (float64, ordered: deterministic), workspace
 * raft_sumsq_multi_workspace_doubles(count) doubles.  raft_adamw_step_multi_f32: raft_adamw_step_f32's arithmetic per element. */
int64_t raft_sumsq_multi_workspace_doubles(int count);
int raft_sumsq_multi_f32(const float *const *xs, const int64_t *ns, int count, double *out, double *workspace, void *stream);
int raft_adamw_step_multi_f32(float *const *vars, const float *const *grads, float *const *ms, float *const *vs, const int64_t *ns,
                              int count, float lr_t, float beta1, float beta2, float epsilon, float weight_decay,
                              const double *global_norm_sq, float clip_norm, void *stream);

/* ---- backward of the volume build and of the state preparation */

/* Generic strided batched GEMM on fp32 MFMA: C[b][m][n] = alpha * sum_k A(b,m,k) B(b,k,n) + beta * C[b][m][n] with
 * A(b,m,k) = a[b*sab + m*sam + k*sak], B(b,k,n) = b[b*sbb + k*sbk + n*sbn], C row-major (ldc) with batch stride scb. */
int raft_gemm_f32(const float *a, int64_t sab, int64_t sam, int64_t sak, const float *b, int64_t sbb, int64_t sbk, int64_t sbn,
                  float *c, int64_t scb, int ldc, int batch, int M, int N, int K, float alpha, float beta, void *stream);
/* Backward of raft_corr_build_f32 (CorrBlock.__init__, corr.py:100-114, 154-162): d_pyr (layout of the pyramid) ->
 * d_fmap1, d_fmap2 (B, h, w, C), overwritten.  fmap2_pyr: the pooled-fmap2 workspace the forward filled; workspace: as
 * many floats again (raft_corr_build_workspace_floats). */
int raft_corr_build_backward_f32(const float *fmap1, const float *fmap2_pyr, const float *d_pyr, const int64_t *level_offsets,
                                 int B, int h, int w, int C, int levels, float *d_fmap1, float *d_fmap2, float *workspace,
                                 void *stream);
/* model.py:84-86 backward: d_cnet (M, hdim + cdim) = [d_net0 * (1 - net0^2) | d_inp where inp > 0]. */
int raft_prepare_state_backward_f32(const float *net0, const float *inp, const float *d_net0, const float *d_inp, int hdim,
                                    int cdim, int64_t M, float *d_cnet, void *stream);

/* ---- normalisation layers of the encoders in training form (extractor.py:6-16) */

/* x viewed as (G groups, P pixels, C channels): tfa InstanceNormalization = (G = B, P = H*W), Keras BatchNormalization with
 * batch statistics = (G = 1, P = B*H*W); biased variance, y = (x - mean) * rstd * gamma + beta [relu], rstd = 1/sqrt(var + eps).
 * mean, rstd (and var when non-NULL): (G, C), kept for the backward / the moving statistics.  Deterministic float64 sums.
 * workspace: raft_norm_workspace_doubles(G, C) doubles. */
int64_t raft_norm_workspace_doubles(int G, int C);
int raft_norm_forward_f32(const float *x, int G, int64_t P, int C, const float *gamma, const float *beta, float eps, int relu,
                          float *y, float *mean, float *rstd, float *var, double *workspace, void *stream);
/* dy is the gradient at the norm's linear output (apply raft_relu_backward_f32 first when relu was fused). */
int raft_norm_backward_f32(const float *x, const float *dy, const float *mean, const float *rstd, const float *gamma, int G,
                           int64_t P, int C, float *dx, float *dgamma, float *dbeta, double *workspace, void *stream);
/* out = relu(alpha * a + beta * b): the residual join of a ResBlock (extractor.py:49). */
int raft_axpby_relu_f32(float alpha, const float *a, float beta, const float *b, float *out, int64_t n, void *stream);

/* bf16 storage of the training tape (BASELINE configs[4]: "bf16" = bf16 storage, fp32 arithmetic): round-to-nearest-even
 * narrowing of n floats into n 16-bit words and the exact widening back. */
int raft_f32_to_bf16(const float *x, void *y, int64_t n, void *stream);
int raft_bf16_to_f32(const void *x, float *y, int64_t n, void *stream);

/* Keras Dropout in training mode (reference extractor.py:109-111, 127-128): y = x * keep / (1 - rate) with keep ~
 * Bernoulli(1 - rate) from a counter-based generator of (seed, element index); mask[i] = keep (bytes).  The backward is
 * dx = dy * mask / (1 - rate).  TensorFlow's random stream is not reproduced: parity of this layer is distributional. */
int raft_dropout_f32(const float *x, int64_t n, float rate, uint64_t seed, float *y, unsigned char *mask, void *stream);
int raft_dropout_backward_f32(const float *dy, const unsigned char *mask, int64_t n, float rate, float *dx, void *stream);

/* Backward of raft_upflow8_f32 (corr.py:93-96): d_up (B, 8h, 8w, 2) -> d_flow (B, h, w, 2), a deterministic gather. */
int raft_upflow8_backward_f32(const float *d_up, int B, int h, int w, float *d_flow, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RAFT_HIP_H_ */
