// Shared helpers for the gfx950 RAFT kernels (internal; the public ABI is include/raft_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/raft_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RAFT_REQUIRE_PTR(p) \
    do {                     \
        if ((p) == nullptr) return RAFT_E_NULL; \
    } while (0)

#define RAFT_REQUIRE(cond, code) \
    do {                          \
        if (!(cond)) return (code); \
    } while (0)

#define RAFT_TRY(expr)              \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != RAFT_OK) return rc__; \
    } while (0)

// Launch-error check: hipError_t values are positive, RAFT_E_* negative, RAFT_OK == hipSuccess == 0.
static inline int raft_launch_status() { return (int)hipGetLastError(); }

static inline bool raft_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

static inline int raft_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Layout of one per-query correlation map (and of the rows of the pooled fmap2 pyramid): 4 x 8 tiles of 32
// floats = one 128-byte line each, tiles row-major, maps padded to whole tiles.  The (2r+2)^2 lookup footprint
// then touches ~6.9 lines of a large map instead of ~12.8 with row-major rows (the lookup is
// bound by the lines it pulls from HBM, not by the 40 useful bytes per footprint row).
// The tile shape is a compile-time constant of the library (2^RAFT_TILE_H_LOG2 x 2^RAFT_TILE_W_LOG2; the product path and its
// Python un-tiling use 4 x 8): tools/ablate/lookup_layout.sh rebuilds the volume build + lookup with other shapes (8 x 4,
// 2 x 16, 4 x 4, 8 x 8, row-major 1 x 8) to measure the lines each pulls (profiles/r10*_lookup_layouts.txt).
#ifndef RAFT_TILE_H_LOG2
#define RAFT_TILE_H_LOG2 2
#endif
#ifndef RAFT_TILE_W_LOG2
#define RAFT_TILE_W_LOG2 3
#endif
constexpr int RAFT_TILE_H = 1 << RAFT_TILE_H_LOG2, RAFT_TILE_W = 1 << RAFT_TILE_W_LOG2, RAFT_TILE_FLOATS = RAFT_TILE_H * RAFT_TILE_W;
__host__ __device__ inline int raft_tiles_x(int w) { return (w + RAFT_TILE_W - 1) / RAFT_TILE_W; }
__host__ __device__ inline int raft_tiles_y(int h) { return (h + RAFT_TILE_H - 1) / RAFT_TILE_H; }
// whole tiles, and whole 128-byte lines (32 floats) for tiles smaller than a line: every map starts on a line
__host__ __device__ inline int raft_map_floats(int h, int w) { return (raft_tiles_y(h) * raft_tiles_x(w) * RAFT_TILE_FLOATS + 31) & ~31; }
__host__ __device__ inline int raft_tiled_index(int y, int x, int tiles_x) {
    return (((y >> RAFT_TILE_H_LOG2) * tiles_x + (x >> RAFT_TILE_W_LOG2)) << (RAFT_TILE_H_LOG2 + RAFT_TILE_W_LOG2)) +
           ((y & (RAFT_TILE_H - 1)) << RAFT_TILE_W_LOG2) + (x & (RAFT_TILE_W - 1));
}
// inverse: float n of a tiled map -> (y, x); positions in the padding of the last tiles (or beyond them) come out >= h / w
__host__ __device__ inline void raft_untiled_yx(int n, int tiles_x, int *y, int *x) {
    const int t = n >> (RAFT_TILE_H_LOG2 + RAFT_TILE_W_LOG2);
    *y = (t / tiles_x) * RAFT_TILE_H + ((n >> RAFT_TILE_W_LOG2) & (RAFT_TILE_H - 1));
    *x = (t % tiles_x) * RAFT_TILE_W + (n & (RAFT_TILE_W - 1));
}

// Tuning switches (include/raft_hip.h: raft_set_option).  Process-global, initialised ONCE from the environment when the
// library is loaded and changed only through raft_set_option afterwards: the launch path reads an atomic int, it never
// calls getenv.  raft_opt(id, dflt) = the switch's value, or dflt while it is unset.
enum RaftOptionId {
    RAFT_OPT_CONV_WINO, RAFT_OPT_SMALL_WINO, RAFT_OPT_GRU_WINO, RAFT_OPT_GRU_WINO4, RAFT_OPT_WINO_TNW, RAFT_OPT_WINO_SB,
    RAFT_OPT_WINO_CK, RAFT_OPT_WINO1D_TM,
    RAFT_OPT_LOOKUP_FUSED, RAFT_OPT_ONDEMAND_BLOCK, RAFT_OPT_ENC_WINO, RAFT_OPT_LOOP_GRAPH,
    RAFT_OPT_WINO_KS, RAFT_OPT_CONV_WINO4, RAFT_OPT_WINO4_KS, RAFT_OPT_MASK_FUSED, RAFT_OPT_ENC_WINO4, RAFT_OPT_CONVC2_KS, RAFT_OPT_CONVF2_KS,
    RAFT_OPT_EVENT_FENCE, RAFT_OPT_CORR_XCD, RAFT_OPT_CORR_POOL,
    RAFT_OPT_COUNT
};
int raft_opt(int id, int dflt);
// raft_set_thread_concurrency (include/raft_hip.h): how many independent launch sequences of about this size share the device
// with the calling thread's launches (>= 1).  The launchers' "does this grid fill the chip?" rules count a grid n times.
int raft_concurrency();
bool raft_opt_is_set(int id);
int raft_opt_generation();   // bumped by every raft_set_option call
// RAFT_CONV_TILE ("<code>" or "<npad>:<taps>:<code>,..."): the tile code forced for a convolution, or -1
int raft_opt_conv_tile(int npad, int taps, bool (*valid)(int code, int npad));

struct PyramidGeom {
    int64_t off[RAFT_MAX_LEVELS];   // float offset of each level of the corr pyramid
    int lh[RAFT_MAX_LEVELS];
    int lw[RAFT_MAX_LEVELS];
    int tx[RAFT_MAX_LEVELS];        // tiles per tile-row of a level's map
    int map[RAFT_MAX_LEVELS];       // floats per query map (padded to whole tiles)
    int levels;
};

static inline int raft_make_geom(int h, int w, int levels, const int64_t *level_offsets, PyramidGeom *g) {
    if (levels < 1 || levels > RAFT_MAX_LEVELS) return RAFT_E_UNSUPPORTED;
    g->levels = levels;
    int ch = h, cw = w;
    for (int l = 0; l < RAFT_MAX_LEVELS; ++l) {
        g->lh[l] = g->lw[l] = g->tx[l] = 1;
        g->map[l] = raft_map_floats(1, 1);
        g->off[l] = 0;
    }
    for (int l = 0; l < levels; ++l) {
        if (ch < 1 || cw < 1) return RAFT_E_SHAPE;   // pooled away: the reference would fail here too
        g->lh[l] = ch;
        g->lw[l] = cw;
        g->tx[l] = raft_tiles_x(cw);
        g->map[l] = raft_map_floats(ch, cw);
        g->off[l] = level_offsets ? level_offsets[l] : 0;
        ch /= 2;
        cw /= 2;
    }
    return RAFT_OK;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_barrier + fence, and the release fence
// waits for EVERY outstanding memory operation of the wave (s_waitcnt vmcnt(0)): the global loads the K loops keep in
// flight across their barriers (next stage's halo tile, weight fragments a few taps ahead) would be drained at each
// stage and re-issued behind the barrier, exposing an L2 round trip per stage.  Nothing in these kernels communicates
// through global memory inside a workgroup, so the barrier only has to wait for the wave's own LDS operations
// (lgkmcnt) -- the "memory" clobber keeps the compiler from moving LDS accesses across it.
#ifndef RAFT_LDS_BARRIER
#define RAFT_LDS_BARRIER 1
#endif
__device__ __forceinline__ void raft_barrier_lds() {
#if RAFT_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

