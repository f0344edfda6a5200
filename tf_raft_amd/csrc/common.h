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

struct PyramidGeom {
    int64_t off[RAFT_MAX_LEVELS];   // float offset of each level of the corr pyramid
    int lh[RAFT_MAX_LEVELS];
    int lw[RAFT_MAX_LEVELS];
    int levels;
};

static inline int raft_make_geom(int h, int w, int levels, const int64_t *level_offsets, PyramidGeom *g) {
    if (levels < 1 || levels > RAFT_MAX_LEVELS) return RAFT_E_UNSUPPORTED;
    g->levels = levels;
    int ch = h, cw = w;
    for (int l = 0; l < levels; ++l) {
        if (ch < 1 || cw < 1) return RAFT_E_SHAPE;   // pooled away: the reference would fail here too
        g->lh[l] = ch;
        g->lw[l] = cw;
        g->off[l] = level_offsets ? level_offsets[l] : 0;
        ch /= 2;
        cw /= 2;
    }
    return RAFT_OK;
}
