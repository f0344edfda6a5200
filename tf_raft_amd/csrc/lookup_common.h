// Shared pieces of the volume and on-demand correlation lookups (internal).
#pragma once
#include "common.h"

// Axis helper, bit-for-bit the reference's arithmetic (corr.py:41-48, 57-60): clamp, floor, ceil,
// weights ceil-g and g-floor.  `c` is the level-scaled centre, d the integer window offset.
struct AxisTap {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ AxisTap axis_tap(float c, int d, int size) {
    float g = c + (float)d;
    g = fminf(fmaxf(g, 0.f), (float)(size - 1));
    float g0 = floorf(g), g1 = ceilf(g);
    AxisTap t;
    t.i0 = (int)g0;
    t.i1 = (int)g1;
    t.w0 = g1 - g;
    t.w1 = g - g0;
    return t;
}

