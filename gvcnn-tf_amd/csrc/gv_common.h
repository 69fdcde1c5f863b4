// Internal helpers shared by the HIP translation units of libgvcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvcnn_hip.h"

#define GV_HIP_CHECK(expr)                           \
    do {                                             \
        hipError_t _e = (expr);                      \
        if (_e != hipSuccess) return (int)_e;        \
    } while (0)

#define GV_LAUNCH_CHECK()                            \
    do {                                             \
        hipError_t _e = hipGetLastError();           \
        if (_e != hipSuccess) return (int)_e;        \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int gv_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool gv_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Bijective XCD-aware remap of a 1-D grid: blocks with equal (bid % 8) share an XCD (and its
// L2) under round-robin dispatch, so give each XCD one contiguous chunk of logical tile ids.
// Placement only changes speed, never results.
__device__ __forceinline__ int gv_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
