// Internal helpers shared by the HIP translation units of libgvcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting and one process may drive several GPUs
// (model.pin_device): remember its outcome per device, not per process.
struct GvPerDeviceOnce {
    std::atomic<signed char> state[64] = {};                     // 0 = not tried, 1 = ok, -1 = failed
    template <typename F>
    bool ok(F&& set) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
        signed char v = state[dev].load(std::memory_order_acquire);
        if (v == 0) {
            v = set() == hipSuccess ? 1 : -1;
            state[dev].store(v, std::memory_order_release);
        }
        return v > 0;
    }
};
// true when `kernel` may be launched with `bytes` of dynamic LDS on the current device
#define GV_BIG_LDS_OK(kernel, bytes)                                                                              \
    ([&] {                                                                                                        \
        static GvPerDeviceOnce once_;                                                                             \
        return once_.ok([&] {                                                                                     \
            return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)(bytes));                                                             \
        });                                                                                                       \
    }())

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Every pooling window must hold at least one valid tap (else an average divides 0 by 0 and a max writes -inf / a 0xff
// argmax byte): the last window starts inside the image and the padding is smaller than the window.
static inline bool gv_pool_geometry_ok(const gv_pool_desc* d) {
    return d->pad_t < d->kh && d->pad_l < d->kw && (int64_t)(d->oh - 1) * d->stride - d->pad_t < d->ih &&
           (int64_t)(d->ow - 1) * d->stride - d->pad_l < d->iw;
}

// Exact division of a 31-bit dividend by a constant (Granlund-Montgomery, 31-bit precision): q = hi32(m * mul) >> sh.
struct GvFastDiv {
    unsigned mul;
    int sh;                                                      // -1: divisor 1 (q = m)
    int d;
};
static inline GvFastDiv gv_fast_div(int d) {
    GvFastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.sh = -1; return f; }
    int l = 0;
    while ((1ll << l) < d) ++l;                                  // l = ceil(log2 d) >= 1
    f.mul = (unsigned)(((1ull << (31 + l)) + (unsigned)d - 1) / (unsigned)d);
    f.sh = l - 1;
    return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ int gv_div(int m, const GvFastDiv& f) {
    return f.sh < 0 ? m : (int)(__umulhi((unsigned)m, f.mul) >> f.sh);
}
#endif

static inline int gv_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool gv_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- where a filter-gradient launch puts its partial tiles ------------------------------------------------------
// A filter gradient splits the pixel axis over workgroups ("slices"); every slice produces a partial of each
// element of dW it owns.  Two ways to combine them:
//   part == nullptr   fp32 atomic adds into dw (the order the slices arrive in decides the rounding: two runs differ in
//                     the last bits) — or, with ONE slice, the plain sum: a single adder per element;
//   part != nullptr   slice s STORES its partials into part + s * stride (an image of dW per slice), and a second launch
//                     (gv_dw_finish) adds the slices in index order into dw: the same bits every run, whatever the
//                     dispatch order.  (gv_conv2d_wgrad_ws; reference: the deterministic CPU gradients of
//                     utils/train_utils.py:217-259.)
struct GvDw {
    float* dw;
    float* part;
    size_t stride;                                               // elements between two slices' images (multiple of 4)
    size_t cap_bytes;                                            // host side: size of the workspace behind `part`
};
static inline GvDw gv_dw_plain(float* dw) { return GvDw{dw, nullptr, 0, 0}; }
// The slices a launch may use with this sink: a workspace holds cap_bytes / image slices; without room for two the
// launch runs un-split (one adder per element is deterministic as well).
static inline int64_t gv_dw_clamp(const GvDw& o, size_t elems, int64_t splits) {
    if (!o.part) return splits;
    const int64_t cap = (int64_t)(o.cap_bytes / (((elems + 3) / 4 * 4) * sizeof(float)));
    return cap < 2 ? 1 : (splits < cap ? splits : cap);
}
// The sink the KERNEL gets for `splits` slices (the plain one for a single slice) ...
static inline GvDw gv_dw_sink(const GvDw& o, size_t elems, int64_t splits) {
    if (!o.part || splits <= 1) return GvDw{o.dw, nullptr, 0, 0};
    return GvDw{o.dw, o.part, (elems + 3) / 4 * 4, o.cap_bytes};
}
// ... and the launch that follows it: dw[i] += part[0][i] + part[1][i] + ... in slice order (nothing to do for a plain sink)
int gv_dw_finish(const GvDw& o, size_t elems, int64_t splits, hipStream_t st);      // train.hip
#if defined(__HIPCC__)
__device__ __forceinline__ void gv_dw_put(const GvDw& o, int64_t slice, size_t idx, float v) {
    if (o.part) o.part[(size_t)slice * o.stride + idx] = v;      // (wave-uniform branch)
    else atomicAdd(&o.dw[idx], v);
}
#endif

// Bijective XCD-aware remap of a 1-D grid: blocks with equal (bid % 8) share an XCD (and its
// L2) under round-robin dispatch, so give each XCD one contiguous chunk of logical tile ids.
// Placement only changes speed, never results.
__device__ __forceinline__ int gv_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
