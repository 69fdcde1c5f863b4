// Element helpers of the 16-bit storage kernels (lowp.hip, train_lp.hip): a 16-bit element travels as an
// unsigned short, arithmetic is fp32, T = __bf16 or _Float16 selects the encoding.
#pragma once
#include "gv_common.h"

namespace gvlp_elem {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ float up(unsigned short b) { return (float)__builtin_bit_cast(T, b); }
template <typename T>
__device__ __forceinline__ unsigned short down(float v) { const T h = (T)v; return __builtin_bit_cast(unsigned short, h); }

template <typename T>
__device__ __forceinline__ void unpack8(u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = up<T>((unsigned short)(v[j] & 0xffffu));
        f[2 * j + 1] = up<T>((unsigned short)(v[j] >> 16));
    }
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (unsigned)down<T>(f[2 * j]) | ((unsigned)down<T>(f[2 * j + 1]) << 16);
    return v;
}

// load / store VEC (8 or 1) consecutive elements as fp32
template <typename T, int VEC>
__device__ __forceinline__ void load_v(const unsigned short* p, float (&f)[8]) {
    if constexpr (VEC == 8) unpack8<T>(*reinterpret_cast<const u32x4*>(p), f);
    else f[0] = up<T>(*p);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_v(unsigned short* p, const float (&f)[8]) {
    if constexpr (VEC == 8) *reinterpret_cast<u32x4*>(p) = pack8<T>(f);
    else *p = down<T>(f[0]);
}

// one 16-byte quad of storage elements <-> fp32 registers: 8 elements of a 16-bit type T, or 4 floats (E = float)
template <typename E, typename T>
__device__ __forceinline__ void unpackq(u32x4 v, float (&f)[8]) {
    if constexpr (sizeof(E) == 2) {
        unpack8<T>(v, f);
    } else {
        const f32x4 w = __builtin_bit_cast(f32x4, v);
        f[0] = w[0]; f[1] = w[1]; f[2] = w[2]; f[3] = w[3];
    }
}
template <typename E, typename T>
__device__ __forceinline__ u32x4 packq(const float (&f)[8]) {
    if constexpr (sizeof(E) == 2) {
        return pack8<T>(f);
    } else {
        const f32x4 w = {f[0], f[1], f[2], f[3]};
        return __builtin_bit_cast(u32x4, w);
    }
}

// the same access pattern on fp32 storage (VEC = 4: one 16-byte load), so kernels templated on the pointer type serve
// both storages
template <typename T, int VEC>
__device__ __forceinline__ void load_v(const float* p, float (&f)[8]) {
    if constexpr (VEC == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    } else {
        f[0] = *p;
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void store_v(float* p, const float (&f)[8]) {
    if constexpr (VEC == 4) {
        f32x4 v = {f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4*>(p) = v;
    } else {
        *p = f[0];
    }
}

}  // namespace gvlp_elem
