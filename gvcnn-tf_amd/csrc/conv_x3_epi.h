// Three-plane ("P3") storage of fp32 activations and the LDS-staged epilogue that writes it.
//
// P3: every fp32 value a is kept as three bf16 planes a = a0 + a1 + a2 (a0 = bf16(a), a1 = bf16(a - a0),
// a2 = bf16(a - a0 - a1)), laid out [pixel][channel/16][plane][16]: 96 bytes per 16 channels, i.e. 6 bytes per value, and
// the 16 k-values one MFMA step needs of one plane are 32 contiguous bytes (the packed filters' layout).  A consumer
// convolution whose loader only has to MOVE these bytes (conv_dma.hip) saves the ~5.5 VALU instructions per element and
// filter tap that splitting fp32 in the loader costs (conv_bf16s.hip) — the split is paid once, here, by the producer.
//
// The epilogue: each wave transposes its accumulators through a private LDS block so that a lane owns 8 CONSECUTIVE
// channels of one pixel (as conv_lp_epi.h does for 16-bit storage); scale/shift/residual/ReLU on fp32 values; then per
// destination either two 16-byte fp32 stores or the three 16-byte plane stores of P3.
#pragma once
#include "conv_common.h"

namespace {

typedef unsigned p3_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p3_bf16x2 __attribute__((ext_vector_type(2)));

// 8 consecutive fp32 -> NP x (8 bf16 packed in 16 bytes)
template <int NP>
__device__ __forceinline__ void p3_split8(const float (&v)[8], p3_u32x4 (&out)[NP]) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[j];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const p3_bf16x2 pr = {(__bf16)x[2 * q], (__bf16)x[2 * q + 1]};
            out[p][q] = __builtin_bit_cast(unsigned, pr);
            if (p + 1 < NP) {
                x[2 * q] -= (float)pr[0];
                x[2 * q + 1] -= (float)pr[1];
            }
        }
    }
}

// byte offset of the 8 channels starting at channel c (a multiple of 8) of pixel `pix` in a P3 tensor with pixel stride
// ld channels (a multiple of 16), relative to the tensor's (group aligned) base; plane p follows at + 32 * p
__device__ __forceinline__ size_t p3_byte_off(size_t pix, int ld, int c) {
    return (pix * (size_t)ld + (size_t)(c & ~15)) * 6 + (size_t)((c >> 3) & 1) * 16;
}

__device__ __forceinline__ void p3_store8(char* base, size_t pix, int ld, int c, const float (&v)[8]) {
    p3_u32x4 pl[3];
    p3_split8<3>(v, pl);
    char* dst = base + p3_byte_off(pix, ld, c);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<p3_u32x4*>(dst + p * 32) = pl[p];
}

// 4 consecutive channels (c a multiple of 4) of a P3 pixel -> fp32 (exact sum of the planes, small terms first)
__device__ __forceinline__ f32x4 p3_load4(const char* base, size_t pix, int ld, int c) {
    const char* src = base + (pix * (size_t)ld + (size_t)(c & ~15)) * 6 + (size_t)(c & 15) * 2;
    f32x4 r;
    unsigned long long pl[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) pl[p] = *reinterpret_cast<const unsigned long long*>(src + p * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a0 = __builtin_bit_cast(float, (unsigned)((pl[0] >> (16 * e)) & 0xffffull) << 16);
        const float a1 = __builtin_bit_cast(float, (unsigned)((pl[1] >> (16 * e)) & 0xffffull) << 16);
        const float a2 = __builtin_bit_cast(float, (unsigned)((pl[2] >> (16 * e)) & 0xffffull) << 16);
        r[e] = (a2 + a1) + a0;
    }
    return r;
}

// 4 consecutive channels (c a multiple of 4) of a P3 pixel <- fp32: three 8-byte stores
__device__ __forceinline__ void p3_store4(char* base, size_t pix, int ld, int c, f32x4 v) {
    char* dst = base + (pix * (size_t)ld + (size_t)(c & ~15)) * 6 + (size_t)(c & 15) * 2;
    float x[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        unsigned long long w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)x[e];
            w |= (unsigned long long)__builtin_bit_cast(unsigned short, h) << (16 * e);
            x[e] -= (float)h;
        }
        *reinterpret_cast<unsigned long long*>(dst + p * 32) = w;
    }
}

template <int TN> struct X3EpiGeom {
    static constexpr int CW = (TN % 2 == 0) ? 64 : 32;     // columns per staging block
    static constexpr int JB = CW / 32;                     // MFMA tiles per block
    static constexpr int BYTES = 32 * CW * 4;              // per wave
};

// y (columns < split, or all) and y2 (GV_CONV_SPLIT: columns >= split) each fp32 or P3 (a.y_p3 / a.y2_p3).  Needs
// 8-aligned column counts and, for a P3 destination, 16-aligned strides / split / slice offsets (checked by the host).
template <int TM, int TN>
__device__ __forceinline__ void x3_epilogue_staged(const gvconv::ConvArgs& a, const f32x16 (&acc)[TM][TN], int m0, int n0,
                                                   int wm, int wn, int lane, float* stage, const float* sstab = nullptr,
                                                   int bn = 0) {
    if (a.dbg & 4) {            // timing ablation: keep the accumulators live without storing the tile
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 1.2345e-30f) a.y[0] = t;
        return;
    }
    constexpr int CW = X3EpiGeom<TN>::CW, JB = X3EpiGeom<TN>::JB;
    constexpr int CPR = CW / 8;                            // 8-column chunks per row
    constexpr int RPP = 64 / CPR;                          // rows per read-back pass
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    const int rrow = lane / CPR, rchunk = lane % CPR;
#pragma unroll
    for (int jb = 0; jb < TN / JB; ++jb) {
        const int col = n0 + (wn * TN + jb * JB) * 32 + rchunk * 8;     // this lane's 8 columns
        const bool live = col < a.cout;                                 // (cout % 8 == 0: a chunk is whole or absent)
        float sc[8], sh[8];
        if (sstab) {                                        // the tile's constants in LDS (conv_dma.hip): 16-byte reads
            const float* t = sstab + (wn * TN + jb * JB) * 32 + rchunk * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(t + 4 * h);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(t + bn + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[4 * h + e] = v0[e]; sh[4 * h + e] = v1[e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = min(col + e, a.cout - 1);
                sc[e] = a.scale[c];
                sh[e] = a.shift[c];
            }
        }
        const bool to_second = a.split > 0 && col >= a.split;
        const int dcol = to_second ? col - a.split : col;
        char* dst = reinterpret_cast<char*>(to_second ? a.y2 : a.y);
        const int dld = to_second ? a.y2_ld : a.y_ld;
        const bool dp3 = to_second ? a.y2_p3 != 0 : a.y_p3 != 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(row_h + (r & 3) + 8 * (r >> 2)) * CW + jj * 32 + col_l] = acc[i][jb * JB + jj][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pass = 0; pass < 32 / RPP; ++pass) {
                const int row = pass * RPP + rrow;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * CW + rchunk * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * CW + rchunk * 8 + 4);
                const int m = m0 + (wm * TM + i) * 32 + row;
                if (m >= a.M || !live) continue;
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                if (a.res) {
                    const float* rp = a.res + (size_t)m * a.res_ld + col;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (col + e < a.relu_limit) ? fmaxf(v[e], 0.f) : v[e];
                }
                if (dp3) {
                    p3_store8(dst, (size_t)m, dld, dcol, v);
                } else {
                    float* yp = reinterpret_cast<float*>(dst) + (size_t)m * dld + dcol;
                    *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(yp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace
