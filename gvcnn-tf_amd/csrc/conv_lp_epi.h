// Pieces shared by the 16-bit-storage convolution kernels (conv_lp.hip: register-staged loader; conv_dma.hip:
// direct global->LDS loader): MFMA wrapper, 16-bit <-> fp32 helpers and the LDS-staged epilogue.
#pragma once
#include <type_traits>

#include "conv_common.h"
#include "conv_stats.h"

namespace {

using gvconv::ConvArgs;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


template <typename T>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (std::is_same<T, __bf16>::value)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ unsigned short to_bits(float v) {
    const T h = (T)v;
    return __builtin_bit_cast(unsigned short, h);
}
template <typename T>
__device__ __forceinline__ float from_bits(unsigned short b) {
    return (float)__builtin_bit_cast(T, b);
}

// Staged epilogue: each wave transposes its accumulators through a private LDS block (32 rows x CW fp32) so
// that a lane ends up with 8 CONSECUTIVE channels of one pixel: scale/shift/residual/ReLU are applied to
// the fp32 values, which are rounded once and leave as 16-byte stores (residual and second output: 16-byte
// loads / stores as well).  Storing straight from the MFMA layout would issue 2-byte stores, one per lane
// per row — 8x the store instructions, and the kernel then spends more time storing than multiplying.
template <int TN> struct EpiGeom {
    static constexpr int CW = (TN % 2 == 0) ? 64 : 32;     // columns per staging block
    static constexpr int JB = CW / 32;                     // MFMA tiles per block
    static constexpr int BYTES = 32 * CW * 4;              // per wave
};

// Two fp32 values -> one dword of two 16-bit values, round to nearest even: ONE v_cvt_pk_bf16_f32 (bf16; two scalar
// conversions cost a conversion each plus the shift / or that packs them).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    if constexpr (std::is_same<T, __bf16>::value) return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2_t));
}
template <typename T>
__device__ __forceinline__ void store_chunk(unsigned short* dst, const float (&v)[8], int nvalid, bool vec) {
    if (vec && nvalid == 8) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack2<T>(v[2 * j], v[2 * j + 1]);
        *reinterpret_cast<u32x4*>(dst) = o;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nvalid) dst[j] = to_bits<T>(v[j]);
    }
}

// The lean epilogue's store: eight values as one 16-byte store; ReLU on the PACKED words — a negative bf16 / f16 is a
// negative int16, so max(x, 0) is one v_pk_max_i16 per two values (fmaxf on the fp32 values: two v_max each, the first
// one canonicalising).  -0.0 and negative NaNs become +0.
template <typename T>
__device__ __forceinline__ void store_chunk_lean(unsigned short* dst, const float (&v)[8], bool relu) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack2<T>(v[2 * j], v[2 * j + 1]);
    if (relu) {
        // (inline asm: __builtin_elementwise_max on the bit-cast short2 words is miscompiled by this hipcc — ONE v_pk_max_i16,
        // of word 0, feeds all four results)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned r;
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(o[j]));
            o[j] = r;
        }
    }
    *reinterpret_cast<u32x4*>(dst) = o;
}

// STATS (conv_stats.h): the launch also produces the train-mode BatchNorm sums of the tensor it stores, from the rounded
// values it stores (smem: the workgroup's dynamic LDS, which holds the tables; mbase: first pixel of the image the
// tile's first row lies in).  STAT_BWD reads the BatchNorm's input z next to every chunk it stores.
template <typename T, int TM, int TN, int STATS = 0>
__device__ __forceinline__ void lp_epilogue_staged(const ConvArgs& a, const f32x16 (&acc)[TM][TN], int m0, int n0,
                                                   int wm, int wn, int lane, float* stage, int rows_valid = 32,
                                                   const float* sstab = nullptr, int bn = 0, char* smem = nullptr,
                                                   int mbase = 0) {
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
    constexpr int CW = EpiGeom<TN>::CW, JB = EpiGeom<TN>::JB;
    constexpr int CPR = CW / 8;                            // 8-column chunks per row
    constexpr int RPP = 64 / CPR;                          // rows per read-back pass
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    const int rrow = lane / CPR, rchunk = lane % CPR;
    const unsigned short* res = reinterpret_cast<const unsigned short*>(a.res);
    unsigned short* y = reinterpret_cast<unsigned short*>(a.y);
    unsigned short* y2 = reinterpret_cast<unsigned short*>(a.y2);
    // LEAN (the STATS instantiations): ONE destination, no activation, whole 16-byte aligned 8-column chunks — the entry
    // point checks it — so none of the second-output / split / ReLU / element-wise branches is compiled in: the unrolled
    // epilogue is straight-line code that every workgroup runs once, and past the instruction cache's 64 KB it is fetched
    // again for every workgroup (the 128 x 192 LDS-DMA tile: 50 KB plain, 80 KB with every branch AND the sums).
    constexpr bool LEAN = STATS != 0;                  // (STAT_LEAN: the same epilogue without the sums)
    constexpr bool HAS = gvconv::stat_has(STATS);
    const bool dual = !LEAN && y2 != nullptr && a.split == 0;
    // 16-byte accesses need 8-element aligned rows, slices and boundaries (true for every layer of both
    // backbones); anything else takes the element-wise branch of store_chunk
    const bool vec = LEAN || ((a.y_ld % 8 == 0) && ((((uintptr_t)y) & 15) == 0) &&
                     (y2 == nullptr || ((a.y2_ld % 8 == 0) && ((((uintptr_t)y2) & 15) == 0))) &&
                     (a.split % 8 == 0) && (res == nullptr || ((a.res_ld % 8 == 0) && ((((uintptr_t)res) & 15) == 0))));
    // output pixel of GEMM row m: m itself, or (y_step 2: one parity class of a stride-2 data gradient) pixel
    // (2*oy + y_py, 2*ox + y_px) of image n in a y_ih x y_iw map
    auto out_pix = [&](int m) -> size_t {
        if (a.y_step != 2) return (size_t)m;
        const int n = gv_div(m, a.y_div_img);
        const int rem = m - n * a.y_div_img.d;
        const int oy = gv_div(rem, a.y_div_row);
        const int ox = rem - oy * a.y_div_row.d;
        return ((size_t)n * a.y_ih + (size_t)(2 * oy + a.y_py)) * a.y_iw + (size_t)(2 * ox + a.y_px);
    };
    // The residual chunks of a block's read-back passes are requested ONE BLOCK AHEAD (those of the first block before it
    // is staged): issued one per pass next to their use, every pass of an HBM-bound ResNet conv3 (K = 64 ... 256, 4x the
    // output channels) waits out one full memory round trip (the y stores in between may alias, so the compiler cannot
    // hoist the loads itself).  conv3 of block1: 0.365 -> 0.305 ms with the block's own chunks up front.
    constexpr int NPASS = 32 / RPP;
    // This lane's rows: (block row i, read-back pass) -> output pixel and "is stored", computed ONCE (the residual fetch,
    // the z fetch of the backward sums and the stores all use them; a wave issues one instruction per 4 cycles at
    // best, so every instruction of this once-per-workgroup code is 4 clocks of a workgroup's life).
    unsigned mpix[TM][NPASS];
    unsigned mvalid = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int row = pass * RPP + rrow;
            const int m = m0 + (wm * TM + i) * 32 + row;
            const bool ok = m < a.M && row < rows_valid;
            mpix[i][pass] = (unsigned)out_pix(ok ? m : m0);
            if (ok) mvalid |= 1u << (i * NPASS + pass);
        }
    u32x4 rvp[2][NPASS];
    auto res_fetch = [&](int b, u32x4 (&dst)[NPASS]) {
        const int jb = b / TM, i = b % TM;
        const int col = n0 + (wn * TN + jb * JB) * 32 + rchunk * 8;
        if (res != nullptr && vec && a.cout - col >= 8) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass)
                dst[pass] = ((mvalid >> (i * NPASS + pass)) & 1u)
                                ? *reinterpret_cast<const u32x4*>(res + (size_t)mpix[i][pass] * a.res_ld + col)
                                : u32x4{0u, 0u, 0u, 0u};
        }
    };
    res_fetch(0, rvp[0]);
    // STATS: everything that comes from the kernel arguments is fetched HERE, once — a scalar load inside the loops
    // below would wait (lgkmcnt) for every LDS access in flight with it.  Per column block: does this lane's chunk / this
    // lane's read column belong to a segment; STAT_BWD: where its z lives.  The z chunks of a block are requested a
    // block ahead, like the residual's.
    constexpr int NJB = TN / JB;
    unsigned long long* st_sums = nullptr;
    const float2* st_ss = nullptr;
    unsigned st_on_mask = 0, st_col_mask = 0;
    int st_dbg = 0;
    const unsigned short* st_z[STATS == gvconv::STAT_BWD ? NJB : 1];
    int st_zld[STATS == gvconv::STAT_BWD ? NJB : 1];
    u32x4 zvp[STATS == gvconv::STAT_BWD ? 2 : 1][STATS == gvconv::STAT_BWD ? NPASS : 1];
    auto z_fetch = [&](int b, u32x4 (&dst)[STATS == gvconv::STAT_BWD ? NPASS : 1]) {
        if constexpr (STATS == gvconv::STAT_BWD) {
            const int jb = b / TM, i = b % TM;
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass)
                dst[pass] = (((st_on_mask >> jb) & 1u) && ((mvalid >> (i * NPASS + pass)) & 1u))
                                ? *reinterpret_cast<const u32x4*>(st_z[jb] + (size_t)mpix[i][pass] * st_zld[jb])
                                : u32x4{0u, 0u, 0u, 0u};
        }
    };
    gvconv::StatWave<HAS ? STATS : gvconv::STAT_FWD> sw;
    const int st_mw0 = m0 + wm * TM * 32;                             // the wave's first pixel
    bool st_has_b = false;
    if constexpr (HAS) {
        st_sums = reinterpret_cast<unsigned long long*>(smem + a.st.lds_off);
        st_ss = reinterpret_cast<const float2*>(smem + a.st.lds_off + (size_t)a.st.slots * bn * 16);
        st_dbg = a.st.dbg;
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            const int col = n0 + (wn * TN + jb * JB) * 32 + rchunk * 8;
            const int k = a.cout - col >= 8 ? gvconv::stat_seg_of(a.st, col) : -1;
            if (k >= 0) st_on_mask |= 1u << jb;
            const int rcol = n0 + (wn * TN + jb * JB) * 32 + lane;
            if (lane < CW && rcol < a.cout && gvconv::stat_seg_of(a.st, rcol) >= 0) st_col_mask |= 1u << jb;
            if constexpr (STATS == gvconv::STAT_BWD) {
                st_z[jb] = k >= 0 ? a.st.seg[k].z + (col - a.st.seg[k].c0) : nullptr;
                st_zld[jb] = k >= 0 ? a.st.seg[k].z_ld : 0;
            }
        }
        sw.begin_wave(a.st, st_mw0, mbase);
        st_has_b = sw.e1 < min(st_mw0 + TM * 32, a.M);
        z_fetch(0, zvp[0]);
    }
#pragma unroll
    for (int jb = 0; jb < TN / JB; ++jb) {
        const int col = n0 + (wn * TN + jb * JB) * 32 + rchunk * 8;     // this lane's 8 columns
        const int nvalid = min(8, a.cout - col);                        // <= 0: nothing to store
        const int st_lcol = (wn * TN + jb * JB) * 32 + rchunk * 8;
        bool st_on = false;
        if constexpr (HAS) {
            st_on = ((st_on_mask >> jb) & 1u) != 0;
            sw.begin_block(st_ss, bn, st_lcol);
        }
        float sc[8], sh[8], sc2[8], sh2[8];
        if (sstab) {                                        // the tile's constants in LDS (conv_dma.hip): 16-byte reads
            const float* t = sstab + (wn * TN + jb * JB) * 32 + rchunk * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(t + 4 * h);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(t + bn + 4 * h);
                f32x4 v2 = {0.f, 0.f, 0.f, 0.f}, v3 = v2;
                if (dual) {
                    v2 = *reinterpret_cast<const f32x4*>(t + 2 * bn + 4 * h);
                    v3 = *reinterpret_cast<const f32x4*>(t + 3 * bn + 4 * h);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[4 * h + e] = v0[e]; sh[4 * h + e] = v1[e]; sc2[4 * h + e] = v2[e]; sh2[4 * h + e] = v3[e]; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = min(col + e, a.cout - 1);
                sc[e] = a.scale[c];
                sh[e] = a.shift[c];
                sc2[e] = dual ? a.scale2[c] : 0.f;
                sh2[e] = dual ? a.shift2[c] : 0.f;
            }
        }
        // split destination: a chunk lies on one side when split % 8 == 0; otherwise decide per element below
        const bool to_second = !LEAN && a.split > 0 && col >= a.split;
        // STAT_LEAN: this column block's destination, once per block (a fused sibling GEMM: columns >= split go to the
        // second destination, columns >= relu_limit — the pooled branch, activated after its pool — keep their sign; both
        // boundaries are chunk aligned there)
        const bool lean_second = STATS == gvconv::STAT_LEAN && a.split > 0 && col >= a.split;
        unsigned short* const lean_base = lean_second ? y2 + (col - a.split) : y + col;
        const unsigned lean_ld = lean_second ? (unsigned)a.y2_ld : (unsigned)a.y_ld;
        const bool lean_relu = a.relu != 0 && col < a.relu_limit;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int b = jb * TM + i;
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(row_h + (r & 3) + 8 * (r >> 2)) * CW + jj * 32 + col_l] = acc[i][jb * JB + jj][r];
            __builtin_amdgcn_wave_barrier();
            if (b + 1 < TM * (TN / JB)) res_fetch(b + 1, rvp[(b + 1) & 1]);
            if constexpr (STATS == gvconv::STAT_BWD) {
                if (b + 1 < TM * (TN / JB)) z_fetch(b + 1, zvp[(b + 1) & 1]);
            }
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int row = pass * RPP + rrow;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * CW + rchunk * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * CW + rchunk * 8 + 4);
                const int m = m0 + (wm * TM + i) * 32 + row;
                if (!((mvalid >> (i * NPASS + pass)) & 1u) || nvalid <= 0) continue;
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const size_t mp = mpix[i][pass];                  // (m itself unless a parity-class launch)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                if (res) {
                    const unsigned short* rp = res + mp * a.res_ld + col;
                    if (LEAN || (vec && nvalid == 8)) {
                        const u32x4 rv = rvp[b & 1][pass];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[2 * j] += from_bits<T>((unsigned short)(rv[j] & 0xffffu));
                            v[2 * j + 1] += from_bits<T>((unsigned short)(rv[j] >> 16));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < nvalid) v[e] += from_bits<T>(rp[e]);
                    }
                }
                if (dual) {
                    float v2[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v2[e] = v[e] * sc2[e] + sh2[e];
                        if (a.relu2) v2[e] = fmaxf(v2[e], 0.f);
                    }
                    store_chunk<T>(y2 + mp * a.y2_ld + col, v2, nvalid, vec);
                }
                if (!LEAN && a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (col + e < a.relu_limit) ? fmaxf(v[e], 0.f) : v[e];
                }
                u32x4 packed;                                     // (sums instantiations: the chunk, rounded and packed ONCE —
                if constexpr (HAS) {                              //  the same words are summed and stored)
#pragma unroll
                    for (int j = 0; j < 4; ++j) packed[j] = pack2<T>(v[2 * j], v[2 * j + 1]);
                    if (st_on && !(st_dbg & 8192)) {              // sums of the values exactly as stored below
                        float rr[8], zv[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            rr[2 * j] = from_bits<T>((unsigned short)(packed[j] & 0xffffu));
                            rr[2 * j + 1] = from_bits<T>((unsigned short)(packed[j] >> 16));
                        }
                        if constexpr (STATS == gvconv::STAT_BWD) {
                            const u32x4 zq = zvp[b & 1][pass];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                zv[2 * j] = from_bits<T>((unsigned short)(zq[j] & 0xffffu));
                                zv[2 * j + 1] = from_bits<T>((unsigned short)(zq[j] >> 16));
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) zv[e] = 0.f;
                        }
                        const int step0 = st_mw0 + i * 32 + pass * RPP;
                        sw.add(rr, zv, m, step0, step0 + RPP);
                    }
                }
                if constexpr (STATS == gvconv::STAT_LEAN) {
                    store_chunk_lean<T>(lean_base + mp * lean_ld, v, lean_relu);
                } else if constexpr (LEAN) {
                    *reinterpret_cast<u32x4*>(y + mp * a.y_ld + col) = packed;
                } else if (a.split > 0 && (a.split % 8) != 0) {   // boundary inside a chunk: element-wise
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (e >= nvalid) continue;
                        const int c = col + e;
                        if (c >= a.split) y2[mp * a.y2_ld + (c - a.split)] = to_bits<T>(v[e]);
                        else y[mp * a.y_ld + c] = to_bits<T>(v[e]);
                    }
                } else if (to_second) {
                    store_chunk<T>(y2 + mp * a.y2_ld + (col - a.split), v, nvalid, vec);
                } else {
                    store_chunk<T>(y + mp * a.y_ld + col, v, nvalid, vec);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (HAS) {                   // the block's column totals: wave -> workgroup table
            const int lcol0 = (wn * TN + jb * JB) * 32;
            sw.template end_block<CW>(st_sums, stage, bn, lane, rrow, rchunk, lcol0, ((st_col_mask >> jb) & 1u) != 0, st_has_b, st_dbg);
        }
    }
}

}  // namespace
