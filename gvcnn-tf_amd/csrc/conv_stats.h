// Train-mode BatchNorm sums folded into a convolution's epilogue (16-bit storage kernels).
//
// The reference normalises every view's graph copy over its own N*h*w values (nets/model.py:129-141,
// nets/inception_utils.py:52-62): statistics are per (view, channel), view of image b = b % G.  Unfused, a
// BatchNorm costs two extra passes over its input per direction (gv_bn_sums_grouped_t, then the apply pass).  Here the
// launch that PRODUCES a tensor also produces that tensor's sums, from the values exactly as it stores them:
//   STAT_FWD  a forward convolution writing z:    acc[g][c] += { sum z, sum z^2 }          (replaces gv_bn_sums_grouped_t)
//   STAT_BWD  a data-gradient launch writing dy:  acc[g][c] += { sum g, sum g*z },  g = dy * [z*scale + shift > 0]
//             (replaces gv_bn_relu_bwd_sums_grouped_t; z is the BatchNorm's input, read in the epilogue; the apply
//             pass turns sum g*z into sum g*zhat = inv * (sum g*z - mean * sum g): GV_ACCUM_RAW_Z).
// The output's channels may belong to several BatchNorm layers (members of a fused sibling GEMM; the concat that a
// block's data gradient writes): up to STAT_MAX_SEG column segments, each with its own accumulator.
//
// Per workgroup: an LDS table [slot][tile column][2] of partial sums, slot = image index relative to the tile's first
// image (a row tile of BM pixels spans a few images).  A wave sums the rows it stores in fp32 in a fixed order (per lane,
// then across its lanes through its private staging block), rounds each column's total to a fixed binary grid and adds
// it to the table as a 64-bit INTEGER (ds_add_u64: associative, so the table does not depend on the order the waves
// arrive in); one thread per column then adds the table to the fp64 accumulators.
// Those additions are exact — hence also independent of their order, and the statistics bitwise reproducible run to
// run — as long as a total stays below 2^53 grid steps (beyond that they degrade to ordinary fp64 rounding).
#pragma once
#include "gv_common.h"

namespace gvconv {

constexpr int STAT_MAX_SEG = 8;
constexpr int STAT_MAX_SLOTS = 64;           // images one row tile may span (before folding by group)
constexpr int STAT_OFF = 0, STAT_FWD = 1, STAT_BWD = 2;
// Kernel template value only (never a ConvStats::mode): no sums, but the LEAN staged epilogue the sums instantiations
// use — one 16-bit destination in whole aligned 8-column chunks, optional residual, uniform ReLU.  Nearly every launch
// of both backbones qualifies (lp_epilogue_lean_ok); the full epilogue (second output, split destination, partial ReLU,
// element-wise stores) is 12 - 30 KB more straight-line code that every workgroup streams through once, and the larger
// tiles then exceed the 64 KB instruction cache.
constexpr int STAT_LEAN = 3;
constexpr bool stat_has(int s) { return s == STAT_FWD || s == STAT_BWD; }
// grid of the partial sums: 2^-30 for sum z (totals exact below 2^23), 2^-24 for sum z^2 (2^29); the backward sums
// are sums of gradients, orders of magnitude smaller: 2^-40 (exact below 2^13)
constexpr double STAT_Q_FWD0 = 1073741824.0, STAT_Q_FWD1 = 16777216.0, STAT_Q_BWD = 1099511627776.0;

struct StatSeg {
    int c0, c1;                 // columns [c0, c1) of the launch's output tensor
    int z_ld, pad_;             // STAT_BWD: pixel stride of z
    const unsigned short* z;    // STAT_BWD: the BatchNorm's input, channel c0 of the segment at z[pixel * z_ld]
    const float* scale;         // STAT_BWD: [G][c1-c0] folded scale / shift of the forward pass (mask = z*scale+shift > 0);
    const float* shift;         //           nullptr: no ReLU behind this BatchNorm (mask = 1)
    double* acc;                // [G][c1-c0][2]; nullptr: no sums for these columns
};

struct ConvStats {
    int mode = 0;               // STAT_*
    int hw;                     // pixels per image of the OUTPUT tensor
    int G;                      // groups: image b belongs to group b % G
    int nseg;
    unsigned hw_magic;          // ceil(2^32 / hw): exact quotients for dividends < 2^32 / hw
    int lds_off;                // byte offset of the tables in dynamic LDS (set by the launcher)
    int slots;                  // table rows: images one row tile can span, at most G (image b0 + r and b0 + r + G share row r)
    int fold;                   // the tile can span more than G images: slot indices are reduced modulo G
    int dbg, pad_;              // timing experiments (gv_conv2d_set_debug): 2048 no publish, 4096 no table adds, 8192 no sums
    StatSeg seg[STAT_MAX_SEG];
};

// images a tile of `bm` consecutive pixels can span
static inline int stat_slots(int bm, int hw) { return (bm - 2) / hw + 2; }
// LDS bytes of the tables: sums [slots][bn][2] int64 (+ scale/shift [slots][bn] float2 for STAT_BWD)
static inline size_t stat_lds_bytes(int mode, int slots, int bn) {
    return mode == STAT_OFF ? 0 : (size_t)slots * bn * (mode == STAT_BWD ? 24 : 16);
}
// table rows for row tiles of `bm` pixels: one per image the tile can span, folded onto the G groups
static inline int stat_rows(int bm, int hw, int G) { const int n = stat_slots(bm, hw); return n < G ? n : G; }
// Can a kernel with row tiles of `bm` pixels fold the sums?  (whole 8-channel chunks per segment, the slot division
// exact, a bounded number of images per tile)
static inline bool stat_tile_ok(const ConvStats& s, int bm, int cout, int wave_rows) {
    if (s.mode == STAT_OFF) return true;
    if (s.hw < 2 || s.G < 1 || s.nseg < 1 || s.nseg > STAT_MAX_SEG || cout % 8 != 0) return false;
    if (s.hw <= wave_rows - 2) return false;          // a wave's rows span at most two images (its two accumulators)
    const int slots = stat_slots(bm, s.hw);
    if (slots > STAT_MAX_SLOTS) return false;
    if ((uint64_t)(bm + s.hw) * (uint64_t)s.hw >= 0xffffffffull) return false;
    for (int i = 0; i < s.nseg; ++i)
        if (s.seg[i].c0 % 8 != 0 || s.seg[i].c1 % 8 != 0 || s.seg[i].c0 < 0 || s.seg[i].c1 > cout || s.seg[i].c0 >= s.seg[i].c1)
            return false;
    return true;
}

#if defined(__HIPCC__)
// segment of output column `col` (-1: none, or a segment without an accumulator)
__device__ __forceinline__ int stat_seg_of(const ConvStats& s, int col) {
    int k = -1;
#pragma unroll
    for (int i = 0; i < STAT_MAX_SEG; ++i)
        if (i < s.nseg && col >= s.seg[i].c0 && col < s.seg[i].c1 && s.seg[i].acc != nullptr) k = i;
    return k;
}

// Workgroup step at the START of the epilogue (the main-loop LDS is free; a barrier follows): zero the sums table;
// STAT_BWD: fetch the (scale, shift) of every (table row, tile column).  b0 = image of the tile's first pixel.
template <int MODE>
__device__ __forceinline__ void stat_table_init(const ConvStats& s, char* smem, int tid, int nthreads, int bn, int n0, int cout,
                                                int b0) {
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(smem + s.lds_off);
    const int n = s.slots * bn;
    for (int i = tid; i < 2 * n; i += nthreads) sums[i] = 0ull;
    if constexpr (MODE == STAT_BWD) {
        float2* ss = reinterpret_cast<float2*>(smem + s.lds_off + (size_t)n * 16);
        for (int i = tid; i < n; i += nthreads) {
            const int slot = i / bn, col = n0 + (i - slot * bn);
            float2 v = make_float2(0.f, 1.f);                        // no ReLU / no segment: the mask is always true
            const int k = col < cout ? stat_seg_of(s, col) : -1;
            if (k >= 0 && s.seg[k].scale != nullptr) {
                const int g = (b0 + slot) % s.G, cs = s.seg[k].c1 - s.seg[k].c0;
                v.x = s.seg[k].scale[(size_t)g * cs + (col - s.seg[k].c0)];
                v.y = s.seg[k].shift[(size_t)g * cs + (col - s.seg[k].c0)];
            }
            ss[i] = v;
        }
    }
}

// A WAVE's sums over the rows it stores (WR = TM*32 consecutive pixels: at most two images, the launcher checks
// hw > WR - 2) for one block of CW columns; a lane owns one 8-column chunk and every (64 / CPR)-th row.  Rows of the
// wave's first image go to accumulator A, rows of the next one to B; which of the two a row belongs to is wave-uniform
// except in the one 8-row step that contains the boundary.  At the end of the block the wave transposes the lanes'
// partials through its private staging block (fixed summation order), and ONE lane per column adds the column's total
// to the workgroup table as a 64-bit integer: no same-address traffic inside a wave, nothing order dependent.
template <int MODE>
struct StatWave {
    float a0[8], a1[8], b0[8], b1[8];
    float scA[8], shA[8], scB[8], shB[8];                            // STAT_BWD: mask constants of the two images
    int e1;                                                          // first pixel of the wave's second image (uniform)
    int rowA, rowB;                                                  // their table rows (uniform)

    // mw0: the wave's first pixel; mbase: first pixel of the image the workgroup tile starts in
    __device__ __forceinline__ void begin_wave(const ConvStats& s, int mw0, int mbase) {
        const int sA = __builtin_amdgcn_readfirstlane((int)__umulhi((unsigned)(mw0 - mbase), s.hw_magic));
        e1 = mbase + (sA + 1) * s.hw;
        rowA = sA;
        rowB = sA + 1;
        if (s.fold) {                                                // more than G images in one workgroup tile
            while (rowA >= s.G) rowA -= s.G;
            while (rowB >= s.G) rowB -= s.G;
        }
    }
    __device__ __forceinline__ void begin_block(const float2* ss, int bn, int lcol) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a0[e] = a1[e] = b0[e] = b1[e] = 0.f;
        if constexpr (MODE == STAT_BWD) {
            const f32x4* pa = reinterpret_cast<const f32x4*>(ss + (size_t)rowA * bn + lcol);
            const f32x4* pb = reinterpret_cast<const f32x4*>(ss + (size_t)rowB * bn + lcol);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = pa[q], w = pb[q];
                scA[2 * q] = v[0]; shA[2 * q] = v[1]; scA[2 * q + 1] = v[2]; shA[2 * q + 1] = v[3];
                scB[2 * q] = w[0]; shB[2 * q] = w[1]; scB[2 * q + 1] = w[2]; shB[2 * q + 1] = w[3];
            }
        }
    }
    // One 8-column chunk of pixel m; step0 / step1: first / one-past-last pixel of the wave's current row step (uniform).
    // r[e]: the stored (rounded) values; zv[e]: z (STAT_BWD).
    __device__ __forceinline__ void add(const float (&r)[8], const float (&zv)[8], int m, int step0, int step1) {
        if (step1 <= e1) {                                           // (uniform) the whole step lies in the first image
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = MODE == STAT_FWD ? r[e] : (fmaf(zv[e], scA[e], shA[e]) > 0.f ? r[e] : 0.f);
                a0[e] += g;
                a1[e] = fmaf(g, MODE == STAT_FWD ? g : zv[e], a1[e]);
            }
        } else if (step0 >= e1) {                                    // (uniform) ... in the second
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = MODE == STAT_FWD ? r[e] : (fmaf(zv[e], scB[e], shB[e]) > 0.f ? r[e] : 0.f);
                b0[e] += g;
                b1[e] = fmaf(g, MODE == STAT_FWD ? g : zv[e], b1[e]);
            }
        } else {                                                     // the step that contains the boundary
            const bool inB = m >= e1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float g = r[e];
                if constexpr (MODE == STAT_BWD) g = fmaf(zv[e], inB ? scB[e] : scA[e], inB ? shB[e] : shA[e]) > 0.f ? r[e] : 0.f;
                const float p = g * (MODE == STAT_FWD ? g : zv[e]);
                a0[e] += inB ? 0.f : g;
                a1[e] += inB ? 0.f : p;
                b0[e] += inB ? g : 0.f;
                b1[e] += inB ? p : 0.f;
            }
        }
    }
    // End of a column block: lane (rrow, rchunk) of NR x CPR; stage = the wave's private block (>= 2*NR*CW floats);
    // lcol0 = first tile column of the block; col_ok = this lane's READ column (lcol0 + lane) takes part.
    template <int CW>
    __device__ __forceinline__ void end_block(unsigned long long* sums, float* stage, int bn, int lane, int rrow, int rchunk,
                                              int lcol0, bool col_ok, bool has_b, int dbg) {
        constexpr int NR = 64 / (CW / 8);
        const double q0 = MODE == STAT_FWD ? STAT_Q_FWD0 : STAT_Q_BWD, q1 = MODE == STAT_FWD ? STAT_Q_FWD1 : STAT_Q_BWD;
        if (dbg & 4096) return;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1 && !has_b) break;                          // (uniform) the wave's rows lie in one image
            const float (&v0)[8] = half ? b0 : a0;
            const float (&v1)[8] = half ? b1 : a1;
            __builtin_amdgcn_wave_barrier();
            f32x4* w0 = reinterpret_cast<f32x4*>(stage + rrow * CW + rchunk * 8);
            f32x4* w1 = reinterpret_cast<f32x4*>(stage + (NR + rrow) * CW + rchunk * 8);
            w0[0] = f32x4{v0[0], v0[1], v0[2], v0[3]};
            w0[1] = f32x4{v0[4], v0[5], v0[6], v0[7]};
            w1[0] = f32x4{v1[0], v1[1], v1[2], v1[3]};
            w1[1] = f32x4{v1[4], v1[5], v1[6], v1[7]};
            __builtin_amdgcn_wave_barrier();
            if (lane < CW && col_ok) {
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    t0 += stage[r * CW + lane];
                    t1 += stage[(NR + r) * CW + lane];
                }
                unsigned long long* t = sums + ((size_t)(half ? rowB : rowA) * bn + lcol0 + lane) * 2;
                const long long i0 = __double2ll_rn((double)t0 * q0), i1 = __double2ll_rn((double)t1 * q1);
                __hip_atomic_fetch_add(t, (unsigned long long)i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(t + 1, (unsigned long long)i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
};

// The strip kernels (conv3x3_halo_lp, conv_stem_patch_lp): a workgroup walks down a column strip of ONE image, so a lane
// (16 row lanes x 4 chunks of 8 channels per 32-column tile) keeps its sums in registers for the whole walk and the
// statistics leave the workgroup once, at the end: per wave a transposition through its private staging block (fixed
// order), then 64 lanes add one column total each to the fp64 accumulators (rounded to the grid: exact additions).
// One segment covering every output column (the launcher checks).
template <int MODE, int TN>
struct StatStrip {
    float s0[TN][8], s1[TN][8];
    float sc[TN][8], sh[TN][8];                                      // STAT_BWD: mask constants of the image's group
    const unsigned short* zb[TN];                                    // STAT_BWD: z at this lane's 8 channels of tile j
    int zld;
    __device__ __forceinline__ void init(const ConvStats& s, int g, int col8, int cout) {
        zld = s.seg[0].z_ld;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = j * 32 + col8;
            zb[j] = s.seg[0].z ? s.seg[0].z + c : nullptr;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s0[j][e] = s1[j][e] = 0.f;
                const int cc = c + e < cout ? c + e : cout - 1;
                sc[j][e] = (MODE == STAT_BWD && s.seg[0].scale) ? s.seg[0].scale[(size_t)g * cout + cc] : 0.f;
                sh[j][e] = (MODE == STAT_BWD && s.seg[0].scale) ? s.seg[0].shift[(size_t)g * cout + cc] : 1.f;
            }
        }
    }
    // r[e]: the stored (rounded) values of pixel m, column tile J; zq: the 16 bytes of z at that chunk (STAT_BWD)
    template <typename T, int J>
    __device__ __forceinline__ void add(const float (&r)[8], const unsigned (&zq)[4]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (MODE == STAT_FWD) {
                s0[J][e] += r[e];
                s1[J][e] = fmaf(r[e], r[e], s1[J][e]);
            } else {
                const unsigned short b = (unsigned short)((e & 1) ? (zq[e >> 1] >> 16) : (zq[e >> 1] & 0xffffu));
                const float zv = (float)__builtin_bit_cast(T, b);
                const float g = fmaf(zv, sc[J][e], sh[J][e]) > 0.f ? r[e] : 0.f;
                s0[J][e] += g;
                s1[J][e] = fmaf(g, zv, s1[J][e]);
            }
        }
    }
    // stage: the wave's private block (>= 1024 floats); lane = (row lane rrow = lane >> 2, chunk c4 = lane & 3)
    __device__ __forceinline__ void finish(const ConvStats& s, float* stage, int lane, int g, int cout) {
        if (s.dbg & 4096) return;
        const double q0 = MODE == STAT_FWD ? STAT_Q_FWD0 : STAT_Q_BWD, q1 = MODE == STAT_FWD ? STAT_Q_FWD1 : STAT_Q_BWD;
        const int rrow = lane >> 2, c4 = lane & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            __builtin_amdgcn_wave_barrier();
            f32x4* w = reinterpret_cast<f32x4*>(stage + (rrow * 4 + c4) * 16);
            w[0] = f32x4{s0[j][0], s1[j][0], s0[j][1], s1[j][1]};
            w[1] = f32x4{s0[j][2], s1[j][2], s0[j][3], s1[j][3]};
            w[2] = f32x4{s0[j][4], s1[j][4], s0[j][5], s1[j][5]};
            w[3] = f32x4{s0[j][6], s1[j][6], s0[j][7], s1[j][7]};
            __builtin_amdgcn_wave_barrier();
            const int oc4 = lane >> 4, oe = (lane >> 1) & 7, ok = lane & 1;     // this lane's output: (chunk, channel, which sum)
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += stage[(r * 4 + oc4) * 16 + oe * 2 + ok];
            const int col = j * 32 + oc4 * 8 + oe;
            if (col < cout && t != 0.f) {
                const double q = ok ? q1 : q0;
                atomicAdd(s.seg[0].acc + ((size_t)g * cout + col) * 2 + ok, rint((double)t * q) / q);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
};
// one segment with an accumulator over every output column: what the strip kernels fold
static inline bool stat_strip_ok_host(const ConvStats& s, int cout) {
    return s.nseg == 1 && s.seg[0].c0 == 0 && s.seg[0].c1 == cout && s.seg[0].acc != nullptr && cout % 8 == 0 && s.G >= 1;
}

// Workgroup epilogue (after a barrier behind the last flush): one thread per tile column adds the table to the fp64
// accumulators, partials rounded to the grid first.  slots_used: slots the tile's rows actually reach.
template <int MODE>
__device__ __forceinline__ void stat_publish(const ConvStats& s, const char* smem, int tid, int bn, int n0, int cout, int b0,
                                             int slots_used) {
    if (tid >= bn || (s.dbg & 2048)) return;
    const int col = n0 + tid;
    if (col >= cout) return;
    const int k = stat_seg_of(s, col);
    if (k < 0) return;
    const long long* sums = reinterpret_cast<const long long*>(smem + s.lds_off);
    const int cs = s.seg[k].c1 - s.seg[k].c0;
    const double r0 = 1.0 / (MODE == STAT_FWD ? STAT_Q_FWD0 : STAT_Q_BWD), r1 = 1.0 / (MODE == STAT_FWD ? STAT_Q_FWD1 : STAT_Q_BWD);
    for (int slot = 0; slot < slots_used; ++slot) {
        const int g = (b0 + slot) % s.G;
        const long long a = sums[((size_t)slot * bn + tid) * 2], b = sums[((size_t)slot * bn + tid) * 2 + 1];
        double* dst = s.seg[k].acc + ((size_t)g * cs + (col - s.seg[k].c0)) * 2;
        if (a) atomicAdd(dst, (double)a * r0);                       // (integer x power of two: exact)
        if (b) atomicAdd(dst + 1, (double)b * r1);
    }
}
#endif

}  // namespace gvconv
