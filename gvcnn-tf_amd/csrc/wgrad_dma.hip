// wgrad_dma.hip — filter gradient on 16-bit storage with LDS-DMA operand staging.
//
//   dW[tap][ci][co] += sum_m X[shift_tap(m)][ci] * dZ[m][co]          (train_utils.py:217-259)
//
// Same decomposition as conv_wgrad_lp (train_lp.hip): one TN GEMM per filter tap, a workgroup owns a (64 TI) x (64 TO)
// tile of one tap and a slice of the pixels, slices combine with fp32 atomics, 2 x 2 waves of TI x TO accumulators,
// fragments by ds_read_b64_tr_b16 (the reduction axis — pixels — is the strided one of both operands).  What changes
// is how the operands reach LDS: global_load_lds_dwordx4 through a ring of ST stages of 32 pixels with counted vmcnt
// waits (conv_dma.hip's pipeline) instead of global -> registers -> ds_write with one stage of look-ahead, and the
// pixel -> (image, row, column) walk is incremental (two small exact magic divisions per stage) instead of two 64-bit
// integer divisions per 16-byte load.  conv_wgrad_lp spent its time there: 8 MFMAs per barrier behind ~200 VALU
// instructions of address arithmetic, 11.7 % matrix-pipe busy.
//
// LDS image of one stage: X [32 pixels][P_x bytes] and dZ [32 pixels][P_z bytes], P = 128 or 256 bytes of channels.
// A transposed read touches, per 32-lane service group, 4 consecutive pixel rows x 64 contiguous bytes; with rows that
// are a multiple of 128 bytes apart they would share banks, so the 64-byte segments of a row are XOR-swizzled by the row
// ((row>>1)&1 for 128-byte rows, row&3 for 256-byte rows).  One DMA instruction writes 1 KiB = 8 or 4 whole rows
// lane-linearly, so the swizzle is applied to the SOURCE channel offset.
#include <type_traits>

#include "lowp.h"

namespace gvconv {
const void* dma_zero_page();
}

namespace {

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ f32x16 wmfma16(ws16x8 a, ws16x8 b, f32x16 c) {
    if constexpr (std::is_same<T, __bf16>::value)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, a), __builtin_bit_cast(wbf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8, a), __builtin_bit_cast(wf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ ws16x4 wlds_read_tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((ws16x4 __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ void wdma16(const char* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wwait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct WgradGeo {
    const unsigned short* x;
    const unsigned short* dz;
    GvDw dw;
    const void* zeros;
    int x_ld, dz_ld, nb, ih, iw, cin, kh, kw, stride, pad_t, pad_l, oh, ow, cout;
    int M, m_per_block;
    unsigned magic_ow, magic_oh;       // ceil(2^32 / d): exact for the small dividends of the incremental pixel walk
};

// (384-byte rows — 192-channel sides — are 96 banks apart: rows r and r + 2 of a read's four rows would share banks;
// swapping the 64-byte segments pairwise on every second row pair moves them 16 banks apart, and 0..5 stays 0..5)
template <int P> __device__ __forceinline__ int seg_key(int row) { return P == 256 ? (row & 3) : ((row >> 1) & 1); }

// PT: pixels per stage (32; 64 since round 4: half the barriers and counted waits per MFMA — a wave runs 8 (18) MFMAs
// between two barriers at 32 pixels and 128 (192) channel sides — for twice the LDS per stage)
template <typename T, int TI, int TO, int ST, int PT = 32>
__global__ __launch_bounds__(256) void conv_wgrad_dma(const WgradGeo g) {
    constexpr int BI = 64 * TI, BO = 64 * TO;
    constexpr int PX = 2 * BI, PZ = 2 * BO;                    // row bytes
    constexpr int X_BYTES = PT * PX, Z_BYTES = PT * PZ, STAGE = X_BYTES + Z_BYTES;
    constexpr int UX = X_BYTES / 1024, UZ = Z_BYTES / 1024;    // DMA instructions per stage
    constexpr int UXW = (UX + 3) / 4, UZW = (UZ + 3) / 4;      // per wave
    constexpr int LPT = UXW + UZW;
    static_assert(TI >= 1 && TI <= 3 && TO >= 1 && TO <= 3, "64-, 128- or 192-channel sides");
    static_assert(ST >= 2 && ST <= 4 && (ST - 1) * LPT < 64, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int ntile_co = (g.cout + BO - 1) / BO, ntile_ci = (g.cin + BI - 1) / BI;
    const int tiles = ntile_co * ntile_ci * g.kh * g.kw;
    const int logical = gv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int b = logical % tiles;
    const int slice = logical / tiles;
    const int tco = b % ntile_co; b /= ntile_co;
    const int tci = b % ntile_ci; b /= ntile_ci;
    const int fr = b / g.kw, fs = b - fr * g.kw;
    const int tap = b;
    const int ci0 = tci * BI, co0 = tco * BO;
    const int m0 = slice * g.m_per_block;
    const int m1 = min(m0 + g.m_per_block, g.M);
    const int nst = (m1 - m0 + PT - 1) / PT;

    const char* xb = reinterpret_cast<const char*>(g.x);
    const char* zb = reinterpret_cast<const char*>(g.dz);
    const char* zero_page = reinterpret_cast<const char*>(g.zeros);

    // ---- loader slots.  X: instruction u covers stage rows [u*RX, (u+1)*RX), lane -> (row, 16-byte unit of the row)
    // (lane -> (row, unit) through the stage-linear 16-byte index u * 64 + lane: a 384-byte row is 24 units, so an
    // instruction's 64 lanes do not cover whole rows there)
    constexpr int LX = PX / 16, LZ = PZ / 16;                  // 16-byte units (lanes) per row
    auto x_rowof = [&](int u) { return (u * 64 + lane) / LX; };
    auto x_unitof = [&](int u) { return (u * 64 + lane) % LX; };
    auto z_rowof = [&](int u) { return (u * 64 + lane) / LZ; };
    auto z_unitof = [&](int u) { return (u * 64 + lane) % LZ; };
    int x_u[UXW], x_n[UXW], x_oy[UXW], x_ox[UXW], x_choff[UXW];
    bool x_chok[UXW];
    const int ohow = g.oh * g.ow;
#pragma unroll
    for (int s = 0; s < UXW; ++s) {
        int u = wave + s * 4;
        u = u < UX ? u : UX - 1;                               // surplus slots re-load the last block (same bytes)
        x_u[s] = u;
        const int row = x_rowof(u), unit = x_unitof(u);
        const int lseg = (unit >> 2) ^ seg_key<PX>(row);
        const int ch = (lseg * 4 + (unit & 3)) * 8;            // channel of this lane's 8 values inside the tile
        x_choff[s] = (ci0 + ch) * 2;
        x_chok[s] = ci0 + ch < g.cin;
        const int m = m0 + row;                                // pixel of this row in stage 0 (may be >= m1: zero page)
        const int n = m / ohow, rem = m - n * ohow;
        x_n[s] = n;
        x_oy[s] = rem / g.ow;
        x_ox[s] = rem - x_oy[s] * g.ow;
    }
    int z_u[UZW], z_row[UZW], z_choff[UZW];
    bool z_chok[UZW];
#pragma unroll
    for (int s = 0; s < UZW; ++s) {
        int u = wave + s * 4;
        u = u < UZ ? u : UZ - 1;
        z_u[s] = u;
        const int row = z_rowof(u), unit = z_unitof(u);
        const int lseg = (unit >> 2) ^ seg_key<PZ>(row);
        const int ch = (lseg * 4 + (unit & 3)) * 8;
        z_row[s] = row;
        z_choff[s] = (co0 + ch) * 2;
        z_chok[s] = co0 + ch < g.cout;
    }
    int issued_stage = 0;                                      // next stage index to issue
    auto issue = [&]() {                                       // stage `issued_stage` into ring slot issued_stage % ST
        char* sb = smem + (issued_stage % ST) * STAGE;
        const int mt = m0 + issued_stage * PT;
#pragma unroll
        for (int s = 0; s < UXW; ++s) {
            const int row = x_rowof(x_u[s]);
            const int iy = x_oy[s] * g.stride + fr - g.pad_t, ix = x_ox[s] * g.stride + fs - g.pad_l;
            const bool ok = x_chok[s] && mt + row < m1 && (unsigned)iy < (unsigned)g.ih && (unsigned)ix < (unsigned)g.iw;
            const size_t off = ((size_t)((unsigned)(x_n[s] * g.ih + iy) * (unsigned)g.iw + (unsigned)ix)) * (size_t)g.x_ld * 2 + x_choff[s];
            wdma16(ok ? xb + off : zero_page, sb + x_u[s] * 1024);
            // this row's pixel in the next stage: + PT pixels (two exact small divisions)
            const unsigned ox2 = (unsigned)x_ox[s] + PT;
            const unsigned cy = __umulhi(ox2, g.magic_ow);
            x_ox[s] = (int)(ox2 - cy * (unsigned)g.ow);
            const unsigned oy2 = (unsigned)x_oy[s] + cy;
            const unsigned cn = __umulhi(oy2, g.magic_oh);
            x_oy[s] = (int)(oy2 - cn * (unsigned)g.oh);
            x_n[s] += (int)cn;
        }
#pragma unroll
        for (int s = 0; s < UZW; ++s) {
            const int m = mt + z_row[s];
            const bool ok = z_chok[s] && m < m1;
            wdma16(ok ? zb + (size_t)m * g.dz_ld * 2 + z_choff[s] : zero_page, sb + X_BYTES + z_u[s] * 1024);
        }
        ++issued_stage;
    };

    // ---- transposed fragment reads: lane -> (pixel row, channel) inside a (16-pixel, 32-channel) operand block
    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int row_l = 8 * (g16 >> 1) + q;                      // second read: + 4
    const int col_l = 16 * (g16 & 1) + 4 * p4;                 // channel within the 32-channel block
    int xo[TI][2], zo[TO][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = row_l + 4 * j;
#pragma unroll
        for (int t = 0; t < TI; ++t) {
            const int cb = 2 * ((wi * TI + t) * 32 + col_l);   // byte column in the X row
            xo[t][j] = row * PX + (((cb >> 6) ^ seg_key<PX>(row)) << 6) + (cb & 63);
        }
#pragma unroll
        for (int u = 0; u < TO; ++u) {
            const int cb = 2 * ((wj * TO + u) * 32 + col_l);
            zo[u][j] = X_BYTES + row * PZ + (((cb >> 6) ^ seg_key<PZ>(row)) << 6) + (cb & 63);
        }
    }

    f32x16 acc[TI][TO];
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int u = 0; u < TO; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // ---- ring: stages 0 .. ST-2 in flight, then per stage: wait (counted), barrier, issue stage i+ST-1, compute stage i
#pragma unroll
    for (int t = 0; t < ST - 1; ++t)
        if (t < nst) issue();
    for (int i = 0; i < nst; ++i) {
        const int last_issued = issued_stage - 1;              // stages <= last_issued are in flight or landed
        const int younger = last_issued - i;                   // issued after stage i
        if (ST >= 4 && younger >= 2) wwait_vm<(ST >= 4 ? 2 * LPT : 0)>();
        else if (ST >= 3 && younger >= 1) wwait_vm<(ST >= 3 ? LPT : 0)>();
        else wwait_vm<0>();
        __builtin_amdgcn_s_barrier();                          // stage i is in LDS for everyone; everyone is done with stage i-1
        if (issued_stage < nst) issue();                       // into the slot stage i-1 occupied
        const char* sb = smem + (i % ST) * STAGE;
#pragma unroll
        for (int k = 0; k < PT; k += 16) {
            ws16x8 av[TI], bv[TO];
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const ws16x4 lo = wlds_read_tr(sb + xo[t][0] + k * PX), hi = wlds_read_tr(sb + xo[t][1] + k * PX);
                av[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int u = 0; u < TO; ++u) {
                const ws16x4 lo = wlds_read_tr(sb + zo[u][0] + k * PZ), hi = wlds_read_tr(sb + zo[u][1] + k * PZ);
                bv[u] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int t = 0; t < TI; ++t)
#pragma unroll
                for (int u = 0; u < TO; ++u) acc[t][u] = wmfma16<T>(av[t], bv[u], acc[t][u]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of stage i are done before the next barrier
    }

    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int u = 0; u < TO; ++u) {
        const int col = co0 + (wj * TO + u) * 32 + li;
        if (col >= g.cout) continue;
#pragma unroll
        for (int t = 0; t < TI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (wi * TI + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < g.cin) gv_dw_put(g.dw, slice, ((size_t)tap * g.cin + ci) * g.cout + col, acc[t][u][r]);
            }
    }
}

#ifndef GV_KERNEL_ONLY     // (single instantiations of the kernel above for instruction counts: no launchers)
unsigned wmagic(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <typename T, int TI, int TO, int ST, int PT = 32>
int launch_wgrad_dma(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld, const GvDw& dw, int64_t target,
                     hipStream_t st) {
    constexpr int BI = 64 * TI, BO = 64 * TO;
    const int64_t M = (int64_t)d->nb * d->oh * d->ow;
    if (M >= 0x7fffffff || (int64_t)d->nb * d->ih * d->iw * d->x_ld * 2 >= 0x7fffffffffffll) return GV_E_UNSUPPORTED;
    if (d->ow >= 32768 || d->oh >= 32768) return GV_E_UNSUPPORTED;
    // ceil(2^32 / 1) does not fit 32 bits (wmagic(1) == 0: the pixel walk would never carry into oy / n and every stage
    // after the first would read the zero page): 1-wide / 1-high maps go to the tap-per-workgroup tiles
    if (d->ow < 2 || d->oh < 2) return GV_E_UNSUPPORTED;
    WgradGeo g;
    g.x = (const unsigned short*)x;
    g.dz = (const unsigned short*)dz;
    g.zeros = gvconv::dma_zero_page();
    if (!g.zeros) return GV_E_UNSUPPORTED;
    g.x_ld = d->x_ld; g.dz_ld = dz_ld; g.nb = d->nb; g.ih = d->ih; g.iw = d->iw; g.cin = d->cin; g.kh = d->kh; g.kw = d->kw;
    g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l; g.oh = d->oh; g.ow = d->ow; g.cout = d->cout;
    g.M = (int)M;
    g.magic_ow = wmagic(d->ow);
    g.magic_oh = wmagic(d->oh);
    const int tiles = d->kh * d->kw * ((d->cin + BI - 1) / BI) * ((d->cout + BO - 1) / BO);
    int64_t splits = (target + tiles - 1) / tiles;
    const int64_t max_splits = (M + 511) / 512;                  // at least 512 pixels per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    splits = gv_dw_clamp(dw, elems, splits);
    int64_t per = (M + splits - 1) / splits;
    per = (per + PT - 1) / PT * PT;
    splits = (M + per - 1) / per;
    if ((int64_t)tiles * splits > 0x7fffffff) return GV_E_UNSUPPORTED;
    g.m_per_block = (int)per;
    g.dw = gv_dw_sink(dw, elems, splits);
    const size_t lds = (size_t)ST * PT * 2 * (BI + BO);
    auto kern = &conv_wgrad_dma<T, TI, TO, ST, PT>;
    if (lds > 64 * 1024) {
        if (!GV_BIG_LDS_OK(kern, 160 * 1024)) return GV_E_UNSUPPORTED;      // (per device)
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * splits)), dim3(256), lds, st, g);
    GV_LAUNCH_CHECK();
    return gv_dw_finish(dw, elems, splits, st);
}

}  // namespace

namespace gvlp {

// k = 0..11: (TI, TO) in {1,2}^2 x workgroup target 1024 / 2048 / 4096 on a four-stage ring (16 - 64 KB: two to four
// workgroups per CU); k = 12..23: the same on TWO stages (8 - 32 KB: five and more per CU — as in the forward kernels a
// resident neighbour covers a workgroup's barriers and its atomic epilogue better than ring depth does)
// k = 24..33 (round 3): 192-channel sides on two stages — (TI, TO) = (3,3), (3,1), (1,3), (3,2), (2,3) at 1024 then 2048
// workgroups.  A 192-channel side on 128-wide tiles is one full and one half-empty tile: a 192 x 192 layer (Mixed_6e's
// 1x7 / 7x1 pairs, Mixed_7a) did 16/9 of its work, loads and instructions included.
// k = 34..42 (round 4): the two-stage tiles (TI, TO) in {1,2}^2 and the 192-channel tiles at a target of 512 workgroups.  With
// the slices STORED and reduced (gv_conv2d_wgrad_ws) every workgroup writes its whole fp32 tile and the reduce reads it
// back: on the 12 x 12 maps (7 tiles of 192 x 192 x 4 B over ~100 slices) that traffic is 2.6x the operands'.
// k = 43..60 (round 4): 64 pixels per stage, two stages — (TI, TO) as q above — at 1024 then 2048 workgroups.
int wgrad_dma_num_cfgs() { return 61; }

int conv_wgrad_dma_launch(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld, const GvDw& dw, int k, hipStream_t st) {
    if (k < 0 || k >= 61) return GV_E_BADARG;
    if (k >= 43) {
        const int q = (k - 43) % 9;
        const int64_t t6 = k - 43 < 9 ? 1024 : 2048;
#define GV_WD6(T)                                                                                       \
    switch (q) {                                                                                        \
        case 0: return launch_wgrad_dma<T, 1, 1, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        case 1: return launch_wgrad_dma<T, 2, 1, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        case 2: return launch_wgrad_dma<T, 1, 2, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        case 3: return launch_wgrad_dma<T, 2, 2, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        case 4: return launch_wgrad_dma<T, 3, 3, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        case 5: return launch_wgrad_dma<T, 3, 1, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        case 6: return launch_wgrad_dma<T, 1, 3, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        case 7: return launch_wgrad_dma<T, 3, 2, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                   \
        default: return launch_wgrad_dma<T, 2, 3, 2, 64>(d, x, dz, dz_ld, dw, t6, st);                  \
    }
        if (d->dtype == GV_BF16) { GV_WD6(__bf16) }
        if (d->dtype == GV_F16) { GV_WD6(_Float16) }
#undef GV_WD6
        return GV_E_UNSUPPORTED;
    }
    if (k >= 34) {
        const int q = k - 34;
#define GV_WD5(T)                                                                                       \
    switch (q) {                                                                                        \
        case 0: return launch_wgrad_dma<T, 1, 1, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        case 1: return launch_wgrad_dma<T, 2, 1, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        case 2: return launch_wgrad_dma<T, 1, 2, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        case 3: return launch_wgrad_dma<T, 2, 2, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        case 4: return launch_wgrad_dma<T, 3, 3, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        case 5: return launch_wgrad_dma<T, 3, 1, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        case 6: return launch_wgrad_dma<T, 1, 3, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        case 7: return launch_wgrad_dma<T, 3, 2, 2>(d, x, dz, dz_ld, dw, 512, st);                      \
        default: return launch_wgrad_dma<T, 2, 3, 2>(d, x, dz, dz_ld, dw, 512, st);                     \
    }
        if (d->dtype == GV_BF16) { GV_WD5(__bf16) }
        if (d->dtype == GV_F16) { GV_WD5(_Float16) }
#undef GV_WD5
        return GV_E_UNSUPPORTED;
    }
    if (k >= 24) {
        const int q = (k - 24) % 5;
        const int64_t t3 = k - 24 < 5 ? 1024 : 2048;
#define GV_WD3(T)                                                                                       \
    switch (q) {                                                                                        \
        case 0: return launch_wgrad_dma<T, 3, 3, 2>(d, x, dz, dz_ld, dw, t3, st);                       \
        case 1: return launch_wgrad_dma<T, 3, 1, 2>(d, x, dz, dz_ld, dw, t3, st);                       \
        case 2: return launch_wgrad_dma<T, 1, 3, 2>(d, x, dz, dz_ld, dw, t3, st);                       \
        case 3: return launch_wgrad_dma<T, 3, 2, 2>(d, x, dz, dz_ld, dw, t3, st);                       \
        default: return launch_wgrad_dma<T, 2, 3, 2>(d, x, dz, dz_ld, dw, t3, st);                      \
    }
        if (d->dtype == GV_BF16) { GV_WD3(__bf16) }
        if (d->dtype == GV_F16) { GV_WD3(_Float16) }
#undef GV_WD3
        return GV_E_UNSUPPORTED;
    }
    const int shape = k % 4 + (k >= 12 ? 4 : 0);
    const int64_t target = 1024ll << ((k % 12) / 4);
#define GV_WD(T)                                                                                        \
    switch (shape) {                                                                                    \
        case 0: return launch_wgrad_dma<T, 1, 1, 4>(d, x, dz, dz_ld, dw, target, st);                   \
        case 1: return launch_wgrad_dma<T, 2, 1, 4>(d, x, dz, dz_ld, dw, target, st);                   \
        case 2: return launch_wgrad_dma<T, 1, 2, 4>(d, x, dz, dz_ld, dw, target, st);                   \
        case 3: return launch_wgrad_dma<T, 2, 2, 4>(d, x, dz, dz_ld, dw, target, st);                   \
        case 4: return launch_wgrad_dma<T, 1, 1, 2>(d, x, dz, dz_ld, dw, target, st);                   \
        case 5: return launch_wgrad_dma<T, 2, 1, 2>(d, x, dz, dz_ld, dw, target, st);                   \
        case 6: return launch_wgrad_dma<T, 1, 2, 2>(d, x, dz, dz_ld, dw, target, st);                   \
        default: return launch_wgrad_dma<T, 2, 2, 2>(d, x, dz, dz_ld, dw, target, st);                  \
    }
    if (d->dtype == GV_BF16) { GV_WD(__bf16) }
    if (d->dtype == GV_F16) { GV_WD(_Float16) }
#undef GV_WD
    return GV_E_UNSUPPORTED;
}

}  // namespace gvlp
#else
}  // namespace
#endif
