// conv_dma.hip — implicit-GEMM convolution whose operands travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write, no VALU between the load and the LDS image), through a
// ring of ST LDS stages with counted vmcnt waits and one raw s_barrier per k-tile.
//
// Both operands are "rows of 16-value groups", each group NP planes x 32 bytes:
//   NP = 1   16-bit storage (GV_BF16 / GV_F16): plain NHWC activations [pixel][channel] and filters [cout][Kpad];
//   NP = 3   fp32 values kept as three bf16 planes a = a0 + a1 + a2 (GV_MATH_BF16X3 on "P3" storage):
//            activations [pixel][channel/16][plane][16], filters [cout][k/16][plane][16]; six plane products per
//            MFMA block give fp32-level accuracy (conv_bf16s.hip explains the arithmetic), and because the PRODUCER's
//            epilogue split the activation once, the loader here moves bytes only.
//
// LDS image of one stage: per operand and plane a [rows][KT values] sub-image (KT = 32 for NP = 1: 64-byte rows, 16 rows
// per DMA instruction; KT = 16 for NP = 3: 32-byte rows, 32 rows per instruction).  One global_load_lds_dwordx4 writes
// 1 KiB = wave-uniform base + lane * 16, i.e. one whole row block; the lane -> (row, 16-byte chunk) map is therefore
// fixed and the bank-conflict swizzle is applied to the SOURCE address: lane (row r, physical chunk c) fetches logical
// chunk c ^ swz(r), and the MFMA fragment read of logical chunk q of row r reads physical chunk q ^ swz(r)
// (swz(r) = (r >> 2) & 3 for 64-byte rows, (r >> 3) & 1 for 32-byte rows: every 16-lane ds_read_b128 service group
// then touches 16 distinct 4-bank groups).  Padding taps of the im2col gather read a zero page.
//
// Pipeline per k-tile kt:  wait until this wave's loads of tile kt have landed (vmcnt counted: the ST-2 younger tiles
// stay in flight) -> s_barrier (every wave's part of tile kt is in LDS, and every wave is done reading tile kt-1) ->
// issue the loads of tile kt+ST-1 into the stage tile kt-1 occupied -> read the fragments of tile kt -> MFMAs.
#include <mutex>
#include <type_traits>

#include "conv_common.h"
#include "conv_lp_epi.h"
#include "conv_x3_epi.h"

namespace {

// KTX: k-tile depth of the 16-bit form (0 = the default below).  KT = 64 makes an LDS row 128 bytes = ONE cache line of
// the source tensor per row and k-tile: a DMA instruction then fetches 8 whole lines instead of 16 half lines (the
// second half of each is fetched a whole tap sweep later, long after the line left the 32 KiB L1) — the programming
// guide prices half-line, fragment-shaped staging loads at +18 ... 45 % with the vector-memory path twice as busy for
// the same L2 / HBM traffic.  Needs cin % 64 == 0 (whole 64-channel chunks inside one filter tap).
template <int NP, int KTX = 0> struct DmaGeom {
    static constexpr int KT = KTX ? KTX : (NP == 1 ? 32 : 16);       // k-tile depth
    static constexpr int RBYTES = KT * 2;               // one plane of one LDS row
    static constexpr int RPI = 1024 / RBYTES;           // rows per DMA instruction
    static constexpr int CPR = RBYTES / 16;             // 16-byte chunks per row
    static constexpr int G16 = NP * 32;                 // bytes of one 16-value group in global memory (all planes)
    static constexpr int KSTEPS = KT / 16;              // MFMA k-steps per k-tile
    // XOR swizzle of a row's 16-byte chunks (applied to the SOURCE address by the loader, to the read address by the
    // fragment reads): 32-byte rows (r >> 3) & 1, 64-byte rows (r >> 2) & 3, 128-byte rows (r >> 1) & 7 — with the last, the
    // 16 rows of every ds_read_b128 service group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and their upper-half twins)
    // land on 16 distinct 4-bank groups: 8 distinct chunks among the even rows, the same 8 among the odd rows, and a
    // row's parity selects the half of the 256-byte bank row.
    __host__ __device__ static constexpr int swz(int r) { return CPR == 8 ? ((r >> 1) & 7) : (CPR == 4 ? ((r >> 2) & 3) : ((r >> 3) & 1)); }
};

__device__ __forceinline__ void dma16(const char* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// plane products of the 3-plane form, small terms first (as conv_bf16s.hip)
__host__ __device__ constexpr int dprod_count(int np) { return np == 3 ? 6 : 1; }
__host__ __device__ constexpr int dprod_pa(int np, int t) { return np == 3 ? (t == 0 ? 2 : (t == 2 || t == 3 ? 1 : 0)) : 0; }
__host__ __device__ constexpr int dprod_pb(int np, int t) { return np == 3 ? (t == 1 ? 2 : (t == 2 || t == 4 ? 1 : 0)) : 0; }

// EPI: 0 = 16-bit output through the LDS-staged epilogue (conv_lp_epi.h); 1 = fp32 output straight from the accumulators
// (conv_common.h); 2 = fp32 or three-plane (P3) output through the staged epilogue of conv_x3_epi.h
// byte offset of the epilogue's scale / shift table: behind the ring and the per-wave staging blocks (which alias the ring)
template <int NP, int WM, int WN, int TM, int TN, int ST, int EPI, int KTX = 0>
constexpr int dma_ss_off() {
    constexpr int ring = ST * NP * (WM * TM * 32 + WN * TN * 32) * DmaGeom<NP, KTX>::RBYTES;
    constexpr int epi = EPI == 1 ? 0 : WM * WN * (EPI == 0 ? EpiGeom<TN>::BYTES : X3EpiGeom<TN>::BYTES);
    return ((ring > epi ? ring : epi) + 15) / 16 * 16;
}
// ... unless the table's bytes would cost a resident workgroup (128 x 192 on four 16-bit stages is exactly half a CU's LDS)
template <int NP, int WM, int WN, int TM, int TN, int ST, int EPI, int KTX = 0>
constexpr bool dma_use_ss() {
    constexpr int base = dma_ss_off<NP, WM, WN, TM, TN, ST, EPI, KTX>(), with = base + 16 * WN * TN * 32, cu = 160 * 1024;
    return EPI != 1 && cu / base == cu / with;
}

// ... in which case the table is published LATE into the ring the main loop has freed, behind the waves' staging blocks
// (where it fits there: the 128 x 128 tile on two stages stages exactly a ring's worth)
template <int NP, int WM, int WN, int TM, int TN, int ST, int EPI, int KTX = 0>
constexpr int dma_late_off() { return WM * WN * (EPI == 0 ? EpiGeom<TN>::BYTES : X3EpiGeom<TN>::BYTES); }
template <int NP, int WM, int WN, int TM, int TN, int ST, int EPI, int KTX = 0>
constexpr bool dma_late_ss() {
    constexpr int ring = ST * NP * (WM * TM * 32 + WN * TN * 32) * DmaGeom<NP, KTX>::RBYTES;
    return EPI != 1 && !dma_use_ss<NP, WM, WN, TM, TN, ST, EPI, KTX>() &&
           dma_late_off<NP, WM, WN, TM, TN, ST, EPI, KTX>() + 16 * WN * TN * 32 <= ring;
}

// STATS (16-bit output only): train-mode BatchNorm sums of the stored tensor folded into the epilogue (conv_stats.h).
template <typename T, int NP, int WM, int WN, int TM, int TN, int ST, int EPI, int STATS = 0, int KTX = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_dma(const ConvArgs a) {
    using G = DmaGeom<NP, KTX>;
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int KT = G::KT, RBYTES = G::RBYTES, RPI = G::RPI, CPR = G::CPR, G16 = G::G16, KS = G::KSTEPS;
    constexpr int UA = BM / RPI, UB = BN / RPI;                    // row blocks (one DMA instruction per plane each)
    constexpr int UAW = (UA + NW - 1) / NW, UBW = (UB + NW - 1) / NW;
    constexpr int A_PLANE = BM * RBYTES, B_PLANE = BN * RBYTES;
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    constexpr int LPT = NP * (UAW + UBW);                          // DMA instructions per wave per k-tile
    static_assert(ST >= 2 && ST <= 4, "2..4 LDS stages");
    static_assert((ST - 1) * LPT < 64, "vmcnt range");
    constexpr int NPR = dprod_count(NP);

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef GV_PHASE_TIMES                                              // profiling build only (tools/phase_times.py): per-workgroup
    unsigned long long gv_pt[7], gv_pi[4] = {0, 0, 0, 0};                                   // timestamps of the kernel's phases
#define GV_PT(i) gv_pt[i] = __builtin_amdgcn_s_memtime()
#else
#define GV_PT(i)
#endif
    GV_PT(0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    // The epilogue's per-column constants (scale, shift, and scale2 / shift2 of a second output) live in LDS behind the
    // ring / staging area: requested here, written once the ring's first stages are on their way, read back by the staged
    // epilogue as 16-byte LDS reads.  Loaded from global memory inside the epilogue they cost every read-back block of
    // every tile one exposed memory round trip (16 - 32 scalar loads issued and awaited between the last MFMA and the
    // first store), which the short-K HBM-bound launches (ResNet conv3: K = 64) cannot hide behind anything.
    constexpr int SS_OFF = dma_ss_off<NP, WM, WN, TM, TN, ST, EPI, KTX>();
    static_assert(WM * WN * 64 >= WN * TN * 32, "one thread per tile column");
    constexpr bool USE_SS = dma_use_ss<NP, WM, WN, TM, TN, ST, EPI, KTX>();
    // ... in which case the table is published LATE, into the ring the main loop has freed (behind the waves' staging
    // blocks): the values wait in four registers meanwhile.  (Without a table every lane of the epilogue fetches its 16
    // constants per column block from global memory — 48 loads and ~250 address instructions per wave of a 128 x 192 tile.)
    constexpr bool LATE_SS = dma_late_ss<NP, WM, WN, TM, TN, ST, EPI, KTX>();
    constexpr int LATE_OFF = dma_late_off<NP, WM, WN, TM, TN, ST, EPI, KTX>();
    constexpr bool ANY_SS = USE_SS || LATE_SS;
    float* sstab = ANY_SS && !(a.dbg & 512) ? reinterpret_cast<float*>(smem + (LATE_SS ? LATE_OFF : SS_OFF)) : nullptr;   // dbg 512: constants from global (A/B)
    float ss_v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool ss_dual = EPI == 0 && a.y2 != nullptr && a.split == 0;
    if (ANY_SS && tid < WN * TN * 32) {
        const int lid0 = gv_xcd_remap(blockIdx.x, gridDim.x);
        const int cc = min((lid0 % a.tiles_n) * (WN * TN * 32) + tid, a.cout - 1);
        ss_v[0] = a.scale[cc];
        ss_v[1] = a.shift[cc];
        if (ss_dual) { ss_v[2] = a.scale2[cc]; ss_v[3] = a.shift2[cc]; }
    }
    auto ss_publish = [&]() {
        if (ANY_SS && sstab != nullptr && tid < WN * TN * 32) {
            constexpr int BN_ = WN * TN * 32;
            sstab[tid] = ss_v[0];
            sstab[BN_ + tid] = ss_v[1];
            if (ss_dual) { sstab[2 * BN_ + tid] = ss_v[2]; sstab[3 * BN_ + tid] = ss_v[3]; }
        }
    };
    // Eight waves = two per SIMD (waves w and w+4: a workgroup's waves go to the SIMDs in a cyclic order), and the k-tile
    // barrier makes both arrive at their MFMAs together: they then share the matrix pipe and idle together through the next
    // barrier / fragment reads / DMA issue.  A higher issue priority for one of the two staggers them inside a k-tile —
    // one runs its MFMAs while the other does everything else: -2 .. -3.5 % on the 256-row tiles (debug bit 256: off, A/B).
    if (NW == 8 && NP == 1 && wave >= 4 && !(a.dbg & 256)) __builtin_amdgcn_s_setprio(2);

    const int lid = gv_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % a.tiles_n;
    const int tile_m = lid / a.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    static_assert(STATS == 0 || EPI == 0, "BatchNorm sums / lean form: the 16-bit staged epilogue");
    constexpr bool HAS_SUMS = gvconv::stat_has(STATS);
    int st_b0 = 0;                                                 // STATS: image of the tile's first pixel
    if constexpr (HAS_SUMS) st_b0 = m0 / a.st.hw;

    // ---- loader state -------------------------------------------------------------------------------------------
    const int lrow = lane / CPR;                                   // row inside a row block
    const int pc = lane % CPR;                                     // physical chunk this lane's 16 bytes land in
    // logical chunk it must fetch: pc ^ swz(row).  Rows of a block are RPI-aligned, so only 128-byte rows (RPI = 8, swizzle
    // bits 1..3 of the row) see the block's index: odd row blocks flip chunk bit 2 = byte 64 of the source offset
    // (x_flip / b_flip below; a wave's blocks wave, wave + NW, ... share their parity while NW is even)
    const int lc = pc ^ G::swz(lrow);
    const int chunk_byte = (lc >> 1) * G16 + (lc & 1) * 16;        // inside one k-tile of a row (KT = 32: two groups)
    constexpr bool RB_SWZ = CPR == 8;
    const char* xb = reinterpret_cast<const char*>(a.x);
    const char* zero_page = reinterpret_cast<const char*>(a.zeros);
    const unsigned pix_bytes = (unsigned)a.x_ld * 2u * NP;         // one pixel of the input tensor
    const int ohow = a.oh * a.ow;

    int a_img[UAW], a_iy0[UAW], a_ix0[UAW];
    int a_rb[UAW];
    int a_flip[RB_SWZ ? UAW : 1];                                   // 128-byte rows: 64 where the row block is odd
#pragma unroll
    for (int i = 0; i < UAW; ++i) {
        int rb = wave + i * NW;
        rb = rb < UA ? rb : UA - 1;                                // surplus slots re-load the last block (same bytes)
        a_rb[i] = rb;
        if constexpr (RB_SWZ) a_flip[i] = (rb & 1) ? 64 : 0;
        int m = m0 + rb * RPI + lrow;
        m = m < a.M ? m : a.M - 1;                                 // rows past M: results never stored
        const int n = gv_div(m, a.y_div_img);
        const int rem = m - n * ohow;
        const int oy = gv_div(rem, a.y_div_row);
        const int ox = rem - oy * a.ow;
        a_img[i] = n * a.ih;
        a_iy0[i] = oy * a.stride - a.pad_t;
        a_ix0[i] = ox * a.stride - a.pad_l;
    }
    // filter tap (fr, fs) and channel fc of this lane's chunk in the NEXT tile to issue
    int fc = lc * 8, fs = 0, fr = 0;
    while (fc >= a.cin) { fc -= a.cin; if (++fs == a.kw) { fs = 0; ++fr; } }
    const char* a_ptr[UAW];
    bool a_ok[UAW];
    auto locate = [&]() {
#pragma unroll
        for (int i = 0; i < UAW; ++i) {
            const int iyn = a_iy0[i] + fr;
            const int ixn = a_ix0[i] + fs;
            const int dmask = (1 << a.dil_shift) - 1;              // zero-dilated input (stride-2 data gradient)
            const int iy = iyn >> a.dil_shift;
            const int ix = ixn >> a.dil_shift;
            const bool ok = iyn >= 0 && ixn >= 0 && ((iyn | ixn) & dmask) == 0 && iy < a.ih && ix < a.iw && fr < a.kh;
            a_ok[i] = ok;
            // (odd row blocks of 128-byte rows fetch the chunk 32 channels away inside the same 64-channel group: cin % 64 == 0)
            const int fce = RB_SWZ ? (fc ^ (a_flip[i] >> 1)) : fc;
            const size_t off = (size_t)((unsigned)(a_img[i] + iy) * (unsigned)a.iw + (unsigned)ix) * pix_bytes +
                               (size_t)((fce >> 4) * G16 + ((fce >> 3) & 1) * 16);
            a_ptr[i] = ok ? xb + off : zero_page;
        }
    };
    // Chunk-major order (a.korder, below) visits another tap of the SAME channel chunk every k-tile: the general locate()
    // — ~50 vector instructions and two divergent branches for a wave's row blocks — would run per k-tile, next to 12 - 16
    // MFMAs (a wave issues one instruction per 4 clocks: that is MFMA time).  For it, everything that depends on the row is
    // computed ONCE: the row's base address (tap (0, 0), this lane's chunk; may lie in front of the tensor: only added to)
    // and a bit per tap "inside the image"; a k-tile then adds one wave-uniform offset (tap + channel chunk, scalar ALU)
    // and tests one bit — 6 vector instructions per row block.  (Needs <= 32 taps and no dilation: ConvArgs::korder.)
    // The three per-row registers of the general form are REUSED for it (both forms live in one kernel, and their
    // registers would add up: six more than the 168 that let three of the two-stage workgroups share a CU):
    // (a_img, a_iy0) := the base address, a_ix0 := the tap mask.
    int q_fr = 0, q_fs = 0;                                        // chunk-major: tap of the next tile to issue (wave-uniform)
    unsigned q_chunk = 0;                                          //              byte offset of its channel chunk
    if (a.korder) {
        // (the taps inside the image form a rectangle [r_lo, r_hi) x [c_lo, c_hi): one row of bits, placed once per
        // filter row — kh iterations, not kh * kw)
        unsigned tapmask[UAW], rowbits[UAW];
        int r_lo[UAW], r_hi[UAW];
#pragma unroll
        for (int i = 0; i < UAW; ++i) {
            r_lo[i] = max(0, -a_iy0[i]);
            r_hi[i] = min(a.kh, a.ih - a_iy0[i]);
            const int c_lo = max(0, -a_ix0[i]), c_hi = min(a.kw, a.iw - a_ix0[i]);
            rowbits[i] = c_hi > c_lo ? (1u << c_hi) - (1u << c_lo) : 0u;
            tapmask[i] = 0u;
        }
        for (int r = 0; r < a.kh; ++r) {
#pragma unroll
            for (int i = 0; i < UAW; ++i)
                if (r >= r_lo[i] && r < r_hi[i]) tapmask[i] |= rowbits[i] << (r * a.kw);
        }
#pragma unroll
        for (int i = 0; i < UAW; ++i) {
            const long long pix0 = (long long)(a_img[i] + a_iy0[i]) * a.iw + a_ix0[i];
            const unsigned long long base = (unsigned long long)(xb + pix0 * (long long)pix_bytes + (RB_SWZ ? (chunk_byte ^ a_flip[i]) : chunk_byte));
            a_img[i] = (int)(unsigned)base;
            a_iy0[i] = (int)(unsigned)(base >> 32);
            a_ix0[i] = (int)tapmask[i];
        }
    }
    int kt_tap = 0;                                                // chunk-major: tap index of the next tile to issue
    auto locate_fast = [&]() {
        const unsigned soff = (unsigned)(q_fr * a.iw + q_fs) * pix_bytes + q_chunk;
        const unsigned bit = 1u << kt_tap;
#pragma unroll
        for (int i = 0; i < UAW; ++i) {
            const char* base = reinterpret_cast<const char*>(((unsigned long long)(unsigned)a_iy0[i] << 32) | (unsigned)a_img[i]);
            a_ok[i] = ((unsigned)a_ix0[i] & bit) != 0u;
            a_ptr[i] = a_ok[i] ? base + soff : zero_page;
        }
    };
    if (a.korder) locate_fast();
    else locate();
    // k-tile order.  Tap-major (k = tap * cin + channel, the packed filter's order): consecutive k-tiles walk the channels
    // of one tap, so a workgroup returns to the same input rows only after streaming its whole tile x cin — with 32
    // workgroups per XCD that working set outruns the 4 MiB L2 and every tap re-fetches the rows from the fabric (measured
    // 6.5x the input per 1x7 layer).  Chunk-major (a.korder: channel chunk outer, filter tap inner; needs KT | cin): the
    // taps of one chunk re-read the SAME few KB per workgroup back to back, and they hit L2.  Same products, another
    // summation order; the filter slice of (chunk, tap) is k-tile tap * cin/KT + chunk of the same packed filter.
    const int b_tap_step = (a.cin / 16) * G16;                     // bytes between the same chunk of consecutive taps
    auto advance = [&]() {
        if (a.korder) {
            if (++q_fs == a.kw) { q_fs = 0; ++q_fr; }
            if (++kt_tap == a.kh * a.kw) { kt_tap = 0; q_fs = 0; q_fr = 0; q_chunk += (KT / 16) * G16; }
            locate_fast();
            return;
        }
        fc += KT;
        if (fc >= a.cin) {
            do { fc -= a.cin; if (++fs == a.kw) { fs = 0; ++fr; } } while (fc >= a.cin);
            locate();
        } else {
#pragma unroll
            for (int i = 0; i < UAW; ++i) a_ptr[i] += a_ok[i] ? (KT / 16) * G16 : 0;
        }
    };
    const char* b_ptr[UBW];
    int b_rb[UBW];
    {
        const size_t row_bytes = (size_t)(a.Kpad / 16) * G16;
#pragma unroll
        for (int i = 0; i < UBW; ++i) {
            int rb = wave + i * NW;
            rb = rb < UB ? rb : UB - 1;
            b_rb[i] = rb;
            int n = n0 + rb * RPI + lrow;
            n = n < a.cout ? n : a.cout - 1;                       // columns past cout are never stored
            b_ptr[i] = reinterpret_cast<const char*>(a.w) + (size_t)n * row_bytes + (RB_SWZ && (rb & 1) ? (chunk_byte ^ 64) : chunk_byte);
        }
    }
    // DMA instruction d (0 .. LPT-1) of the tile at the current loader state: A units first, then B
    auto dma_one = [&](int d, char* sb) {
        if (d < NP * UAW) {
            const int i = d / NP, p = d % NP;
            dma16(a_ptr[i] + (a_ok[i] ? p * 32 : 0), sb + p * A_PLANE + a_rb[i] * 1024);
        } else {
            const int e = d - NP * UAW;
            const int i = e / NP, p = e % NP;
            dma16(b_ptr[i] + p * 32, sb + NP * A_PLANE + p * B_PLANE + b_rb[i] * 1024);
        }
    };
    auto step_state = [&]() {                                      // loader state -> next tile
        const bool wrap = a.korder && kt_tap + 1 == a.kh * a.kw;   // (read before advance() moves it)
        const int bstep = !a.korder ? (KT / 16) * G16 : (wrap ? (KT / 16) * G16 - (a.kh * a.kw - 1) * b_tap_step : b_tap_step);
        advance();
#pragma unroll
        for (int i = 0; i < UBW; ++i) b_ptr[i] += bstep;
    };
    auto issue = [&](int stage) {                                  // (prologue) the tile at the current loader state
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int d = 0; d < LPT; ++d) dma_one(d, sb);
        step_state();
    };

    // ---- fragments ------------------------------------------------------------------------------------------------
    const int fr_row = lane & 31, fr_h = lane >> 5;
    const int fsw = G::swz(fr_row);
    int foff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) foff[s] = ((s * 2 + fr_h) ^ fsw) * 16;
    const int a_frag = (wm * TM * 32 + fr_row) * RBYTES;
    const int b_frag = NP * A_PLANE + (wn * TN * 32 + fr_row) * RBYTES;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (NP == 1) {
        // ---- 16-bit storage: branch-free k-tile bodies (compile-time ISSUE / MORE / wait counts), the next k-step's
        // fragment reads issued one per MFMA behind the matrix instructions --------------------------------------
        // Fragments are double buffered in registers at K-STEP granularity (one k-step = 16 k-values = one MFMA per
        // accumulator and plane product): while the MFMAs of k-step q run, the fragments of k-step q+1 are already on their
        // way from LDS.  Inside a tile that needs no synchronisation; the first k-step of tile kt+1 is read right after the
        // barrier that publishes tile kt+1, which sits BEFORE the last k-step's MFMAs of tile kt — so the barrier wait, the
        // fragment-read latency and the DMA instructions of the tile ST ahead (issued after that barrier into the stage tile
        // kt just vacated, spread between the MFMAs) are all covered by matrix work of the same wave.
        constexpr int NMFS = NPR * TM * TN;                            // MFMAs per k-step
        constexpr int NRD = (TM + TN) * NP;                            // fragment reads per k-step
        constexpr int GAP = NMFS / (LPT + 1) > 0 ? NMFS / (LPT + 1) : 1;
        u32x4 fa[2][TM][NP], fb[2][TN][NP];
        // fragment read q (0 .. NRD-1) of k-step s of the tile in `stage`, into register set S
        auto read_one = [&](auto setc, const char* sb, int s, int q) {
            constexpr int S = decltype(setc)::value;
            if (q < TM * NP) {
                const int i = q / NP, p = q % NP;
                fa[S][i][p] = *reinterpret_cast<const u32x4*>(sb + p * A_PLANE + a_frag + i * 32 * RBYTES + foff[s]);
            } else {
                const int e = q - TM * NP;
                const int j = e / NP, p = e % NP;
                fb[S][j][p] = *reinterpret_cast<const u32x4*>(sb + p * B_PLANE + b_frag + j * 32 * RBYTES + foff[s]);
            }
        };
        auto read_frags = [&](auto setc, int stage, int s) {
    #pragma unroll
            for (int q = 0; q < NRD; ++q) read_one(setc, smem + stage * STAGE, s, q);
        };
        // The MFMAs of one k-step on register set S.  Between them, one per MFMA: the fragment reads of the NEXT k-step
        // (into set S^1, from `rsb` / k-step `rs`; READ) and the DMA instructions of the tile ST ahead (ISSUE, spread over
        // the k-step).  Reads issued BEHIND an MFMA have their latency covered by the matrix pipe; issued in front of the
        // k-step's first MFMA they cost the wave an LDS round trip per k-tile with nothing to overlap it.
        auto mfmas = [&](auto setc, auto readc, auto issuec, const char* rsb, int rs, char* nb) {
            constexpr int S = decltype(setc)::value;
            constexpr bool READ = decltype(readc)::value, ISSUE = decltype(issuec)::value;
            constexpr int RD_AT = NP == 3 ? -1 : 0;                    // (NP = 3: all reads first; one per MFMA spills there)
            if constexpr (READ && RD_AT < 0) {
    #pragma unroll
                for (int q = 0; q < NRD; ++q) read_one(std::integral_constant<int, S ^ 1>{}, rsb, rs, q);
            }
    #pragma unroll
            for (int m = 0; m < NMFS; ++m) {
                const int t = m / (TM * TN);
                const int i = (m / TN) % TM, j = m % TN;
                acc[i][j] = mfma16<T>(fa[S][i][dprod_pa(NP, t)], fb[S][j][dprod_pb(NP, t)], acc[i][j]);
                if constexpr (READ && RD_AT >= 0) {
                    if (m < NRD) read_one(std::integral_constant<int, S ^ 1>{}, rsb, rs, m);
                }
                if constexpr (ISSUE) {
                    if ((m + 1) % GAP == 0 && (m + 1) / GAP <= LPT) dma_one((m + 1) / GAP - 1, nb);
                }
            }
            if constexpr (READ && RD_AT >= 0) {
    #pragma unroll
                for (int q = NMFS; q < NRD; ++q) read_one(std::integral_constant<int, S ^ 1>{}, rsb, rs, q);
            }
            if constexpr (ISSUE) {
    #pragma unroll
                for (int d = NMFS / GAP; d < LPT; ++d) dma_one(d, nb);  // (fewer MFMAs than DMA instructions)
                step_state();
            }
        };
        const int ktiles = a.ktiles;
        using TT = std::true_type;
        using FF = std::false_type;
        // tile kt whose first k-step's fragments sit in register set PAR; ISSUE: the tile ST ahead exists and is issued
        // here; MORE: tile kt+1 exists (its first fragments are read here); YOUNGER: DMA tiles issued after tile kt+1 so far
        auto tile_step = [&](auto parc, auto issuec, auto morec, auto youngerc, int stage) {
            constexpr int PAR = decltype(parc)::value;
            constexpr bool ISSUE = decltype(issuec)::value, MORE = decltype(morec)::value;
            constexpr int YOUNGER = decltype(youngerc)::value;
            const int next = stage + 1 == ST ? 0 : stage + 1;
            const char* sb = smem + stage * STAGE;
    #pragma unroll
            for (int s = 0; s + 1 < KS; ++s) {                        // k-steps inside the tile: no synchronisation needed
                if (((PAR + s) & 1) == 0) mfmas(std::integral_constant<int, 0>{}, TT{}, FF{}, sb, s + 1, nullptr);
                else mfmas(std::integral_constant<int, 1>{}, TT{}, FF{}, sb, s + 1, nullptr);
            }
            constexpr int LASTSET = (PAR + KS - 1) & 1;
            if constexpr (MORE) {
                wait_vm<YOUNGER * LPT>();
                __builtin_amdgcn_s_barrier();                          // tile kt+1 is in LDS; every wave has read all of tile kt
            }
            mfmas(std::integral_constant<int, LASTSET>{}, std::integral_constant<bool, MORE>{},
                  std::integral_constant<bool, ISSUE>{}, smem + next * STAGE, 0, smem + stage * STAGE);
        };
        // run tiles [k0, k1) with compile-time flags; handles the set parity for odd KS by pairing tiles
        auto run_tiles = [&](auto issuec, auto morec, auto youngerc, int k0, int k1, int& stage, int& par) {
            for (int kt = k0; kt < k1; ++kt) {
                if (KS % 2 == 0 || par == 0) tile_step(std::integral_constant<int, 0>{}, issuec, morec, youngerc, stage);
                else tile_step(std::integral_constant<int, 1>{}, issuec, morec, youngerc, stage);
                if (KS % 2 != 0) par ^= 1;
                stage = stage + 1 == ST ? 0 : stage + 1;
            }
        };

        // ---- main loop ------------------------------------------------------------------------------------------------
        GV_PT(5);
    #pragma unroll
        for (int t = 0; t < ST; ++t) {
            if (t < ktiles) issue(t);
#ifdef GV_PHASE_TIMES
            gv_pi[t] = __builtin_amdgcn_s_memtime();
#endif
        }
        GV_PT(6);
        if constexpr (USE_SS) ss_publish();
        {
            const int younger = (ktiles < ST ? ktiles : ST) - 1;      // tiles issued after tile 0
            if (younger >= 3) wait_vm<(ST >= 4 ? 3 * LPT : 0)>();
            else if (younger == 2) wait_vm<(ST >= 3 ? 2 * LPT : 0)>();
            else if (younger == 1) wait_vm<LPT>();
            else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        GV_PT(1);
        read_frags(std::integral_constant<int, 0>{}, 0, 0);
        {
            int stage = 0, par = 0;
            // steady state: tile kt+ST exists -> after the barrier of tile kt, tiles kt+2 .. kt+ST-1 are younger than kt+1
            const int steady = ktiles - ST > 0 ? ktiles - ST : 0;
            run_tiles(TT{}, TT{}, std::integral_constant<int, ST - 2>{}, 0, steady, stage, par);
            // drain: nothing left to issue; the number of younger tiles shrinks to zero
            int kt = steady;
            if constexpr (ST >= 4) { if (kt < ktiles - 3) { run_tiles(FF{}, TT{}, std::integral_constant<int, 2>{}, kt, kt + 1, stage, par); ++kt; } }
            if constexpr (ST >= 3) { if (kt < ktiles - 2) { run_tiles(FF{}, TT{}, std::integral_constant<int, 1>{}, kt, kt + 1, stage, par); ++kt; } }
            if (kt < ktiles - 1) { run_tiles(FF{}, TT{}, std::integral_constant<int, 0>{}, kt, kt + 1, stage, par); ++kt; }
            run_tiles(FF{}, FF{}, std::integral_constant<int, 0>{}, kt, ktiles, stage, par);
        }

    } else {
        // ---- three-plane operands: the same pipeline with run-time tail handling (the branch-free form above makes one
        // basic block of 36-48 MFMAs + 18-21 reads per k-tile that the scheduler reorders into hundreds of spilled VGPRs)
        // Fragments are double buffered in registers at K-STEP granularity (one k-step = 16 k-values = one MFMA per
        // accumulator and plane product): while the MFMAs of k-step q run, the fragments of k-step q+1 are already on their
        // way from LDS.  Inside a tile that needs no synchronisation; the first k-step of tile kt+1 is read right after the
        // barrier that publishes tile kt+1, which sits BEFORE the last k-step's MFMAs of tile kt — so the barrier wait, the
        // fragment-read latency and the DMA instructions of the tile ST ahead (issued after that barrier into the stage tile
        // kt just vacated, spread between the MFMAs) are all covered by matrix work of the same wave.
        constexpr int NMFS = NPR * TM * TN;                            // MFMAs per k-step
        constexpr int GAP = NMFS / (LPT + 1) > 0 ? NMFS / (LPT + 1) : 1;
        u32x4 fa[2][TM][NP], fb[2][TN][NP];
        auto read_frags = [&](auto setc, int stage, int s) {
            constexpr int S = decltype(setc)::value;
            const char* sb = smem + stage * STAGE;
    #pragma unroll
            for (int i = 0; i < TM; ++i)
    #pragma unroll
                for (int p = 0; p < NP; ++p)
                    fa[S][i][p] = *reinterpret_cast<const u32x4*>(sb + p * A_PLANE + a_frag + i * 32 * RBYTES + foff[s]);
    #pragma unroll
            for (int j = 0; j < TN; ++j)
    #pragma unroll
                for (int p = 0; p < NP; ++p)
                    fb[S][j][p] = *reinterpret_cast<const u32x4*>(sb + p * B_PLANE + b_frag + j * 32 * RBYTES + foff[s]);
        };
        // do_read: after the first product group (TM*TN MFMAs) the fragments of the next tile's first k-step are read
        // into the other register set — BEHIND matrix instructions, so the LDS round trip is covered; in front of the
        // k-step's first MFMA the compiler's lgkmcnt wait for them idles the wave once per k-tile (the run-time branch
        // keeps the reads where they are written)
        auto mfmas = [&](auto setc, bool do_issue, char* nb, bool do_read = false, int rstage = 0) {
            constexpr int S = decltype(setc)::value;
    #pragma unroll
            for (int m = 0; m < NMFS; ++m) {
                const int t = m / (TM * TN);
                const int i = (m / TN) % TM, j = m % TN;
                acc[i][j] = mfma16<T>(fa[S][i][dprod_pa(NP, t)], fb[S][j][dprod_pb(NP, t)], acc[i][j]);
                if (m + 1 == TM * TN) {
                    if (do_read) read_frags(std::integral_constant<int, S ^ 1>{}, rstage, 0);
                }
                if ((m + 1) % GAP == 0 && (m + 1) / GAP <= LPT) {
                    if (do_issue) dma_one((m + 1) / GAP - 1, nb);
                }
            }
            if (do_issue) {
    #pragma unroll
                for (int d = NMFS / GAP; d < LPT; ++d) dma_one(d, nb);  // (fewer MFMAs than DMA instructions)
                step_state();
            }
        };
        const int ktiles = a.ktiles;
        auto wait_tile = [&](int t) {                                  // this wave's DMA of tile t has landed
            const int last_issued = t + ST - 2 < ktiles - 1 ? t + ST - 2 : ktiles - 1;   // tiles <= t-1+ST-1 were issued so far
            const int younger = last_issued - t;
            if (ST >= 4 && younger >= 2) wait_vm<(ST >= 4 ? 2 * LPT : 0)>();
            else if (ST >= 3 && younger >= 1) wait_vm<(ST >= 3 ? LPT : 0)>();
            else wait_vm<0>();
        };
        // Eight waves, one k-step per tile (NP = 3): the two waves of a SIMD run HALF A TILE OUT OF PHASE.  Every wave's tile
        // is [first half of the MFMAs, nothing else] [second half + the next tile's fragment reads + the DMA issue]; waves
        // 0-3 pass the tile barrier in front of the first half, waves 4-7 between the halves.  The barrier releases both at
        // once, so one wave of each SIMD is in its pure-MFMA half while its partner does everything else, and they swap.
        // (Same barrier count for every wave; tile kt's stage is overwritten only behind barrier kt+1, by which time both
        // groups have consumed their fragments of tile kt.)  Debug bit 256: everyone passes the barrier in front (A/B).
        constexpr bool PHASED = NW == 8 && KS == 1 && (NMFS % 2 == 0);
        const bool late_barrier = PHASED && wave >= NW / 2 && !(a.dbg & 256);
        auto mfmas_half = [&](auto setc, auto halfc, bool do_issue, char* nb, bool do_read, int rstage) {
            constexpr int S = decltype(setc)::value;
            constexpr int HALF = decltype(halfc)::value;
            constexpr int H = NMFS / 2;
            constexpr int GAP2 = H / (LPT + 1) > 0 ? H / (LPT + 1) : 1;
    #pragma unroll
            for (int mm = 0; mm < H; ++mm) {
                const int m = HALF * H + mm;
                const int t = m / (TM * TN);
                const int i = (m / TN) % TM, j = m % TN;
                acc[i][j] = mfma16<T>(fa[S][i][dprod_pa(NP, t)], fb[S][j][dprod_pb(NP, t)], acc[i][j]);
                if (HALF == 1) {
                    if (mm == 0) {
                        if (do_read) read_frags(std::integral_constant<int, S ^ 1>{}, rstage, 0);
                    }
                    if ((mm + 1) % GAP2 == 0 && (mm + 1) / GAP2 <= LPT) {
                        if (do_issue) dma_one((mm + 1) / GAP2 - 1, nb);
                    }
                }
            }
            if (HALF == 1) {
                if (do_issue) {
    #pragma unroll
                    for (int d = H / GAP2; d < LPT; ++d) dma_one(d, nb);
                    step_state();
                }
            }
        };
        // tile kt whose first k-step's fragments sit in register set PAR
        auto tile_step = [&](auto parc, int kt, int stage) {
            constexpr int PAR = decltype(parc)::value;
            const int next = stage + 1 == ST ? 0 : stage + 1;
            if constexpr (PHASED) {
                const bool more = kt + 1 < ktiles;
                if (more && !late_barrier) {
                    wait_tile(kt + 1);
                    __builtin_amdgcn_s_barrier();
                }
                mfmas_half(std::integral_constant<int, PAR>{}, std::integral_constant<int, 0>{}, false, nullptr, false, 0);
                if (more && late_barrier) {
                    wait_tile(kt + 1);
                    __builtin_amdgcn_s_barrier();
                }
                mfmas_half(std::integral_constant<int, PAR>{}, std::integral_constant<int, 1>{}, kt + ST < ktiles,
                           smem + stage * STAGE, more, next);
                return;
            }
    #pragma unroll
            for (int s = 0; s + 1 < KS; ++s) {
                if (((PAR + s) & 1) == 0) { read_frags(std::integral_constant<int, 1>{}, stage, s + 1); mfmas(std::integral_constant<int, 0>{}, false, nullptr); }
                else { read_frags(std::integral_constant<int, 0>{}, stage, s + 1); mfmas(std::integral_constant<int, 1>{}, false, nullptr); }
            }
            constexpr int LASTSET = (PAR + KS - 1) & 1;
            const bool more = kt + 1 < ktiles;
            if (more) {
                wait_tile(kt + 1);
                __builtin_amdgcn_s_barrier();                          // tile kt+1 is in LDS; every wave has read all of tile kt
            }
            mfmas(std::integral_constant<int, LASTSET>{}, kt + ST < ktiles, smem + stage * STAGE, more, next);
        };

        // ---- main loop ------------------------------------------------------------------------------------------------
    #pragma unroll
        for (int t = 0; t < ST; ++t)
            if (t < ktiles) issue(t);
        if constexpr (USE_SS) ss_publish();
        {
            const int younger = (ktiles < ST ? ktiles : ST) - 1;      // tiles issued after tile 0
            if (younger >= 3) wait_vm<(ST >= 4 ? 3 * LPT : 0)>();
            else if (younger == 2) wait_vm<(ST >= 3 ? 2 * LPT : 0)>();
            else if (younger == 1) wait_vm<LPT>();
            else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        read_frags(std::integral_constant<int, 0>{}, 0, 0);
        int stage = 0;
        if constexpr (KS % 2 == 0) {
            for (int kt = 0; kt < ktiles; ++kt) {
                tile_step(std::integral_constant<int, 0>{}, kt, stage);
                stage = stage + 1 == ST ? 0 : stage + 1;
            }
        } else {
            int kt = 0;
            for (; kt + 1 < ktiles; kt += 2) {
                tile_step(std::integral_constant<int, 0>{}, kt, stage);
                stage = stage + 1 == ST ? 0 : stage + 1;
                tile_step(std::integral_constant<int, 1>{}, kt + 1, stage);
                stage = stage + 1 == ST ? 0 : stage + 1;
            }
            if (kt < ktiles) tile_step(std::integral_constant<int, 0>{}, kt, stage);
        }

    }
    GV_PT(2);
    __syncthreads();                                               // every wave is done with the ring: reuse it for staging
    if constexpr (LATE_SS) ss_publish();                           // (the ring is free now)
    if constexpr (EPI == 0) {
        if constexpr (HAS_SUMS) {                                  // the sums table (in the freed ring where it fits)
            gvconv::stat_table_init<STATS>(a.st, smem, tid, NW * 64, BN, n0, a.cout, st_b0);
            __syncthreads();
        } else if constexpr (LATE_SS) {
            __syncthreads();
        }
        lp_epilogue_staged<T, TM, TN, STATS>(a, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem + wave * EpiGeom<TN>::BYTES), 32,
                                             sstab, BN, smem, st_b0 * (HAS_SUMS ? a.st.hw : 0));
        if constexpr (HAS_SUMS) {
            if (!(a.st.dbg & 16384)) __syncthreads();              // every lane's runs are in the table
            const int last = (m0 + BM < a.M ? m0 + BM : a.M) - 1;
            gvconv::stat_publish<STATS>(a.st, smem, tid, BN, n0, a.cout, st_b0, min(last / a.st.hw - st_b0 + 1, a.st.slots));
        }
    } else if constexpr (EPI == 1)
        gvconv::conv_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, lane);
    else {
        if constexpr (LATE_SS) __syncthreads();
        x3_epilogue_staged<TM, TN>(a, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem + wave * X3EpiGeom<TN>::BYTES),
                                   sstab, BN);
    }
#ifdef GV_PHASE_TIMES
    GV_PT(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GV_PT(4);
    if (a.phase_buf && lane == 0) {                                // [workgroup][wave][8]: t0..t4, HW_ID, XCC_ID
        unsigned long long* o = a.phase_buf + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = gv_pt[i];
        o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        {                                                          // per prologue tile: 16-bit clock counts
            unsigned long long prev = gv_pt[5], w = 0;
            for (int i = 0; i < 4; ++i) {
                const unsigned long long dt = gv_pi[i] > prev ? gv_pi[i] - prev : 0;
                w = (w << 16) | (dt > 65535 ? 65535 : dt);
                if (gv_pi[i]) prev = gv_pi[i];
            }
            o[6] = w;
        }
        o[7] = ((gv_pt[5] - gv_pt[0]) << 32) | ((gv_pt[6] - gv_pt[5]) & 0xffffffffull);   // setup | DMA issue (NP = 1)
    }
#endif
}

#ifndef GV_KERNEL_ONLY     // (tools/kernel_asm.sh compiles single instantiations of the kernel above: no launchers)
// one zero page per device for the padding taps of the gather (lazily allocated OUTSIDE any stream capture: every
// engine runs eagerly once before it captures)
const void* zero_page_for_current_device() {
    static std::mutex mu;
    static void* pages[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!pages[dev]) {
        void* p = nullptr;
        if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 4096) != hipSuccess) { (void)hipFree(p); return nullptr; }
        pages[dev] = p;
    }
    return pages[dev];
}

template <typename T, int NP, int WM, int WN, int TM, int TN, int ST, int EPI, int KTX = 0>
int launch_dma(const ConvArgs& a0, hipStream_t st) {
    using G = DmaGeom<NP, KTX>;
    if constexpr (KTX == 64) {                                     // whole 64-channel chunks inside one tap, no K padding
        if (a0.cin % 64 != 0 || a0.x_ld % 8 != 0) return GV_E_UNSUPPORTED;
    }
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    ConvArgs a = a0;
    a.Kpad = (a.K + G::KT - 1) / G::KT * G::KT;
    a.ktiles = a.Kpad / G::KT;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    const int64_t nwg = (int64_t)gv_ceil_div(a.M, BM) * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    a.zeros = zero_page_for_current_device();
    if (!a.zeros) return GV_E_UNSUPPORTED;
    a.korder = (a.kh * a.kw > 1 && a.kh * a.kw <= 32 && a.kw < 32 && a.cin % G::KT == 0 && a.dil_shift == 0 && !(a.dbg & 128)) ? 1 : 0;   // dbg 128: tap-major (A/B)
    const size_t ring = (size_t)ST * NP * (BM + BN) * G::RBYTES;
    const size_t epi = EPI == 1 ? 0 : (size_t)(WM * WN) * (EPI == 0 ? EpiGeom<TN>::BYTES : X3EpiGeom<TN>::BYTES);
    static_assert(dma_ss_off<NP, WM, WN, TM, TN, ST, EPI, KTX>() >= (int)((size_t)ST * NP * (BM + BN) * G::RBYTES), "table behind the ring");
    (void)ring; (void)epi;
    size_t lds = (size_t)dma_ss_off<NP, WM, WN, TM, TN, ST, EPI, KTX>() +
                 (dma_use_ss<NP, WM, WN, TM, TN, ST, EPI, KTX>() ? 4 * BN * sizeof(float) : 0);
    auto go = [&](auto mode) -> int {                              // (one instantiation, and one attribute cache, per kernel)
        auto kern = &conv_dma<T, NP, WM, WN, TM, TN, ST, EPI, decltype(mode)::value, KTX>;
        if (lds > 160 * 1024) return GV_E_UNSUPPORTED;
        if (lds > 64 * 1024) {
            const bool ok = GV_BIG_LDS_OK(kern, 160 * 1024);
            if (!ok) return GV_E_UNSUPPORTED;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WM * WN * 64), lds, st, a);
        GV_LAUNCH_CHECK();
        return GV_OK;
    };
    if (a.st.mode != gvconv::STAT_OFF) {                           // BatchNorm sums in the epilogue (conv_stats.h)
        if constexpr (EPI == 0 && NP == 1) {
            if (!gvconv::stat_tile_ok(a.st, BM, a.cout, TM * 32)) return GV_E_UNSUPPORTED;
            a.st.slots = gvconv::stat_rows(BM, a.st.hw, a.st.G);
            a.st.fold = gvconv::stat_slots(BM, a.st.hw) > a.st.G ? 1 : 0;
            // the tables live in the ring the epilogue has freed, behind the waves' staging blocks, where they fit in
            // front of the epilogue's constants; else behind everything (more LDS per workgroup)
            const size_t tab = gvconv::stat_lds_bytes(a.st.mode, a.st.slots, BN);
            constexpr size_t epi_b = (size_t)WM * WN * EpiGeom<TN>::BYTES, ss_off = dma_ss_off<NP, WM, WN, TM, TN, ST, EPI, KTX>();
            constexpr size_t late = dma_late_ss<NP, WM, WN, TM, TN, ST, EPI, KTX>() ? (size_t)16 * BN : 0;   // (the late constants table)
            if (epi_b + late + tab <= ss_off) {
                a.st.lds_off = (int)(epi_b + late);
            } else {
                a.st.lds_off = (int)((lds + 15) / 16 * 16);
                lds = (size_t)a.st.lds_off + tab;
            }
            if (a.st.mode == gvconv::STAT_FWD) return go(std::integral_constant<int, gvconv::STAT_FWD>{});
            return go(std::integral_constant<int, gvconv::STAT_BWD>{});
        } else {
            return GV_E_UNSUPPORTED;
        }
    }
    if constexpr (EPI == 0 && NP == 1) {                           // 16-bit output: the lean epilogue where the launch allows it
        if (gvconv::lp_epilogue_lean_ok(a)) return go(std::integral_constant<int, gvconv::STAT_LEAN>{});
    }
    return go(std::integral_constant<int, 0>{});
}

template <typename T>
int launch_dma_lp(int cfg, const ConvArgs& a, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_dma<T, 1, 2, 2, 2, 2, 4, 0>(a, st);      // 128 x 128, 4 waves, 4 stages (64 KB: two per CU)
        case 1: return launch_dma<T, 1, 4, 2, 2, 2, 3, 0>(a, st);      // 256 x 128, 8 waves
        case 2: return launch_dma<T, 1, 2, 4, 2, 2, 3, 0>(a, st);      // 128 x 256, 8 waves
        case 3: return launch_dma<T, 1, 2, 4, 4, 2, 3, 0>(a, st);      // 256 x 256, 8 waves
        case 4: return launch_dma<T, 1, 2, 2, 2, 3, 4, 0>(a, st);      // 128 x 192
        case 5: return launch_dma<T, 1, 4, 2, 2, 3, 3, 0>(a, st);      // 256 x 192, 8 waves
        case 6: return launch_dma<T, 1, 2, 2, 2, 1, 4, 0>(a, st);      // 128 x 64
        case 7: return launch_dma<T, 1, 2, 2, 1, 2, 4, 0>(a, st);      // 64 x 128
        case 8: return launch_dma<T, 1, 4, 1, 1, 3, 4, 0>(a, st);      // 128 x 96
        case 9: return launch_dma<T, 1, 2, 2, 4, 2, 3, 0>(a, st);      // 256 x 128, FOUR waves (128 x 64 each): two per CU
        case 10: return launch_dma<T, 1, 2, 2, 2, 4, 3, 0>(a, st);     // 128 x 256, four waves
        case 11: return launch_dma<T, 1, 2, 2, 3, 2, 3, 0>(a, st);     // 192 x 128, four waves
        // two-stage rings: a third (fourth) workgroup per CU instead of a third stage
        case 12: return launch_dma<T, 1, 2, 2, 3, 2, 2, 0>(a, st);     // 192 x 128, 2 stages (40 KB: three per CU)
        case 13: return launch_dma<T, 1, 2, 2, 2, 2, 2, 0>(a, st);     // 128 x 128, 2 stages (32 KB: four per CU)
        case 14: return launch_dma<T, 1, 2, 2, 2, 3, 2, 0>(a, st);     // 128 x 192, 2 stages (40 KB)
        case 15: return launch_dma<T, 1, 2, 2, 4, 2, 2, 0>(a, st);     // 256 x 128, four waves, 2 stages (48 KB: three per CU)
        // round 4: 64-deep k-tiles (128-byte LDS rows = whole cache lines of the source per row; DmaGeom): cin % 64 == 0
        case 16: return launch_dma<T, 1, 2, 2, 2, 3, 2, 0, 64>(a, st); // 128 x 192, 2 stages (80 KB: two per CU)
        case 17: return launch_dma<T, 1, 4, 2, 2, 3, 2, 0, 64>(a, st); // 256 x 192, 8 waves, 2 stages (112 KB)
        case 18: return launch_dma<T, 1, 2, 2, 2, 2, 2, 0, 64>(a, st); // 128 x 128, 2 stages (64 KB: two per CU)
        case 19: return launch_dma<T, 1, 2, 2, 2, 2, 3, 0, 64>(a, st); // 128 x 128, 3 stages (96 KB)
        case 20: return launch_dma<T, 1, 4, 2, 2, 2, 2, 0, 64>(a, st); // 256 x 128, 8 waves, 2 stages (96 KB)
        case 21: return launch_dma<T, 1, 2, 4, 4, 2, 2, 0, 64>(a, st); // 256 x 256, 8 waves, 2 stages (128 KB)
        // tall tiles for the 64- / 96-column layers of Mixed_5 (M = 240 000): the 128 x 96 tile reads 55 flop per staged byte
        case 22: return launch_dma<T, 1, 4, 1, 2, 3, 2, 0>(a, st);     // 256 x 96, 4 waves (64 x 96 each), 2 stages (44 KB: three per CU)
        case 23: return launch_dma<T, 1, 4, 1, 2, 2, 2, 0>(a, st);     // 256 x 64, 4 waves (64 x 64 each), 2 stages (40 KB)
        case 24: return launch_dma<T, 1, 4, 1, 2, 3, 2, 0, 64>(a, st); // 256 x 96, 64-deep k-tiles (88 KB)
    }
    return GV_E_UNSUPPORTED;
}

// fp32 values as three bf16 planes (P3 input); fp32 output straight from the accumulators, or P3 output staged
template <int EPI>
int launch_dma_x3_e(int cfg, const ConvArgs& a, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_dma<__bf16, 3, 2, 2, 2, 2, 3, EPI>(a, st);      // 128 x 128, 4 waves, 3 stages of 24 KB: two per CU
        case 1: return launch_dma<__bf16, 3, 4, 2, 2, 2, 3, EPI>(a, st);      // 256 x 128, 8 waves
        case 2: return launch_dma<__bf16, 3, 2, 4, 2, 2, 3, EPI>(a, st);      // 128 x 256, 8 waves
        case 3: return launch_dma<__bf16, 3, 2, 2, 2, 3, 3, EPI>(a, st);      // 128 x 192, 4 waves
        case 4: return launch_dma<__bf16, 3, 4, 1, 1, 3, 4, EPI>(a, st);      // 128 x 96
        case 5: return launch_dma<__bf16, 3, 2, 2, 2, 1, 4, EPI>(a, st);      // 128 x 64
        case 6: return launch_dma<__bf16, 3, 2, 2, 1, 2, 4, EPI>(a, st);      // 64 x 128
        case 7: return launch_dma<__bf16, 3, 4, 2, 2, 3, 2, EPI>(a, st);      // 256 x 192, 8 waves, 2 stages of 42 KB
        case 8: return launch_dma<__bf16, 3, 4, 2, 2, 3, 3, EPI>(a, st);      // 256 x 192, 8 waves, 3 stages (126 KB)
        case 9: return launch_dma<__bf16, 3, 4, 1, 2, 2, 3, EPI>(a, st);      // 256 x 64, 4 waves (64 x 64 each)
        case 10: return launch_dma<__bf16, 3, 4, 1, 2, 3, 3, EPI>(a, st);     // 256 x 96, 4 waves (64 x 96 each)
        case 11: return launch_dma<__bf16, 3, 2, 2, 2, 2, 4, EPI>(a, st);     // 128 x 128, 4 stages
        case 12: return launch_dma<__bf16, 3, 2, 2, 4, 2, 2, EPI>(a, st);     // 256 x 128, FOUR waves (128 x 64 each), 2 stages
        case 13: return launch_dma<__bf16, 3, 8, 1, 1, 5, 3, EPI>(a, st);     // 256 x 160, 8 waves (32 x 160 each): the 160-channel
                                                                              // 1x7 / 7x1 layers of Mixed_6c / 6d in ONE column tile
        case 14: return launch_dma<__bf16, 3, 4, 1, 1, 5, 4, EPI>(a, st);     // 128 x 160, 4 waves
        case 15: return launch_dma<__bf16, 3, 2, 2, 2, 3, 2, EPI>(a, st);     // 128 x 192, 4 waves, 2 stages (61 KB: two per CU)
        // two-stage forms of the narrow tiles: a second (third) workgroup per CU covers a workgroup's barriers and epilogue
        // better than a third stage covers its loads — Mixed_5's 64- / 96-column layers -9 ... -15 % (128 x 96, 128 x 64,
        // 256 x 128 and 256 x 160 on two stages were measured too and won nowhere)
        case 16: return launch_dma<__bf16, 3, 4, 1, 2, 3, 2, EPI>(a, st);     // 256 x 96, 4 waves, 2 stages (66 KB: two per CU)
        case 17: return launch_dma<__bf16, 3, 4, 1, 2, 2, 2, EPI>(a, st);     // 256 x 64, 4 waves, 2 stages (60 KB: two per CU)
        case 18: return launch_dma<__bf16, 3, 2, 2, 2, 2, 2, EPI>(a, st);     // 128 x 128, 2 stages (48 KB: three per CU)
    }
    return GV_E_UNSUPPORTED;
}

int launch_dma_x3(int cfg, const ConvArgs& a, hipStream_t st) {
    // fp32 destinations with 16-byte aligned rows also take the staged epilogue (16-byte stores); dbg bit 64 forces the
    // direct one (A/B)
    const bool vec = (a.y_ld % 4 == 0) && gv_aligned16(a.y) && a.cout % 8 == 0 && (a.res == nullptr || (a.res_ld % 4 == 0 && gv_aligned16(a.res)));
    if (a.split > 0) return launch_dma_x3_e<2>(cfg, a, st);      // fused siblings: the staged epilogue routes the columns
    return (a.y_p3 || (vec && !(a.dbg & 64))) ? launch_dma_x3_e<2>(cfg, a, st) : launch_dma_x3_e<1>(cfg, a, st);
}

}  // namespace

namespace gvconv {

const void* dma_zero_page() { return zero_page_for_current_device(); }   // (wgrad_dma.hip shares it)

constexpr int kDmaX3Tiles = 19;
int dma_x3_num_cfgs() { return kDmaX3Tiles + ws_x3_num_cfgs(); }   // + the wave-specialised kernel's tiles (conv_ws_x3.hip)

// P3 input: whole 16-channel groups inside one filter tap
bool dma_x3_ok(const ConvArgs& a) { return a.cin % 16 == 0 && a.x_ld % 16 == 0; }

int dma_x3_launch(int cfg, const ConvArgs& a, hipStream_t st) {
    return cfg >= kDmaX3Tiles ? ws_x3_launch(cfg - kDmaX3Tiles, a, st) : launch_dma_x3(cfg, a, st);
}

constexpr int kDmaLpTiles = 25;
int dma_lp_num_cfgs() { return kDmaLpTiles + ws_lp_num_cfgs(); }   // + the wave-specialised kernel's tiles (conv_ws.hip)

// the DMA loader's layer class: whole 8-channel chunks inside one filter tap, 16-byte aligned pixels, 16-bit input
bool dma_lp_ok(const ConvArgs& a, bool generic, bool xf32) {
    return !generic && !xf32 && a.cin % 8 == 0 && a.x_ld % 8 == 0;
}

int dma_lp_launch(int dtype, int cfg, const ConvArgs& a, hipStream_t st) {
    if (cfg >= kDmaLpTiles) return ws_lp_launch(dtype, cfg - kDmaLpTiles, a, st);
    if (dtype == GV_BF16) return launch_dma_lp<__bf16>(cfg, a, st);
    if (dtype == GV_F16) return launch_dma_lp<_Float16>(cfg, a, st);
    return GV_E_UNSUPPORTED;
}

}  // namespace gvconv

#else
}  // namespace
#endif
