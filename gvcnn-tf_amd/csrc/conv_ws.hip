// conv_ws.hip — wave-specialised 16-bit convolution: LOADER waves feed an LDS ring, CONSUMER waves do nothing but
// LDS fragment reads and MFMAs (round 5; the reference's conv+BN+ReLU sites: nets/inception_v3.py:137-338,
// nets/resnet_v2.py:73-95).
//
// Why another convolution kernel.  conv_dma.hip is an implicit GEMM: every k-tile DMAs a fresh [BM rows][32 channels]
// slice of the im2col matrix, so a 1x7 layer moves each input pixel SEVEN times from L2 to LDS (a 3x3 nine, a 5x5
// twenty-five times).  The L2 -> LDS path of a CU delivers about 70 GB/s (MI355X_MICROARCH.md, "Indexed rows: gather
// into LDS", rows served by the XCD's L2); a 256 x 192 tile at the full MFMA rate needs 89 GB/s of im2col rows, a
// 256 x 128 tile 114, a 256 x 64 tile 190 — the operand feed, not the matrix pipe, sets those kernels' ceiling, and
// every wave also stalls 100 - 200 clocks per DMA instruction it issues next to its own MFMAs.  Here
//   * STRIP mode (stride-1 "same-grid" convolutions: oh x ow = ih x iw): the tile's BM output pixels are BM consecutive
//     pixels of the flattened (image, y, x) grid, and tap (dy, dx) of pixel m is pixel m + dy*iw + dx — so ONE strip
//     of halo_lo + BM + halo_hi consecutive input pixels per 32-channel chunk serves all taps: the consumers read their
//     A fragments at a per-tap ROW SHIFT (rows outside the image are redirected to a zero row in LDS by a per-lane tap
//     mask), and only the filter slice is new per tap.  A 1x7 layer's A traffic falls 7x, a 3x3's 9x.
//   * GEMM mode (1x1): the strip is the tile itself and changes every k-step — a plain loader-fed GEMM ring.
//   * four loader waves (one per SIMD) do all address arithmetic and every global_load_lds; eight consumer waves (two
//     per SIMD, 64 x 32*TN outputs each) never touch vector memory inside the k-loop.  One raw s_barrier per k-step is
//     both the FULL and the FREE signal; the loaders run NB - 1 k-steps (and one strip) ahead behind counted vmcnt.
//
// LDS: [zero row][B ring: NB slots of BN rows x 64 B][A strips: NA buffers of strip_rows x 64 B]; rows are 64 bytes
// (32 channels), 16-byte chunks XOR-swizzled by (row >> 2) & 3 — on the SOURCE address by the loader (a DMA instruction
// writes 1 KiB lane-linearly) and on the read address by the consumer; a row shift keeps the 16 rows of every
// ds_read_b128 service group distinct modulo 16, so the shifted reads stay conflict-free.
// k order: channel chunk outer, filter tap inner (conv_dma.hip's chunk-major order): k-step (c, t) multiplies strip c
// at shift(t) with filter columns [t*cin + 32c, +32) of the packed [cout][Kpad] filter.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"
#include "conv_lp_epi.h"

namespace {

constexpr int WS_ZERO = 128;                   // bytes in front of the ring: the zero row
// KT channels per k-step: LDS rows of 64 bytes (KT = 32; 16 rows per DMA instruction, chunk swizzle (row >> 2) & 3) or of
// 128 bytes (KT = 64: whole cache lines of the source, 8 rows per instruction, swizzle (row >> 1) & 7, half the barriers
// and tap address updates per MFMA; needs cin % 64 == 0)
template <int KT> struct WsGeom {
    static constexpr int RB = KT * 2;          // bytes of one LDS row
    static constexpr int RPI = 1024 / RB;      // rows per DMA instruction
    static constexpr int CPR = RB / 16;        // 16-byte chunks per row
    static constexpr int KS = KT / 16;         // MFMA k-steps per k-step of the ring
    __host__ __device__ static constexpr int swz(int row) { return CPR == 4 ? ((row >> 2) & 3) : ((row >> 1) & 7); }
};

struct WsArgs {
    int taps, nchunks, nk;                     // kh*kw, cin/32, taps*nchunks
    int halo_lo;                               // strip rows in front of the tile's first pixel
    int strip_blocks, strip_bytes;             // 16-row blocks per strip buffer, bytes per buffer
    int na;                                    // strip buffers
    int strip_off;                             // bytes in front of a strip buffer's rows (DIET: its own zero row, 128)
    int tab_off;                               // DIET: LDS offset of the tap table [taps][BM] of 16-bit entries
    int pad_;
};

__device__ __forceinline__ void ws_dma16(const char* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void ws_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void ws_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// bytes of LDS the epilogue re-uses
template <int WM, int WN, int TN>
constexpr int ws_epi_bytes() { return WM * WN * 8192 + 16 * WN * TN * 32; }   // staging blocks (two 32 x 32 fp32 blocks per wave) + constants table

// The epilogue of an interior tile with one 16-bit destination (no residual, no second output, ReLU on every column or
// none): straight-line, and software-pipelined over the wave's 32 x 32 blocks through TWO staging blocks — the LDS writes of
// block b+1 are issued in front of the read-back of block b, so a block's write -> read round trip is covered by its
// predecessor's conversion and stores.  The general staged epilogue (conv_lp_epi.h) serialises write, wait, read, convert,
// store per block under a dozen run-time branches (residual, row bounds, second output, partial ReLU), and in this kernel —
// one workgroup per CU, nothing to overlap it with — it was 13 - 28 % of a workgroup's life
// (profiles/r5_ws_phase_times_*.txt, "epi"; profiles/r5_ws_ablation.txt, dbg 4).
template <typename T, int TM, int TN>
__device__ __forceinline__ void ws_epilogue_fast(const ConvArgs& a, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                 int lane, char* stage2, const float* sstab, int bn) {
    constexpr int NBLK = TM * TN;
    const int col_l = lane & 31, row_h = 4 * (lane >> 5);
    const int rrow = lane >> 2, rchunk = lane & 3;                 // read-back: 16 rows x 4 chunks of 8 columns per pass
    const int woff = (row_h * 32 + col_l) * 4;                     // + ((r & 3) + 8 * (r >> 2)) * 128 per accumulator register
    const int roff = (rrow * 32 + rchunk * 8) * 4;                 // + 2048 for the second pass
    unsigned short* const y = reinterpret_cast<unsigned short*>(a.y);
    const bool relu = a.relu != 0;
    const size_t row0 = (size_t)(m0 + wm * TM * 32 + rrow) * (size_t)a.y_ld;
    auto put = [&](int b) {
        const int i = b % TM, j = b / TM;
        char* dst = stage2 + (b & 1) * 4096 + woff;
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(dst + ((r & 3) + 8 * (r >> 2)) * 128) = acc[i][j][r];
    };
    put(0);
    float sc[8], sh[8];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int i = b % TM, j = b / TM;
        if (b + 1 < NBLK) put(b + 1);
        __builtin_amdgcn_wave_barrier();
        const int col = n0 + (wn * TN + j) * 32 + rchunk * 8;
        if (i == 0) {                                              // a column block's constants: once per TM row blocks
            const float* t = sstab + (wn * TN + j) * 32 + rchunk * 8;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(t + 4 * hh);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(t + bn + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[4 * hh + e] = v0[e]; sh[4 * hh + e] = v1[e]; }
            }
        }
        const bool live = col + 8 <= a.cout;                       // (cout % 8 == 0: a chunk is whole or absent)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const char* src = stage2 + (b & 1) * 4096 + roff + pass * 2048;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 16);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (live) store_chunk_lean<T>(y + row0 + (size_t)((i * 32 + pass * 16) * a.y_ld) + col, v, relu);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// NL loader waves.  Eight consumers + four loaders = one 768-thread workgroup per CU; four consumers + two loaders = a
// 384-thread workgroup, two per CU (three waves per SIMD either way: 168 registers) — the second workgroup's k-loop
// covers the first one's epilogue, at twice the filter traffic per flop.
// DIET (round 6, strip mode): the consumers' per-k-step address arithmetic — strip row of the tap, swizzle, out-of-image
// select: ~20 vector instructions per k-step and row block — is replaced by a per-tile TABLE in LDS: entry [tap][row] is the
// fragment's byte offset / 16 inside a strip buffer, or 0 = the buffer's own zero row; a k-step then needs one ds_read_u16
// and two vector instructions per row block.  Loop body: ~15 vector + ~18 scalar instructions instead of ~34 + ~25 (12 MFMAs,
// 10 ds_read_b128 either way).  Selected by debug bit 2097152 / GV_WS_DIET=1 — see launch_ws for the measurement.
template <typename T, int WM, int WN, int TM, int TN, int NB, int STATS, bool GEMM, int KT = 32, int NL = 4, bool DIET = false>
__global__ __launch_bounds__((WM * WN + NL) * 64, (WM * WN + NL) * 3 / 12) void conv_ws(const ConvArgs a, const WsArgs w) {
    constexpr int WS_NLW = NL;
    using G = WsGeom<KT>;
    constexpr int WS_RB = G::RB, RPI = G::RPI, CPR = G::CPR, KS = G::KS;
    constexpr int NC = WM * WN;                                    // consumer waves
    constexpr int NT = (NC + WS_NLW) * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int B_SLOT = BN * WS_RB;
    constexpr int OFF_B = WS_ZERO, OFF_S = OFF_B + NB * B_SLOT;
    constexpr int UB = BN / RPI, LB = (UB + WS_NLW - 1) / WS_NLW;  // filter row blocks: DMA instructions per loader and k-step
    constexpr int LA = BM / RPI / WS_NLW;                          // GEMM mode: strip instructions per loader and k-step
    constexpr int PER = GEMM ? LA + LB : LB;                       // loads per loader wave and k-step the vmcnt counts rely on
    static_assert(!GEMM || BM % (RPI * WS_NLW) == 0, "whole strip blocks per loader");
    static_assert(KS % 2 == 0, "the fragment register sets alternate per MFMA k-step");
    static_assert(NB >= 3 && (NB - 1) * PER < 64, "ring depth / vmcnt range");
    static_assert(STATS == 0 || STATS == gvconv::STAT_LEAN, "BatchNorm sums: conv_dma.hip");

    extern __shared__ __attribute__((aligned(128))) char smem[];
#ifdef GV_PHASE_TIMES
    unsigned long long gv_pt[5] = {0, 0, 0, 0, 0}, gv_wait = 0;
#define WS_PT(i) gv_pt[i] = __builtin_amdgcn_s_memtime()
#define WS_WAIT_BEGIN() const unsigned long long gv_w0 = __builtin_amdgcn_s_memtime()
#define WS_WAIT_END() gv_wait += __builtin_amdgcn_s_memtime() - gv_w0
#else
#define WS_PT(i)
#define WS_WAIT_BEGIN()
#define WS_WAIT_END()
#endif
    WS_PT(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = gv_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % a.tiles_n, tile_m = lid / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = w.nk, NTAP = w.taps;

    // the epilogue's per-column constants: requested now, published into LDS behind the staging blocks once the ring is free
    float ss_v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool ss_dual = a.y2 != nullptr && a.split == 0;
    if (tid < BN) {
        const int cc = min(n0 + tid, a.cout - 1);
        ss_v[0] = a.scale[cc];
        ss_v[1] = a.shift[cc];
        if (ss_dual) { ss_v[2] = a.scale2[cc]; ss_v[3] = a.shift2[cc]; }
    }
    f32x16 acc[TM][TN];

    if (wave >= NC) {
        // =================================================== loader ===================================================
        const int lw = wave - NC;
        if (!(a.dbg & 16384)) {                                    // (debug bit 16384: consumers alone, no loads, no barriers — timing only)
        // A loader shares its SIMD with two consumers whose MFMA / LDS / VALU streams never pause, and issue arbitration
        // goes by priority, then age — the loaders are the workgroup's youngest waves.  Their few instructions per k-step
        // go first (debug bit 4096: no priority, A/B).
        if (!(a.dbg & 4096)) __builtin_amdgcn_s_setprio(3);
        const int lrow = lane / CPR;                               // row inside a row block
        // logical 16-byte chunk this lane fetches (swizzle at the source) in row block `blk`: with 128-byte rows the swizzle
        // reaches the block's parity
        auto lchunk = [&](int blk) -> int { return (lane % CPR) ^ G::swz(blk * RPI + lrow); };
        const char* xb = reinterpret_cast<const char*>(a.x);
        const unsigned pix_bytes = (unsigned)a.x_ld * 2u;
        const char* b_ptr[LB];
        int b_dst[LB];
        {
            const size_t wrow = (size_t)a.Kpad * 2;
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                int rb = lw + i * WS_NLW;
                rb = rb < UB ? rb : UB - 1;                        // surplus slots re-load the last block (same bytes)
                int n = n0 + rb * RPI + lrow;
                n = n < a.cout ? n : a.cout - 1;                   // columns past cout are never stored
                b_ptr[i] = reinterpret_cast<const char*>(a.w) + (size_t)n * wrow + lchunk(rb) * 16;
                b_dst[i] = OFF_B + rb * 1024;
            }
        }
        {
            int bq_t = 0, bq_c = 0, bq_slot = 0;                       // next filter slice to issue: tap, chunk, ring slot
            int sq_c = 0, sq_buf = 0;                                  // next strip to issue: chunk, buffer
            // timing ablations (results garbage): 32768 barriers but no loads; 65536 loads but no barriers anywhere
            const bool nodma = (a.dbg & 32768) != 0, nobar = (a.dbg & 65536) != 0;
            auto issue_b = [&]() {
                const unsigned koff = (unsigned)(bq_t * a.cin + bq_c * KT) * 2u;
                char* sb = smem + bq_slot * B_SLOT;
    #pragma unroll
                for (int i = 0; i < LB; ++i)
                    if (!nodma) ws_dma16(b_ptr[i] + koff, sb + b_dst[i]);
                if (++bq_t == NTAP) { bq_t = 0; ++bq_c; }
                bq_slot = bq_slot + 1 == NB ? 0 : bq_slot + 1;
            };
            auto issue_strip = [&]() {
                char* sb = smem + OFF_S + sq_buf * w.strip_bytes;
                // (a loader's blocks lw, lw + 4, ... share their parity: one chunk per lane)
                const unsigned coff = (unsigned)sq_c * (unsigned)WS_RB + (unsigned)lchunk(lw) * 16u;
                if constexpr (GEMM) {
    #pragma unroll
                    for (int i = 0; i < LA; ++i) {
                        const int blk = lw + i * WS_NLW;
                        int p = m0 + blk * RPI + lrow;
                        p = p < a.M ? p : a.M - 1;
                        if (!nodma) ws_dma16(xb + (size_t)(unsigned)p * pix_bytes + coff, sb + blk * 1024);
                    }
                } else {
                    for (int blk = lw; blk < w.strip_blocks; blk += WS_NLW) {
                        int p = m0 - w.halo_lo + blk * RPI + lrow;     // rows outside [0, M) are only ever read by masked taps
                        p = p < 0 ? 0 : (p < a.M ? p : a.M - 1);
                        if (!nodma) ws_dma16(xb + (size_t)(unsigned)p * pix_bytes + coff, sb + w.strip_off + blk * 1024);
                    }
                }
                ++sq_c;
                sq_buf = sq_buf + 1 == w.na ? 0 : sq_buf + 1;
            };
            // STRIP mode, behind the prologue: a strip is issued PIECEWISE.  Its buffer is free when the chunk two behind it
            // ends and it is needed a whole chunk later; issued at once, its 6 - 10 DMA instructions (~210 clocks each next
            // to busy consumers: profiles/r5_ws_ablation.txt) make the loader miss the next barrier once per chunk — the
            // consumers then spent 17 - 24 % of the k-loop waiting although the loaders idle half of it on average.  `ppi`
            // blocks per k-step finish it within taps - NB + 2 k-steps: NB - 2 filter slices are then issued behind its last
            // piece, which is what the counted waits need to cover it.
            int sp_blk = 1 << 30, sp_c = 0, sp_buf = 0;            // pending strip: next block (none: past the end), chunk, buffer
            const int ppi = ((w.strip_blocks + WS_NLW - 1) / WS_NLW + max(NTAP - NB + 2, 1) - 1) / max(NTAP - NB + 2, 1);
            auto strip_begin = [&]() {
                sp_blk = lw; sp_c = sq_c; sp_buf = sq_buf;
                ++sq_c;
                sq_buf = sq_buf + 1 == w.na ? 0 : sq_buf + 1;
            };
            auto strip_pieces = [&]() {
                char* sb = smem + OFF_S + sp_buf * w.strip_bytes;
                const unsigned coff = (unsigned)sp_c * (unsigned)WS_RB + (unsigned)lchunk(lw) * 16u;
                for (int i = 0; i < ppi && sp_blk < w.strip_blocks; ++i, sp_blk += WS_NLW) {
                    int p = m0 - w.halo_lo + sp_blk * RPI + lrow;
                    p = p < 0 ? 0 : (p < a.M ? p : a.M - 1);
                    if (!nodma) ws_dma16(xb + (size_t)(unsigned)p * pix_bytes + coff, sb + w.strip_off + sp_blk * 1024);
                }
            };
            // prologue: strip 0 and filter slice 0 first (the consumers' first fragments), then the rest of both rings
            issue_strip();
            issue_b();
            if constexpr (GEMM) {
    #pragma unroll
                for (int q = 1; q < NB; ++q) { issue_strip(); issue_b(); }        // (nk >= NB: the launcher checks)
            } else {
    #pragma unroll
                for (int q = 1; q < NB; ++q) issue_b();
            }
            WS_PT(1);
            ws_wait_vm<(NB - 2) * PER>();                              // everything up to filter slice 1 has landed
            if (!nobar) __builtin_amdgcn_s_barrier();
            // the second strip only now: issued in front of that barrier's wait it would be waited for (vmcnt retires in
            // order and the wait leaves only the youngest few instructions in flight) — a whole strip's issue and landing
            // added to every workgroup's start; it is needed a chunk later, and taps >= NB - 1 filter slices behind it
            // cover it in the counted waits below
            if constexpr (!GEMM) { if (w.nchunks > 1) { strip_begin(); strip_pieces(); } }   // (its first pieces now: the window below counts from here)
            int ft = 0;                                                // tap of k-step j
            for (int j = 0; j + 1 < nk; ++j) {
                // k-step j+2 must be in LDS before barrier j (the consumers read its first fragments before barrier j+1): the
                // NB-3 k-steps issued after it may stay in flight (a strip issued in between only makes the wait stricter; in
                // the drain wait for everything)
                WS_WAIT_BEGIN();
                if (j + NB - 1 <= nk - 1) ws_wait_vm<(NB - 3) * PER>();
                else ws_wait_vm<0>();
                if (!nobar) __builtin_amdgcn_s_barrier();              // ... and every consumer is done with k-step j's LDS
                WS_WAIT_END();
                const bool last_tap = ft + 1 == NTAP;
                if constexpr (GEMM) {
                    if (j + NB < nk) { issue_strip(); issue_b(); }
                } else {
                    if (last_tap && sq_c < w.nchunks) strip_begin();   // the strip this chunk occupied is free: chunk + NA
                    strip_pieces();
                    if (j + NB < nk) issue_b();
                }
                ft = last_tap ? 0 : ft + 1;
            }
        }
        }
        WS_PT(2);
    } else {
        // ================================================== consumer ==================================================
        const int wm = wave / WN, wn = wave % WN;
        const int r = lane & 31, h = lane >> 5;
        if (tid < 32) reinterpret_cast<unsigned*>(smem)[tid] = 0u; // the zero row: bytes [0, 128)
        if constexpr (DIET) {                                      // ... and one in front of each of the two strip buffers
            if (tid < 64) reinterpret_cast<unsigned*>(smem + OFF_S + (tid >> 5) * w.strip_bytes)[tid & 31] = 0u;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        // this lane's rows: strip row of the un-shifted tap and one bit per tap "inside the image"
        int rbase[TM];
        unsigned tapmask[TM];
        {
            const int ohow = a.oh * a.ow;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int lr = (wm * TM + i) * 32 + r;
                rbase[i] = lr + w.halo_lo;
                int m = m0 + lr;
                m = m < a.M ? m : a.M - 1;
                const int n = gv_div(m, a.y_div_img);
                const int rem = m - n * ohow;
                const int y = gv_div(rem, a.y_div_row);
                const int x = rem - y * a.ow;
                const int r_lo = max(0, a.pad_t - y), r_hi = min(a.kh, a.ih + a.pad_t - y);
                const int c_lo = max(0, a.pad_l - x), c_hi = min(a.kw, a.iw + a.pad_l - x);
                const unsigned rowbits = c_hi > c_lo ? (1u << c_hi) - (1u << c_lo) : 0u;
                unsigned tm = 0u;
                for (int fr = 0; fr < a.kh; ++fr)
                    if (fr >= r_lo && fr < r_hi) tm |= rowbits << (fr * a.kw);
                tapmask[i] = tm;
            }
        }
        int b0[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int brow = (wn * TN + j) * 32 + r;
            b0[j] = OFF_B + brow * WS_RB + ((G::swz(brow) ^ h) << 4);
        }
        // DIET: the tap table.  The WN waves that own the same rows share the taps between them; an entry is the same for
        // both k halves (h is folded in by the reader)
        int tab0 = 0;                                              // this lane's first row's column of the table
        if constexpr (DIET) {
            tab0 = w.tab_off + ((wm * TM) * 32 + r) * 2;
            int fr = 0, fs = wn;                                   // tap wn, wn + WN, ...: (fr, fs) kept by increments
            while (fs >= a.kw) { fs -= a.kw; ++fr; }
            for (int t = wn; t < NTAP; t += WN) {
                const int off = (fr - a.pad_t) * a.iw + (fs - a.pad_l);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = rbase[i] + off;
                    const int rel = (128 + row * WS_RB + (G::swz(row) << 4)) >> 4;
                    const int e = ((tapmask[i] >> t) & 1u) ? rel : 0;
                    if (h == 0) *reinterpret_cast<unsigned short*>(smem + tab0 + t * (BM * 2) + i * 64) = (unsigned short)e;
                }
                fs += WN;
                while (fs >= a.kw) { fs -= a.kw; ++fr; }
            }
        }
        // A fragment address (k16-step 0; step 1 is this ^ 32) of row block i for the tap at row shift `off` / mask bit
        // `bit`, strip buffer at byte `sbase`; a tap outside the image reads the zero row
        // (bit arithmetic, not a select: hipcc turns the select into exec-masked branches that cut the k-step's basic block)
        auto a_addr = [&](int i, int off, int tap, int sbase) -> int {
            const int row = rbase[i] + off;
            const int ad = sbase + row * WS_RB + ((G::swz(row) ^ h) << 4);
            const int in = -(int)((tapmask[i] >> tap) & 1u);       // all ones: the tap lies inside the image
            return (ad & in) | ((h << 4) & ~in);
        };
        int q_fr = 0, q_fs = 0, q_tap = 0;                         // tap of the k-step whose addresses are in aa[]
        int sbase = OFF_S, sbuf = 0, bslot = 0;                    // its strip buffer / filter slot (bytes)
        int aa[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (DIET) {                                  // (tap 0, straight from the arithmetic: the table is not published yet)
                const int row = rbase[i] - (a.pad_t * a.iw + a.pad_l);
                const int ad = sbase + 128 + row * WS_RB + ((G::swz(row) ^ h) << 4);
                aa[i] = (tapmask[i] & 1u) ? ad : sbase + (h << 4);
            } else {
                aa[i] = a_addr(i, -(a.pad_t * a.iw + a.pad_l), 0, sbase);
            }
        }
        u32x4 fa[2][TM], fb[2][TN];
        auto lds16 = [&](int addr) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + addr); };
        // one fragment read (q-th of the TM + TN of a k16-step) into register set S
        auto read_one = [&](auto setc, int q, int sx, const int (&av)[TM], int bs) {
            constexpr int S = decltype(setc)::value;
            if (q < TM) fa[S][q] = lds16(av[q] ^ sx);
            else fb[S][q - TM] = lds16(bs + (b0[q - TM] ^ sx));
        };
        // the MFMAs of one k16-step on set S; behind them, one per MFMA, the reads of the next k16-step into set S^1
        // BAR: the k-step's barrier sits behind MFMA number BARPOS of this half — by then every fragment of register
        // set S has been an MFMA operand, i.e. every LDS read of the k-step that ends here has returned (no lgkmcnt drain:
        // the reads of the NEXT k-step, issued around the barrier, stay in flight), and the wave has matrix work queued
        // while it waits.
        constexpr int BARPOS = TN < TM * TN - 1 ? TN : TM * TN - 1;    // MFMA index (fa[1] and every fb have been used)
        auto half = [&](auto setc, auto readc, int sx, const int (&av)[TM], int bs, auto barc, bool do_bar = true) {
            constexpr int S = decltype(setc)::value;
            constexpr bool READ = decltype(readc)::value;
            constexpr bool BAR = decltype(barc)::value;
#pragma unroll
            for (int m = 0; m < TM * TN; ++m) {
                const int i = m / TN, j = m % TN;
                acc[i][j] = mfma16<T>(fa[S][i], fb[S][j], acc[i][j]);
                if constexpr (READ) {
                    if (m < TM + TN) read_one(std::integral_constant<int, S ^ 1>{}, m, sx, av, bs);
                }
                if constexpr (BAR) {
                    if (m == BARPOS && do_bar) {
                        WS_WAIT_BEGIN();
                        __builtin_amdgcn_s_barrier();              // k-step j+2 is in LDS; k-step j's LDS may be overwritten
                        WS_WAIT_END();
                    }
                }
            }
            if constexpr (READ) {
#pragma unroll
                for (int q = TM * TN; q < TM + TN; ++q) read_one(std::integral_constant<int, S ^ 1>{}, q, sx, av, bs);
            }
        };
        using TT = std::true_type;
        using FF = std::false_type;
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        const bool nosync = (a.dbg & (16384 | 65536)) != 0;
        ws_wait_lds();
        WS_PT(1);
        if (!nosync) __builtin_amdgcn_s_barrier();                 // strip 0, filter slices 0 and 1 and the zero row are in LDS
        WS_PT(2);
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) read_one(S0{}, q, 0, aa, 0);
        // Stagger (debug bit 8192: off, A/B).  The two consumers of a SIMD are waves w and w + 4; run in lockstep they reach
        // their MFMA bursts, their LDS reads and the barrier together and idle together.  Waves 4-7 therefore take the
        // k-step's barrier HALF A K-STEP LATER in their own instruction stream (in the middle of the following k-step's
        // first half instead of this one's last): every SIMD then has one wave in front of the barrier and one behind it.
        // Legal: a wave's reads of k-step j have returned long before either point, and k-step j+2 — published by barrier j
        // — is first read behind both.
        const bool late = wave >= NC / 2 && (a.dbg & 8192) != 0;      // (debug bit 8192: the stagger, A/B — measured neutral to negative)
        const bool early = !late && !nosync, lateb = late && !nosync;
        for (int j = 0; j + 1 < nk; ++j) {
            // the next k-step's tap, strip buffer and slot (selects, no branches: a branch here would cut the basic block
            // and keep this arithmetic from being scheduled between the MFMAs below)
            int an[TM];
            if constexpr (DIET) {
                const bool wrap = q_tap + 1 == NTAP;
                q_tap = wrap ? 0 : q_tap + 1;
                const int sb1 = sbuf + 1 == w.na ? 0 : sbuf + 1;
                sbuf = wrap ? sb1 : sbuf;
                sbase = OFF_S + sbuf * w.strip_bytes;
                const int tp = tab0 + q_tap * (BM * 2);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int e = *reinterpret_cast<const unsigned short*>(smem + tp + i * 64);
                    an[i] = ((e ^ h) << 4) + sbase;
                }
            } else {
                const bool wrap = q_tap + 1 == NTAP;
                const bool row_end = q_fs + 1 == a.kw;
                q_tap = wrap ? 0 : q_tap + 1;
                q_fr = wrap ? 0 : (row_end ? q_fr + 1 : q_fr);
                q_fs = (wrap || row_end) ? 0 : q_fs + 1;
                const int sb1 = sbuf + 1 == w.na ? 0 : sbuf + 1;
                sbuf = wrap ? sb1 : sbuf;
                sbase = OFF_S + sbuf * w.strip_bytes;
                const int off = (q_fr - a.pad_t) * a.iw + (q_fs - a.pad_l);
#pragma unroll
                for (int i = 0; i < TM; ++i) an[i] = a_addr(i, off, q_tap, sbase);
            }
            const int bnext = bslot + B_SLOT == NB * B_SLOT ? 0 : bslot + B_SLOT;
            // the k-step's first KS - 1 MFMA k-steps; behind each one's MFMAs the next one's fragments (same strip rows and
            // filter slot, the chunk index advanced by XOR); the late waves' barrier j-1 sits in number KS/2 - 1
#pragma unroll
            for (int q = 0; q + 1 < KS; ++q) {
                if (q == KS / 2 - 1) {
                    if (q % 2 == 0) half(S0{}, TT{}, (q + 1) << 5, aa, bslot, TT{}, lateb && j > 0);
                    else half(S1{}, TT{}, (q + 1) << 5, aa, bslot, TT{}, lateb && j > 0);
                } else {
                    if (q % 2 == 0) half(S0{}, TT{}, (q + 1) << 5, aa, bslot, FF{});
                    else half(S1{}, TT{}, (q + 1) << 5, aa, bslot, FF{});
                }
            }
            // the last one: the next k-step's first fragments (in LDS since the PREVIOUS barrier) and — early waves — barrier j
            half(S1{}, TT{}, 0, an, bnext, TT{}, early);
#pragma unroll
            for (int i = 0; i < TM; ++i) aa[i] = an[i];
            bslot = bnext;
        }
#pragma unroll
        for (int q = 0; q + 1 < KS; ++q) {
            if (q == KS / 2 - 1) {
                if (q % 2 == 0) half(S0{}, TT{}, (q + 1) << 5, aa, bslot, TT{}, lateb && nk > 1);
                else half(S1{}, TT{}, (q + 1) << 5, aa, bslot, TT{}, lateb && nk > 1);
            } else {
                if (q % 2 == 0) half(S0{}, TT{}, (q + 1) << 5, aa, bslot, FF{});
                else half(S1{}, TT{}, (q + 1) << 5, aa, bslot, FF{});
            }
        }
        half(S1{}, FF{}, 0, aa, bslot, FF{});
        WS_PT(3);
    }
    // ====================================================== epilogue ======================================================
    __syncthreads();                                               // every wave is done with the ring: the staging blocks alias it
    constexpr int SS_OFF = NC * 8192;                              // (two staging blocks per wave, whatever the epilogue)
    float* sstab = !(a.dbg & 512) ? reinterpret_cast<float*>(smem + SS_OFF) : nullptr;   // dbg 512: constants from global (A/B)
    if (sstab != nullptr && tid < BN) {
        sstab[tid] = ss_v[0];
        sstab[BN + tid] = ss_v[1];
        if (ss_dual) { sstab[2 * BN + tid] = ss_v[2]; sstab[3 * BN + tid] = ss_v[3]; }
    }
    __syncthreads();
    if (wave < NC) {
        const int wm = wave / WN, wn = wave % WN;
        // interior tile, one plain destination: the pipelined straight-line epilogue (debug bit 1048576: the general one, A/B)
        const bool fast = STATS == gvconv::STAT_LEAN && sstab != nullptr && a.res == nullptr && a.y2 == nullptr && a.split == 0 &&
                          m0 + BM <= a.M && (!a.relu || a.relu_limit >= a.cout) && !(a.dbg & (4 | 1048576));
        if (fast)
            ws_epilogue_fast<T, TM, TN>(a, acc, m0, n0, wm, wn, lane, smem + wave * 8192, sstab, BN);
        else
            lp_epilogue_staged<T, TM, TN, STATS>(a, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem + wave * 8192), 32,
                                                 sstab, BN, smem, 0);
    }
#ifdef GV_PHASE_TIMES
    WS_PT(4);
    if (a.phase_buf && lane == 0) {                                // [workgroup][wave][8]: t0..t4, wait clocks, HW_ID | XCC_ID, role
        unsigned long long* o = a.phase_buf + ((size_t)blockIdx.x * 16 + wave) * 8;   // (16 slots per workgroup, whatever its size)
        for (int i = 0; i < 5; ++i) o[i] = gv_pt[i];
        o[5] = gv_wait;
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        o[7] = wave >= NC ? 1ull : 0ull;
    }
#endif
    (void)NT;
}

#ifndef GV_KERNEL_ONLY
// STRIP mode: stride 1, output grid = input grid, <= 32 taps; GEMM mode: 1x1.  Whole 32-channel chunks, 16-byte pixels.
bool ws_shape_ok(const ConvArgs& a, int kt) {
    return a.stride == 1 && a.dil_shift == 0 && a.oh == a.ih && a.ow == a.iw && a.cin % kt == 0 && a.x_ld % 8 == 0 &&
           a.kh * a.kw <= 32 && a.kw < 32 && a.pad_t < a.kh && a.pad_l < a.kw && a.pool == 0 && a.xscale == nullptr &&
           a.y_step == 0 && a.st.mode == gvconv::STAT_OFF && a.K % 32 == 0;
}

template <typename T, int WM, int WN, int TM, int TN, int NB, int KT = 32, int NL = 4>
int launch_ws(const ConvArgs& a0, hipStream_t st) {
    using G = WsGeom<KT>;
    constexpr int WS_RB = G::RB, RPI = G::RPI;
    constexpr int NC = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    if (!ws_shape_ok(a0, KT)) return GV_E_UNSUPPORTED;
    ConvArgs a = a0;
    a.Kpad = a.K;                                                  // (K % 32 == 0: the packed filter has no padding)
    WsArgs w;
    w.taps = a.kh * a.kw;
    w.nchunks = a.cin / KT;
    w.nk = w.taps * w.nchunks;
    const bool gemm = w.taps == 1;
    if (w.nk < NB || (!gemm && w.taps < NB - 1)) return GV_E_UNSUPPORTED;   // (the vmcnt counts of the loader assume it)
    w.halo_lo = a.pad_t * a.iw + a.pad_l;
    const int halo_hi = (a.kh - 1 - a.pad_t) * a.iw + (a.kw - 1 - a.pad_l);
    w.strip_blocks = gemm ? BM / RPI : gv_ceil_div(w.halo_lo + BM + halo_hi, RPI);
    w.na = gemm ? NB : 2;
    w.pad_ = 0;
    // strip mode: the tap-table form (DIET), when asked for, where its table and the strip buffers' zero rows still fit the
    // workgroup's LDS share
    const size_t lds_cap = (size_t)(NL == 2 ? 80 : 160) * 1024;
    // Measured NEUTRAL (profiles/r6_ws_diet_ab.txt: whole plans c3 0.249 -> 0.247, c5 0.283 -> 0.282; single layers +4 ... +6 %
    // as warm repeats): the consumers' instruction count is not what bounds this kernel.  Kept behind debug bit 2097152 /
    // GV_WS_DIET=1 with its parity tests; the arithmetic form stays the product.
    static const bool env_on = getenv("GV_WS_DIET") != nullptr;
    bool diet = !gemm && ((a.dbg & 2097152) || env_on);
    if (diet && (size_t)WS_ZERO + (size_t)NB * BN * WS_RB + 2 * ((size_t)128 + w.strip_blocks * 1024) +
                    ((size_t)w.taps * BM * 2 + 127) / 128 * 128 > lds_cap)
        diet = false;
    w.strip_off = diet ? 128 : 0;
    w.strip_bytes = w.strip_off + w.strip_blocks * 1024;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    const int64_t nwg = (int64_t)gv_ceil_div(a.M, BM) * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    size_t ring = (size_t)WS_ZERO + (size_t)NB * BN * WS_RB + (size_t)w.na * w.strip_bytes;
    w.tab_off = (int)ring;
    if (diet) ring += ((size_t)w.taps * BM * 2 + 127) / 128 * 128;   // (16-bit entries: offsets / 16 — any strip fits)
    const size_t epi = (size_t)ws_epi_bytes<WM, WN, TN>();
    const size_t lds = ring > epi ? ring : epi;
    if (lds > (NL == 2 ? 80 : 160) * 1024) return GV_E_UNSUPPORTED;   // (two of the small workgroups per CU)
    auto go = [&](auto mode, auto gm) -> int {
        constexpr bool GM = decltype(gm)::value;
        if (!GM && diet) {
            auto kern = &conv_ws<T, WM, WN, TM, TN, NB, decltype(mode)::value, false, KT, NL, true>;
            if (lds > 64 * 1024) {
                const bool ok = GV_BIG_LDS_OK(kern, 160 * 1024);
                if (!ok) return GV_E_UNSUPPORTED;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3((NC + NL) * 64), lds, st, a, w);
            GV_LAUNCH_CHECK();
            return GV_OK;
        }
        auto kern = &conv_ws<T, WM, WN, TM, TN, NB, decltype(mode)::value, decltype(gm)::value, KT, NL>;
        if (lds > 64 * 1024) {
            const bool ok = GV_BIG_LDS_OK(kern, 160 * 1024);
            if (!ok) return GV_E_UNSUPPORTED;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3((NC + NL) * 64), lds, st, a, w);
        GV_LAUNCH_CHECK();
        return GV_OK;
    };
    const bool lean = gvconv::lp_epilogue_lean_ok(a);
    using L = std::integral_constant<int, gvconv::STAT_LEAN>;
    using F = std::integral_constant<int, 0>;
    // 1x1: the vmcnt counts cover A tile + filter slice per k-step
    constexpr bool GEMM_OK = (NB - 1) * (BM / RPI / NL + (BN / RPI + NL - 1) / NL) < 64;
    // sixteen-wave workgroups live on 128 registers: only the lean epilogue fits
    if (NC + NL > 12 && !lean) return GV_E_UNSUPPORTED;
    if (gemm) {
        if constexpr (GEMM_OK) {
            if constexpr (NC + NL > 12) return go(L{}, std::true_type{});
            else return lean ? go(L{}, std::true_type{}) : go(F{}, std::true_type{});
        } else {
            return GV_E_UNSUPPORTED;
        }
    }
    if constexpr (NC + NL > 12) return go(L{}, std::false_type{});
    else return lean ? go(L{}, std::false_type{}) : go(F{}, std::false_type{});
}

template <typename T>
int launch_ws_cfg(int cfg, const ConvArgs& a, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_ws<T, 4, 2, 2, 3, 4>(a, st);         // 256 x 192: 8 consumers of 64 x 96
        case 1: return launch_ws<T, 4, 2, 2, 2, 4>(a, st);         // 256 x 128
        case 2: return launch_ws<T, 8, 1, 2, 3, 4>(a, st);         // 512 x 96
        case 3: return launch_ws<T, 8, 1, 2, 2, 4>(a, st);         // 512 x 64
        case 4: return launch_ws<T, 4, 2, 2, 1, 4>(a, st);         // 256 x 64
        // 64-channel k-steps (cin % 64 == 0): 128-byte LDS rows, three filter slots
        case 5: return launch_ws<T, 4, 2, 2, 3, 3, 64>(a, st);     // 256 x 192
        case 6: return launch_ws<T, 4, 2, 2, 2, 3, 64>(a, st);     // 256 x 128
        case 7: return launch_ws<T, 4, 2, 2, 1, 3, 64>(a, st);     // 256 x 64
        case 8: return launch_ws<T, 8, 1, 2, 2, 3, 64>(a, st);     // 512 x 64
        // four consumers + two loaders: two workgroups per CU (their k-loops cover each other's epilogue; Mixed_5's 3x3)
        case 9: return launch_ws<T, 4, 1, 2, 3, 4, 32, 2>(a, st);  // 256 x 96
        case 10: return launch_ws<T, 4, 1, 2, 2, 4, 32, 2>(a, st); // 256 x 64
    }
    return GV_E_UNSUPPORTED;
}
#endif

}  // namespace

#ifndef GV_KERNEL_ONLY
namespace gvconv {

int ws_lp_num_cfgs() { return 11; }

int ws_lp_launch(int dtype, int cfg, const ConvArgs& a, hipStream_t st) {
    static const bool off = getenv("GV_NO_WS") != nullptr;       // (A/B of whole plans: the autotuner then never sees these tiles)
    if (off) return GV_E_UNSUPPORTED;
    if (dtype == GV_BF16) return launch_ws_cfg<__bf16>(cfg, a, st);
    if (dtype == GV_F16) return launch_ws_cfg<_Float16>(cfg, a, st);
    return GV_E_UNSUPPORTED;
}

}  // namespace gvconv
#endif
