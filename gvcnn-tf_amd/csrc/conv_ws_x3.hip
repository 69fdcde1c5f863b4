// conv_ws_x3.hip — the wave-specialised strip convolution (conv_ws.hip) for fp32 values kept as three bf16 planes
// ("P3" input, GV_MATH_BF16X3: the headline configuration's Mixed_5 / Mixed_6 / Mixed_7 layers; reference conv+BN+ReLU
// sites nets/inception_v3.py:226-338).
//
// conv_dma<NP = 3> (the implicit GEMM these layers ran on) issues 6 LDS-DMA instructions per wave and k-tile next to
// 36 MFMAs — 160-200 clocks of issue each, i.e. as long as the MFMAs themselves — and DMAs every input pixel once per
// filter tap.  Here, as in conv_ws.hip:
//   * a tile is BM CONSECUTIVE pixels of the flattened (image, y, x) grid of a stride-1 same-grid convolution; ONE strip
//     of halo + BM + halo pixels per 16-channel chunk serves all taps (tap (dy, dx) of pixel m is strip row m + dy*iw + dx;
//     a per-lane tap mask redirects taps outside the image to zero bytes), only the filter slice is new per tap;
//   * four LOADER waves (one per SIMD) do all address arithmetic and every global_load_lds; eight CONSUMER waves (two per
//     SIMD) only read fragments and issue MFMAs.  One raw s_barrier per k-step is FULL and FREE at once, FULL given one
//     barrier early; loaders wait on counted vmcnt and issue a strip piecewise over the chunk in front of its use.
// What the three planes change: a k-step is 16 channels of one tap = 6 plane products = 36 MFMAs per 64 x 96 consumer
// (1 152 matrix-pipe clocks) for 15 fragment reads, against 12 MFMAs per 5 reads on 16-bit storage — the consumers are
// matrix-bound, and the loaders have twice the time per byte.  Registers: twelve waves leave 168 per wave; 96 hold the
// accumulators, so the fragments are SINGLE buffered (52 registers) and re-loaded one by one behind the MFMA that last
// reads them, each at least a product group (6 MFMAs, 192 clocks) in front of its next use:
//   product order (plane of a, plane of b): (0,2) (0,1) (0,0) (2,0) (1,0) (1,1); planes 2 share one register block T:
//   T = b2 | a0 a0 -> reload a0 | T = a2 -> reload T = next b2 | a1 -> reload b0 | a1 b1 -> reload a1, b1.
//
// LDS: [filter ring: NB slots x 3 planes x BN rows x 32 B][2 strip buffers x 3 planes x (64 zero bytes + SBMAX KiB)];
// rows are 32 bytes (16 channels of one plane), the two 16-byte chunks XOR-swizzled by (row >> 3) & 1 — at the SOURCE
// address by the loader, at the read address by the consumer; any row shift keeps the 16 rows of a ds_read_b128 service
// group on distinct banks.  k order: channel chunk outer, filter tap inner; k-step (c, t) multiplies strip c at shift(t)
// with columns [(t*cin/16 + c)*16, +16) of the packed [cout][K/16][plane][16] filter.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"
#include "conv_lp_epi.h"
#include "conv_x3_epi.h"

namespace {

constexpr int X3W_HEAD = 64;                   // zero bytes in front of every plane image of a strip buffer
constexpr int X3W_NL = 4;                      // loader waves

struct WsX3Args {
    int taps, nchunks, nk;                     // kh*kw, cin/16, taps*nchunks
    int halo_lo;                               // strip rows in front of the tile's first pixel
    int strip_blocks;                          // 32-row blocks of one plane of a strip
    int pad_[3];
};

__device__ __forceinline__ void x3w_dma16(const char* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void x3w_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// GEMM = false: STRIP mode (above; P3 input by LDS-DMA).  GEMM = true: 1x1 convolutions on plain FP32 input (the fused
// sibling GEMMs of an Inception block read the block's fp32 concat buffer): the A operand is a ring of three
// [3 planes][BM rows][32 B] tiles which the loader waves fill through registers — global_load_dwordx4, split into three
// bf16 planes (the 5.5 VALU instructions per element the register-staged kernel's MFMA waves spend in their own loop),
// ds_write_b64 — two k-steps ahead; the filter ring and the consumers are the strip mode's.  (The filter slices
// through registers as well — global_load_dwordx4 + ds_write_b128 instead of five LDS-DMA instructions per k-step — measured
// 2 - 5 % slower: profiles/r5_wsg_filter_through_registers_ab.txt.)
// (GEMM mode's register-staged A loads: see the loader)
__device__ __forceinline__ void x3w_gload16(f32x4& dst, const char* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void x3w_wait_vm_reg(f32x4& r) {       // (one statement per register: the repeated waits are free)
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}

template <int WM, int WN, int TM, int TN, int NB, int SBMAX, int GEMM = 0>
__global__ __launch_bounds__((WM * WN + X3W_NL) * 64, 3) void conv_ws_x3(const ConvArgs a, const WsX3Args w) {
    constexpr int NC = WM * WN;                                    // consumer waves
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int B_PLANE = BN * 32, B_SLOT = 3 * B_PLANE;
    constexpr int A_PLANE = GEMM ? BM * 32 : X3W_HEAD + SBMAX * 1024, S_BUF = 3 * A_PLANE;
    constexpr int NA = GEMM ? 3 : 2;                               // A tiles (GEMM) / strip buffers
    constexpr int OFF_S = NB * B_SLOT;
    constexpr int UB3 = 3 * (BN / 32);                             // filter DMA instructions per k-step (row blocks x planes)
    constexpr int LB = (UB3 + X3W_NL - 1) / X3W_NL;                // ... per loader wave: what the vmcnt counts rely on
    static_assert(NC == 8, "eight consumer waves + four loaders: three waves per SIMD");
    static_assert(TM <= TN && TM * TN >= 4, "plane-2 fragments of a and b share one register block");
    static_assert(NB >= 3 && (NB - 1) * LB < 64, "ring depth / vmcnt range");
    static_assert(GEMM == 0 || (NB == 4 && BM % (16 * X3W_NL) == 0), "GEMM mode: four filter slots, 16-row load instructions");
    (void)NA;

    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = gv_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % a.tiles_n, tile_m = lid / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = w.nk, NTAP = w.taps;
    // timing ablations (results garbage): 16384 consumers alone (no loads, no barriers); 32768 barriers but no loads;
    // 65536 loads but no barriers
    const bool nosync = (a.dbg & (16384 | 65536)) != 0;

    // the epilogue's per-column constants: requested now, published into LDS once the ring is free
    // (by the LOADER waves: two registers a consumer cannot spare)
    static_assert(BN <= X3W_NL * 64, "one loader thread per tile column");
    float ss_v[2] = {0.f, 0.f};
    f32x16 acc[TM][TN];

    if (wave >= NC) {
        // =================================================== loader ===================================================
        const int lw = wave - NC;
        if (tid - NC * 64 < BN) {
            const int cc = min(n0 + tid - NC * 64, a.cout - 1);
            ss_v[0] = a.scale[cc];
            ss_v[1] = a.shift[cc];
        }
        if (!(a.dbg & 16384)) {
            // (conv_ws.hip: the loaders' few instructions go first.  Measured here and neutral: no priority; a higher priority
            // for one of a SIMD's two consumers; an s_sleep behind every DMA instruction, -1 ... -3 %:
            // profiles/r5_x3ws_priority_and_pacing_ab.txt)
            __builtin_amdgcn_s_setprio(3);
            const bool nodma = (a.dbg & 32768) != 0, nobar = (a.dbg & 65536) != 0;
            const int lrow = lane >> 1;                            // row inside a 32-row block
            const int lc16 = (((lane & 1) ^ ((lrow >> 3) & 1)) << 4);   // byte offset of the logical chunk this lane fetches
            const char* xb = reinterpret_cast<const char*>(a.x);
            // filter slice: this loader's instructions d = lw, lw + 4, ... -> (row block d / 3, plane d % 3)
            const char* b_ptr[LB];
            int b_dst[LB];
            {
                const size_t wrow = (size_t)(a.Kpad / 16) * 96;
#pragma unroll
                for (int i = 0; i < LB; ++i) {
                    int d = lw + i * X3W_NL;
                    d = d < UB3 ? d : UB3 - 1;                     // surplus slots re-load the last unit (same bytes)
                    const int rb = d / 3, p = d - rb * 3;
                    int n = n0 + rb * 32 + lrow;
                    n = n < a.cout ? n : a.cout - 1;               // columns past cout are never stored
                    b_ptr[i] = reinterpret_cast<const char*>(a.w) + (size_t)n * wrow + p * 32 + lc16;
                    b_dst[i] = p * B_PLANE + rb * 1024;
                }
            }
            int bq_t = 0, bq_c = 0, bq_slot = 0;                   // next filter slice to issue: tap, chunk, ring slot
            const int gtap = (a.cin >> 4) * 96;                    // bytes between the same chunk of consecutive taps
            auto issue_b = [&]() {
                const unsigned koff = (unsigned)(bq_t * gtap + bq_c * 96);
                char* sb = smem + bq_slot * B_SLOT;
#pragma unroll
                for (int i = 0; i < LB; ++i)
                    if (!nodma) x3w_dma16(b_ptr[i] + koff, sb + b_dst[i]);
                if (++bq_t == NTAP) { bq_t = 0; ++bq_c; }
                bq_slot = bq_slot + 1 == NB ? 0 : bq_slot + 1;
            };
            if constexpr (GEMM != 0) {
                // ---- A tiles through registers: a wave owns BM / 4 rows, an instruction 16 rows x 64 bytes (four lanes per row)
                constexpr int RPW = BM / X3W_NL, NI = RPW / 16;
                const int qd = lane & 3;
                const char* a_ptr[NI];
                int a_dst[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = lw * RPW + i * 16 + (lane >> 2);
                    int m = m0 + row;
                    m = m < a.M ? m : a.M - 1;                     // rows past M: results never stored
                    a_ptr[i] = xb + (size_t)(unsigned)m * ((size_t)a.x_ld * 4) + qd * 16;
                    a_dst[i] = OFF_S + row * 32 + ((((qd >> 1) ^ ((row >> 3) & 1))) << 4) + (qd & 1) * 8;
                }
                // The loads are inline asm with hand-counted waits: a compiler-visible load is waited for with vmcnt(0) across
                // the loop's back edge — the loads of k-step j+3, issued a moment ago, with it.  A stage's registers pass
                // through the wait statement ("+v"), so no use can be scheduled in front of it.
                f32x4 stg[2][NI];
                auto load_a = [&](auto sc, int c) {                // k-step c (past the end: the last one again — never written)
                    constexpr int S = decltype(sc)::value;
                    c = c < nk ? c : nk - 1;
#pragma unroll
                    for (int i = 0; i < NI; ++i) x3w_gload16(stg[S][i], a_ptr[i] + (size_t)c * 64);
                };
                auto wait_a = [&](auto sc, auto nc) {              // all but the N youngest vector-memory operations are done
                    constexpr int S = decltype(sc)::value;
                    constexpr int N = decltype(nc)::value;
#pragma unroll
                    for (int i = 0; i < NI; ++i) x3w_wait_vm_reg<N>(stg[S][i]);
                };
                auto write_a = [&](auto sc, int slot) {            // split (round to nearest even, the exact remainder next) + store
                    constexpr int S = decltype(sc)::value;
                    char* sb = smem + slot * S_BUF;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        float v[4] = {stg[S][i][0], stg[S][i][1], stg[S][i][2], stg[S][i][3]};
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const unsigned lo = pack2<__bf16>(v[0], v[1]), hi = pack2<__bf16>(v[2], v[3]);
                            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                            *reinterpret_cast<u32x2_t*>(sb + p * A_PLANE + a_dst[i]) = u32x2_t{lo, hi};
                            if (p < 2) {
                                v[0] -= __builtin_bit_cast(float, lo << 16);
                                v[1] -= __builtin_bit_cast(float, lo & 0xffff0000u);
                                v[2] -= __builtin_bit_cast(float, hi << 16);
                                v[3] -= __builtin_bit_cast(float, hi & 0xffff0000u);
                            }
                        }
                    }
                };
                using S0 = std::integral_constant<int, 0>;
                using S1 = std::integral_constant<int, 1>;
                // prologue, shaped like four iterations of the loop below: [filter slice k, A loads of k-step k]
                auto fetch = [&](auto sc, int k) {
                    issue_b();
                    load_a(sc, k);
                };
                fetch(S0{}, 0);
                fetch(S1{}, 1);
                wait_a(S0{}, std::integral_constant<int, 0>{});
                wait_a(S1{}, std::integral_constant<int, 0>{});
                write_a(S0{}, 0);
                write_a(S1{}, 1);
                fetch(S0{}, 2);
                fetch(S1{}, 3);                                    // (nk >= 4: the launcher checks)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!nobar) __builtin_amdgcn_s_barrier();          // k-steps 0 and 1 are in LDS
                // iteration j: k-step j+2 into LDS (its loads are one iteration's [LB + NI operations] away from the youngest),
                // barrier j, then filter slice j+4 into the slot k-step j vacated and the loads of k-step j+4 into the registers
                // k-step j+2 vacated
                // (past the last slice the DMAs re-load slice nk-1 into its own slot — the same bytes over themselves —
                // so that every iteration issues the same LB + NI operations: ONE wait statement, whose count holds in the tail
                // too, and through which the stage's registers pass — with two alternative statements the compiler copied the
                // registers in front of one of them, i.e. read them while the loads were in flight)
                auto iter = [&](auto sc, int j) {
                    wait_a(sc, std::integral_constant<int, LB + NI>{});
                    if (j + 2 < nk) {
                        int slot = j + 2;
                        slot -= (slot / 3) * 3;
                        write_a(sc, slot);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if (!nobar) __builtin_amdgcn_s_barrier();
                    if (j + 4 >= nk) {                             // nothing new: slice nk-1 again, in place
                        bq_c = nk - 1;
                        bq_slot = (nk - 1) & 3;
                    }
                    fetch(sc, j + 4);
                };
                for (int j = 0; j + 1 < nk; j += 2) {
                    iter(S0{}, j);
                    if (j + 2 < nk) iter(S1{}, j + 1);
                }
                // the last iterations' filter DMAs (slice nk-1 over itself) and register loads are still in flight, and the
                // epilogue's staging blocks alias the ring: nothing may land behind the barrier in front of it
                x3w_wait_vm<0>();
            } else {
            const unsigned pix_bytes = (unsigned)a.x_ld * 6u;
            // strips: unit e = lw, lw + 4, ... of a chunk -> (block e / 3, plane e % 3); e + 4 = 3 (blk + 1) + (p + 1)
            int sq_c = 0, sq_buf = 0;                              // next strip to begin: chunk, buffer
            int sp_blk = 1 << 30, sp_p = 0, sp_c = 0, sp_buf = 0;  // pending strip: next unit (none: past the end), chunk, buffer
            auto strip_begin = [&]() {
                sp_blk = lw / 3; sp_p = lw - sp_blk * 3; sp_c = sq_c; sp_buf = sq_buf;
                ++sq_c;
                sq_buf ^= 1;
            };
            auto strip_units = [&](int n) {
                char* sb = smem + OFF_S + sp_buf * S_BUF + X3W_HEAD;
                const unsigned coff = (unsigned)sp_c * 96u + (unsigned)lc16;
                for (int i = 0; i < n && sp_blk < w.strip_blocks; ++i) {
                    int p = m0 - w.halo_lo + sp_blk * 32 + lrow;   // rows outside [0, M) are only ever read by masked taps
                    p = p < 0 ? 0 : (p < a.M ? p : a.M - 1);
                    if (!nodma) x3w_dma16(xb + (size_t)(unsigned)p * pix_bytes + coff + sp_p * 32, sb + sp_p * A_PLANE + sp_blk * 1024);
                    ++sp_p; ++sp_blk;
                    if (sp_p == 3) { sp_p = 0; ++sp_blk; }
                }
            };
            // units of a strip per loader, and per k-step so that a strip is complete NB - 2 filter slices in front of its use
            const int upl = (3 * w.strip_blocks + X3W_NL - 1) / X3W_NL;
            const int win = max(NTAP - NB + 2, 1);
            const int ppi = (upl + win - 1) / win;
            // prologue: strip 0 and filter slice 0 first (the consumers' first fragments), then the rest of the ring
            strip_begin();
            strip_units(1 << 20);
            issue_b();
#pragma unroll
            for (int q = 1; q < NB; ++q) issue_b();
            x3w_wait_vm<(NB - 2) * LB>();                          // everything up to filter slice 1 has landed
            if (!nobar) __builtin_amdgcn_s_barrier();
            if (w.nchunks > 1) { strip_begin(); strip_units(ppi); }   // the second strip behind that barrier (conv_ws.hip)
            int ft = 0;                                            // tap of k-step j
            for (int j = 0; j + 1 < nk; ++j) {
                // k-step j+2 must be in LDS before barrier j: the NB-3 slices issued after it may stay in flight (strip units
                // issued in between only make the wait stricter); in the drain wait for everything
                if (j + NB - 1 <= nk - 1) x3w_wait_vm<(NB - 3) * LB>();
                else x3w_wait_vm<0>();
                if (!nobar) __builtin_amdgcn_s_barrier();          // ... and every consumer is done with k-step j's LDS
                const bool last_tap = ft + 1 == NTAP;
                if (last_tap && sq_c < w.nchunks) strip_begin();   // the strip this chunk occupied is free: chunk + 2
                strip_units(ppi);
                if (j + NB < nk) issue_b();
                ft = last_tap ? 0 : ft + 1;
            }
            x3w_wait_vm<0>();                                      // (nothing in flight when the epilogue re-uses the ring)
            }
        }
    } else {
        // ================================================== consumer ==================================================
        const int wm = wave / WN, wn = wave % WN;
        const int r = lane & 31, h = lane >> 5;
        if constexpr (GEMM == 0) {
            if (tid < 96) {                                        // the zero heads of the six plane images
                const int img = tid >> 4;
                *reinterpret_cast<unsigned*>(smem + OFF_S + (img / 3) * S_BUF + (img % 3) * A_PLANE + (tid & 15) * 4) = 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        // this lane's rows: strip row of the un-shifted tap and one bit per tap "inside the image"
        const int rbase0 = wm * TM * 32 + r + w.halo_lo;          // (row block i: + 32 i)
        unsigned tapmask[TM];
        if constexpr (GEMM != 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) tapmask[i] = 0u;          // (unused)
        } else {
            const int ohow = a.oh * a.ow;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int lr = (wm * TM + i) * 32 + r;
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
        const int h16 = h << 4;
        const int b_lane = (wn * TN * 32 + r) * 32 + ((((r >> 3) & 1) << 4) ^ h16);   // + slot; + j * 1024 + plane * B_PLANE
        // plane-0 fragment address of row block i for the tap at row shift `off` / mask bit `tap`, strip buffer at byte
        // `sbase` (+ plane * A_PLANE); a tap outside the image reads the buffer's zero head
        // (bit arithmetic, not a select: conv_ws.hip)
        auto a_addr = [&](int i, int off, int tap, int sbase) -> int {
            if constexpr (GEMM != 0) return sbase + ((wm * TM + i) * 32 + r) * 32 + ((((r >> 3) & 1) << 4) ^ h16);   // (sbase: the A tile)
            const int row = rbase0 + i * 32 + off;
            const int ad = sbase + X3W_HEAD + row * 32 + ((((row >> 3) & 1) << 4) ^ h16);
            const int in = -(int)((tapmask[i] >> tap) & 1u);       // all ones: the tap lies inside the image
            return (ad & in) | ((sbase + h16) & ~in);
        };
        auto lds16 = [&](int addr) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + addr); };
        u32x4 A0[TM], A1[TM], B0[TN], B1[TN], T[TN];
        // one k-step on the fragments in registers.  aa: this k-step's A addresses on entry (its plane 2 is read with them in
        // the first product group); next_addr(), called behind that group, replaces aa / bn by the NEXT k-step's, which every
        // re-load uses (NEXT: re-load every fragment behind its last use).  bar: the k-step's barrier behind the second MFMA
        // of the last product group — by then every fragment of this k-step has been an MFMA operand, i.e. every LDS read
        // of it has returned, and the wave has matrix work queued while it waits
        // (the compiler's scheduler would gather the reads into one burst in front of the barrier and move the MFMAs around
        // them: sched_barrier pins every read behind the MFMA that frees its register)
#define X3W_PIN() __builtin_amdgcn_sched_barrier(0)
        auto kstep = [&](auto nextc, int (&aa)[TM], int& bn, bool bar, auto&& next_addr) {
            constexpr bool NEXT = decltype(nextc)::value;
            constexpr int NM = TM * TN;
            X3W_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {                         // (0,2): a0 x T = b2; T <- this k-step's a2
                const int i = m / TN, j = m % TN;
                acc[i][j] = mfma16<__bf16>(A0[i], T[j], acc[i][j]);
                if (i == TM - 1 && j < TM) { T[j] = lds16(aa[j] + 2 * A_PLANE); X3W_PIN(); }
            }
            X3W_PIN();
            next_addr();                                           // aa, bn <- the next k-step's addresses (scheduled among these MFMAs)
#pragma unroll
            for (int m = 0; m < NM; ++m) {                         // (0,1)
                const int i = m / TN, j = m % TN;
                acc[i][j] = mfma16<__bf16>(A0[i], B1[j], acc[i][j]);
            }
            X3W_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {                         // (0,0); a0 <- next; (b2's columns a2 does not occupy)
                const int i = m / TN, j = m % TN;
                acc[i][j] = mfma16<__bf16>(A0[i], B0[j], acc[i][j]);
                if constexpr (NEXT) {
                    if (i == 0 && j >= TM && j < TN) { T[j] = lds16(bn + j * 1024 + 2 * B_PLANE); X3W_PIN(); }
                    if (j == TN - 1) { A0[i] = lds16(aa[i]); X3W_PIN(); }
                }
            }
            X3W_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {                         // (2,0): T = a2; T <- next b2
                const int i = m / TN, j = m % TN;
                acc[i][j] = mfma16<__bf16>(T[i], B0[j], acc[i][j]);
                if constexpr (NEXT) {
                    if (j == TN - 1) { T[i] = lds16(bn + i * 1024 + 2 * B_PLANE); X3W_PIN(); }
                }
            }
            X3W_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {                         // (1,0); b0 <- next
                const int i = m / TN, j = m % TN;
                acc[i][j] = mfma16<__bf16>(A1[i], B0[j], acc[i][j]);
                if constexpr (NEXT) {
                    if (i == TM - 1) { B0[j] = lds16(bn + j * 1024); X3W_PIN(); }
                }
            }
            X3W_PIN();
#pragma unroll
            for (int m = 0; m < NM; ++m) {                         // (1,1); a1, b1 <- next
                const int i = m / TN, j = m % TN;
                acc[i][j] = mfma16<__bf16>(A1[i], B1[j], acc[i][j]);
                if constexpr (NEXT) {
                    if (m == 1) {
                        X3W_PIN();
                        if (bar) __builtin_amdgcn_s_barrier();     // k-step j+2 is in LDS; k-step j's LDS may be overwritten
                        X3W_PIN();
                    }
                    if (j == TN - 1) { A1[i] = lds16(aa[i] + A_PLANE); X3W_PIN(); }
                    if (i == TM - 1) { B1[j] = lds16(bn + j * 1024 + B_PLANE); X3W_PIN(); }
                }
            }
            X3W_PIN();
        };
        int q_fr = 0, q_fs = 0, q_tap = 0;                         // tap of the k-step whose addresses are in aa[]
        int sbuf = 0, bslot = 0;                                   // its strip buffer (index) / filter slot (bytes)
        int aa[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) aa[i] = a_addr(i, -(a.pad_t * a.iw + a.pad_l), 0, OFF_S);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!nosync) __builtin_amdgcn_s_barrier();                 // strip 0, filter slices 0 and 1 and the zero heads are in LDS
#pragma unroll
        for (int i = 0; i < TM; ++i) { A0[i] = lds16(aa[i]); A1[i] = lds16(aa[i] + A_PLANE); }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            B0[j] = lds16(b_lane + j * 1024);
            B1[j] = lds16(b_lane + j * 1024 + B_PLANE);
            T[j] = lds16(b_lane + j * 1024 + 2 * B_PLANE);
        }
        int bn = 0;
        // the next k-step's tap, strip buffer and slot (selects, no branches: conv_ws.hip)
        auto next_addr = [&]() {
            if constexpr (GEMM != 0) {
                sbuf = sbuf + 1 == NA ? 0 : sbuf + 1;
#pragma unroll
                for (int i = 0; i < TM; ++i) aa[i] = a_addr(i, 0, 0, OFF_S + sbuf * S_BUF);
                bslot = bslot + B_SLOT == NB * B_SLOT ? 0 : bslot + B_SLOT;
                bn = b_lane + bslot;
                return;
            }
            const bool wrap = q_tap + 1 == NTAP;
            const bool row_end = q_fs + 1 == a.kw;
            q_tap = wrap ? 0 : q_tap + 1;
            q_fr = wrap ? 0 : (row_end ? q_fr + 1 : q_fr);
            q_fs = (wrap || row_end) ? 0 : q_fs + 1;
            sbuf = wrap ? sbuf ^ 1 : sbuf;
            const int sbase = OFF_S + sbuf * S_BUF;
            const int off = (q_fr - a.pad_t) * a.iw + (q_fs - a.pad_l);
#pragma unroll
            for (int i = 0; i < TM; ++i) aa[i] = a_addr(i, off, q_tap, sbase);
            bslot = bslot + B_SLOT == NB * B_SLOT ? 0 : bslot + B_SLOT;
            bn = b_lane + bslot;
        };
        for (int j = 0; j + 1 < nk; ++j) {
            kstep(std::true_type{}, aa, bn, !nosync, next_addr);
        }
        kstep(std::false_type{}, aa, bn, false, [] {});
#undef X3W_PIN
    }
    // ====================================================== epilogue ======================================================
    __syncthreads();                                               // every wave is done with the ring: the staging blocks alias it
    constexpr int SS_OFF = NC * X3EpiGeom<TN>::BYTES;
    float* sstab = reinterpret_cast<float*>(smem + SS_OFF);
    if (tid >= NC * 64 && tid - NC * 64 < BN) {
        sstab[tid - NC * 64] = ss_v[0];
        sstab[BN + tid - NC * 64] = ss_v[1];
    }
    __syncthreads();
    if (wave < NC) {
        const int wm = wave / WN, wn = wave % WN;
        x3_epilogue_staged<TM, TN>(a, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem + wave * X3EpiGeom<TN>::BYTES), sstab, BN);
    }
}

#ifndef GV_KERNEL_ONLY
// the staged epilogue's destinations (conv_x3_epi.h): 8-column chunks, 16-byte fp32 stores
bool ws_x3_epi_ok(const ConvArgs& a) {
    if (a.cout % 8 != 0 || (a.y2 != nullptr && a.split == 0)) return false;
    if (a.res != nullptr && (a.res_ld % 4 != 0 || !gv_aligned16(a.res))) return false;
    if (!a.y_p3 && (a.y_ld % 4 != 0 || !gv_aligned16(a.y))) return false;
    if (a.split > 0 && (a.split % 8 != 0 || (!a.y2_p3 && (a.y2_ld % 4 != 0 || !gv_aligned16(a.y2))))) return false;
    return true;
}

// STRIP mode: stride 1, output grid = input grid, 2 ... 32 taps, whole 16-channel groups of three-plane input
bool ws_x3_shape_ok(const ConvArgs& a) {
    return a.stride == 1 && a.dil_shift == 0 && a.oh == a.ih && a.ow == a.iw && a.cin % 16 == 0 && a.x_ld % 16 == 0 &&
           a.kh * a.kw <= 32 && a.kw < 32 && a.pad_t < a.kh && a.pad_l < a.kw && a.pool == 0 && a.xscale == nullptr &&
           a.y_step == 0 && a.st.mode == gvconv::STAT_OFF && ws_x3_epi_ok(a);
}

template <int WM, int WN, int TM, int TN, int NB, int SBMAX>
int launch_ws_x3(const ConvArgs& a0, hipStream_t st) {
    constexpr int NC = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    if (!ws_x3_shape_ok(a0)) return GV_E_UNSUPPORTED;
    ConvArgs a = a0;
    a.Kpad = a.K;                                                  // (cin % 16 == 0: the packed filter has no padding)
    WsX3Args w;
    w.taps = a.kh * a.kw;
    w.nchunks = a.cin / 16;
    w.nk = w.taps * w.nchunks;
    if (w.nk < NB || w.taps < NB - 1) return GV_E_UNSUPPORTED;     // (the vmcnt counts of the loader assume it)
    w.halo_lo = a.pad_t * a.iw + a.pad_l;
    const int halo_hi = (a.kh - 1 - a.pad_t) * a.iw + (a.kw - 1 - a.pad_l);
    w.strip_blocks = gv_ceil_div(w.halo_lo + BM + halo_hi, 32);
    if (w.strip_blocks > SBMAX) return GV_E_UNSUPPORTED;
    w.pad_[0] = w.pad_[1] = w.pad_[2] = 0;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    const int64_t nwg = (int64_t)gv_ceil_div(a.M, BM) * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    constexpr size_t ring = (size_t)NB * 96 * BN + 6 * (size_t)(X3W_HEAD + SBMAX * 1024);
    constexpr size_t epi = (size_t)NC * X3EpiGeom<TN>::BYTES + 2 * BN * sizeof(float);
    constexpr size_t lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    auto kern = &conv_ws_x3<WM, WN, TM, TN, NB, SBMAX, 0>;
    if (lds > 64 * 1024) {
        const bool ok = GV_BIG_LDS_OK(kern, 160 * 1024);
        if (!ok) return GV_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3((NC + X3W_NL) * 64), lds, st, a, w);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

// GEMM mode: 1x1 / stride 1 on plain fp32 input, whole 16-channel groups, 16-byte aligned pixels
template <int WM, int WN, int TM, int TN, int MODE>
int launch_wsg_x3(const ConvArgs& a0, hipStream_t st) {
    constexpr int NC = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32, NB = 4;
    const ConvArgs& c = a0;
    if (!(c.kh == 1 && c.kw == 1 && c.stride == 1 && c.pad_t == 0 && c.pad_l == 0 && c.dil_shift == 0 && c.oh == c.ih &&
          c.ow == c.iw && c.cin % 16 == 0 && c.x_ld % 4 == 0 && gv_aligned16(c.x) && c.pool == 0 && c.xscale == nullptr &&
          c.y_step == 0 && c.st.mode == gvconv::STAT_OFF && ws_x3_epi_ok(c)))
        return GV_E_UNSUPPORTED;
    ConvArgs a = a0;
    a.Kpad = a.K;
    WsX3Args w;
    w.taps = 1;
    w.nchunks = a.cin / 16;
    w.nk = w.nchunks;
    if (w.nk < 4) return GV_E_UNSUPPORTED;                         // (the loader's prologue issues four k-steps)
    w.halo_lo = 0;
    w.strip_blocks = 0;
    w.pad_[0] = w.pad_[1] = w.pad_[2] = 0;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    const int64_t nwg = (int64_t)gv_ceil_div(a.M, BM) * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    constexpr size_t ring = (size_t)NB * 96 * BN + 9 * (size_t)BM * 32;
    constexpr size_t epi = (size_t)NC * X3EpiGeom<TN>::BYTES + 2 * BN * sizeof(float);
    constexpr size_t lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    auto kern = &conv_ws_x3<WM, WN, TM, TN, NB, 0, MODE>;
    if (lds > 64 * 1024) {
        const bool ok = GV_BIG_LDS_OK(kern, 160 * 1024);
        if (!ok) return GV_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3((NC + X3W_NL) * 64), lds, st, a, w);
    GV_LAUNCH_CHECK();
    return GV_OK;
}
#endif

}  // namespace

#ifndef GV_KERNEL_ONLY
namespace gvconv {

int ws_x3_num_cfgs() { return 5; }

int ws_x3_launch(int cfg, const ConvArgs& a, hipStream_t st) {
    static const bool off = getenv("GV_NO_WS") != nullptr;       // (A/B of whole plans: the autotuner then never sees these tiles)
    if (off) return GV_E_UNSUPPORTED;
    switch (cfg) {
        case 0: return launch_ws_x3<4, 2, 2, 3, 4, 13>(a, st);     // 256 x 192: 8 consumers of 64 x 96 (154 KB)
        case 1: return launch_ws_x3<4, 2, 2, 2, 4, 13>(a, st);     // 256 x 128
        case 2: return launch_ws_x3<8, 1, 2, 3, 3, 21>(a, st);     // 512 x 96 (Mixed_5's 3x3 layers)
        case 3: return launch_ws_x3<8, 1, 2, 2, 3, 21>(a, st);     // 512 x 64 (Mixed_5's 5x5 layers)
        case 4: return launch_ws_x3<8, 1, 1, 5, 4, 13>(a, st);     // 256 x 160: 8 consumers of 32 x 160 (Mixed_6c / 6d's 160-column layers)
    }
    return GV_E_UNSUPPORTED;
}

// fp32 input (the register-staged kernel's class, conv_bf16s.hip): the GEMM mode
int wsg_x3_num_cfgs() { return 2; }

int wsg_x3_launch(int cfg, const ConvArgs& a, hipStream_t st) {
    static const bool off = getenv("GV_NO_WS") != nullptr;
    if (off) return GV_E_UNSUPPORTED;
    switch (cfg) {
        case 0: return launch_wsg_x3<4, 2, 2, 3, 1>(a, st);        // 256 x 192 (144 KB)
        case 1: return launch_wsg_x3<4, 2, 2, 2, 1>(a, st);        // 256 x 128
    }
    return GV_E_UNSUPPORTED;
}

}  // namespace gvconv
#endif
