// conv_chain.hip — the TAIL of one ResNet-v2 bottleneck unit and the HEAD of the next as ONE launch (round 6).
// Three forms of one kernel: the chain (gv_bottleneck_chain_fwd: conv3 + shortcut, next pre-activation, next conv1), the
// whole unit (gv_bottleneck_unit_fwd: FRONT = the unit's own conv2 3x3 in front of the chain) and the chain's second half
// alone (TAIL: gv_conv2d_fwd_xpre's class 1x1 / cin = 4 cout, offered as a tile configuration of that entry point).
//
// Reference: nets/resnet_v2.py:87-91 (conv3 1x1 + bias, `output = shortcut + residual`) of unit u, then, in unit u+1,
// :75 (`preact = batch_norm(inputs, activation_fn=relu)`) and :83-84 (conv1 1x1 -> BatchNorm -> ReLU).  Between two units
// of one block the shortcut is the identity (resnet_v2.py:76-77), so the only reader of `preact` is conv1.
//
// Why.  At 56x56 / 28x28 every 1x1 convolution of the bottleneck is HBM-bound (K = 64 ... 512 against 4x as many output
// channels): as separate launches the unit's output (4d channels) is written by conv3 and read back by the next conv1 —
// 40 % of the bytes the pair moves.  Here a workgroup keeps its rows of `out` on chip: GEMM 1 (d -> 4d) is evaluated in
// 64-column chunks; a chunk leaves through the staged epilogue (+ bias + shortcut, ONE rounding, 16-byte stores: a row's
// 64 columns are one whole 128-byte line), the rounded values go through the next unit's folded BatchNorm + ReLU — exactly
// what gv_conv2d_fwd_xpre's loader computes from the stored tensor — into a wave-private LDS tile, and are at once the A
// operand of GEMM 2's partial sum over those 64 channels (4d -> d).  `out` is written once and never read back; the
// pre-activation never exists in memory.
//
// Structure.  Memory-bound streaming, so no operand re-use tricks: a wave owns 32 rows (pixels) for ALL columns — the
// chain needs no cross-wave hand-off; its rows of x stay in registers as A fragments (direct 16-byte global loads).  The
// two filters are streamed chunk by chunk through a two-slot LDS ring by LDS-DMA, shared by the workgroup's waves (one
// barrier per chunk: FULL and FREE at once; chunk c + 1 lands while chunk c is multiplied, behind a counted vmcnt).  The
// shortcut is prefetched one or two chunks ahead into registers.  d = 64: 4 waves, 68.5 KB of LDS, two workgroups per CU;
// d = 128: 8 waves, 137 KB, one.  Values: the same products in the same k order as the two launches (k ascending in
// 16-steps, fp32 accumulators, v = acc*scale + shift (+ residual), one rounding) — bit for bit.
#include <type_traits>

#include "conv_common.h"
#include "conv_lp_epi.h"

namespace {

struct ChainArgs {
    const unsigned short* x;       // [M][x_ld]   d channels: the unit's conv2 output (after BatchNorm + ReLU)
    const unsigned short* w1;      // packed [4d][d]: conv3
    const float* sc1;              // [4d] scale / shift of conv3 (bias only: scale = 1)
    const float* sh1;
    const unsigned short* res;     // [M][res_ld] 4d channels: the shortcut
    unsigned short* y;             // [M][y_ld]   4d channels: shortcut + residual (the unit's output)
    const float* psc;              // [4d] the next unit's pre-activation BatchNorm, folded
    const float* psh;
    const unsigned short* w2;      // packed [d][4d]: the next unit's conv1
    const float* sc2;              // [d] its BatchNorm, folded
    const float* sh2;
    unsigned short* z;             // [M][z_ld]   d channels: relu(bn(conv1(relu(bn(y)))))
    int M, x_ld, res_ld, y_ld, z_ld;
    int relu2;
    int dbg;
    // FRONT (gv_bottleneck_unit_fwd): x is the unit's conv1 output and the kernel begins with the unit's conv2 — 3x3, stride 1,
    // SAME, BatchNorm + ReLU (nets/resnet_v2.py:85-86) — whose 32 rows per wave never leave the registers
    const unsigned short* w0;      // packed [d][9 * d]: conv2 (k = tap * d + channel)
    const float* sc0;              // [d] its BatchNorm, folded
    const float* sh0;
    const char* zeros;             // a zero page (>= 512 bytes) for the taps outside the image
    int ih, iw;
    GvFastDiv div_img, div_row;    // m / (ih * iw), rem / iw
};

// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KiB of LDS at lds_addr), as INLINE ASM: with the builtin the compiler's
// wait-count pass answers every later use of a loaded REGISTER with s_waitcnt vmcnt(0) while an LDS-DMA is in flight (measured
// on this kernel: the shortcut prefetch two chunks ahead was waited for together with the DMA issued a moment before).
// Hidden from the pass, the DMAs only make its counts conservative (more operations are in flight than it believes, so its
// vmcnt(N) waits for at least what it meant to); the DMAs' own landing is waited for by hand (ch_wait_vm).
__device__ __forceinline__ void ch_dma16(const char* gsrc, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void ch_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename F, int... Cs>
__device__ __forceinline__ void ch_for_each(F&& f, std::integer_sequence<int, Cs...>) {
    (f(std::integral_constant<int, Cs>{}), ...);
}

// A CHUNK is 64 columns of GEMM 1 (= 64 k of GEMM 2): a row's 64 columns are ONE 128-byte line of the shortcut and of y, so
// every line is read / written whole by 8 lanes of one instruction.  (The first form of this kernel worked in 32-column
// chunks: 64-byte halves of a line in two chunks ~8000 clocks apart; by then the half-used line had left the XCD's L2 and
// was fetched / written back twice — same-box experiment profiles/r6_chain_full_lines_ab.txt: 0.379 -> 0.320 ms.)
// KX = 2 (round 6, the PROJECTION form: a unit whose depth changes, nets/resnet_v2.py:79-81): GEMM 1's K is 2d — x holds
// [the unit's conv2 output | the unit's pre-activation] and W1 the K-concatenated [conv3 ; projection shortcut] filter, so
// that  conv3(x2) + shortcut(x0)  is ONE accumulation and no shortcut tensor exists (no residual operand: NORES).
template <int D, int KX = 1> struct ChainGeom {
    static constexpr int K1 = D * KX, N1 = 4 * D, N2 = D;
    static constexpr int KS1 = K1 / 16;            // MFMA k-steps of GEMM 1
    static constexpr int NB2 = N2 / 32;            // 32-column accumulator blocks of GEMM 2
    static constexpr int NCH = N1 / 64;            // chunks
    static constexpr int RBW1 = K1 * 2;            // bytes of one row of a W1 chunk ([64 columns][K1])
    static constexpr int CPR1 = RBW1 / 16;         // 16-byte pieces per row
    static constexpr int RPP1 = 1024 / RBW1;       // rows per DMA instruction
    static constexpr int W1C = 64 * RBW1;          // bytes of a W1 chunk
    static constexpr int W2C = N2 * 128;           // bytes of a W2 chunk ([N2 columns][64 k]: 128-byte rows)
    static constexpr int SLOT = W1C + W2C;
    static constexpr int NP1 = W1C / 1024, NPC = SLOT / 1024;
    static constexpr int NR = 2;                   // ring slots: chunk c + 1 lands while chunk c is multiplied
    static constexpr int STAGE = 32 * 64 * 4;      // a wave's staging block: 32 rows x 64 columns fp32 (its first half doubles
                                                   // as the wave's z tile: 32 rows x 64 channels of 16 bits)
    // swizzle of a row's 16-byte pieces: 128-byte rows (r >> 1) & 7, 256-byte and longer rows r & 15 — the 16 rows of a
    // ds_read_b128 service group then fall on 16 different bank groups
    __host__ __device__ static constexpr int swz1(int r) { return CPR1 == 8 ? ((r >> 1) & 7) : (r & 15); }
    __host__ __device__ static constexpr int swz128(int r) { return (r >> 1) & 7; }
    // FRONT: a conv2 tap's filter tile is [d columns][d channels] = d rows of RBW1 bytes (the shape of W1's rows); a slot
    // takes TPS taps, the nine taps arrive in NPRE "pre-chunks" in front of the NCH chunks
    static constexpr int TAPB = D * RBW1;
    static constexpr int TPS = SLOT / TAPB > 0 ? SLOT / TAPB : 1;   // (d = 256 has no FRONT form: TAIL only)
    static constexpr int NPRE = (9 + TPS - 1) / TPS;
    // FRONT: fragment loads run FD taps ahead (a d = 64 tap is 8 MFMAs, ~260 clocks, an L2-hit load 500 - 800: one tap of lead
    // stalled every tap; d = 128's tap is 32 MFMAs and its fragments are 32 registers: one tap ahead).  front_loads: the loads
    // issued while the taps [lo, hi) are multiplied (tap t requests tap t + FD)
    static constexpr int FD = D <= 64 ? 3 : 1;
    static constexpr int front_loads(int lo, int hi) { int n = 0; for (int t = lo; t < hi && t < 9; ++t) n += t + FD < 9 ? KS1 : 0; return n; }
    template <int NW> static constexpr int lds_bytes() { return NR * SLOT + NW * STAGE + (4 * N1 + 4 * N2) * 4; }
};

// RVS: register sets of the shortcut prefetch (2: a chunk's shortcut is requested two chunks ahead; 1: one chunk ahead)
// TAIL: only the SECOND half — z = relu(bn(conv1x1(relu(y * pscale + pshift)))) with y read from memory through the prefetch
// registers (a.res = y): gv_conv2d_fwd_xpre's class 4d -> d as a streaming launch (no GEMM 1, no store of y; the ring carries
// conv1's filter only).  Serves the identity units the chain does not (d = 256: block3 of ResNet-v2-50).
template <typename T, int D, int NW, int RVS, bool FRONT = false, bool TAIL = false, int KX = 1>
__global__ __launch_bounds__(NW * 64, 2) void conv_chain_lp(const ChainArgs a) {
    using G = ChainGeom<D, KX>;
    constexpr bool NORES = KX > 1;                 // the shortcut is part of GEMM 1: nothing to prefetch, nothing to add
    static_assert(!(FRONT && TAIL), "one or the other");
    static_assert(KX == 1 || (!FRONT && !TAIL), "the projection form: the plain chain only");
    constexpr int N1 = G::N1, N2 = G::N2, KS1 = G::KS1, NB2 = G::NB2, NCH = G::NCH;
    constexpr int RBW1 = G::RBW1, CPR1 = G::CPR1, RPP1 = G::RPP1;
    constexpr int W1C = TAIL ? 0 : G::W1C, SLOT = W1C + G::W2C, NP1 = W1C / 1024, NPC = SLOT / 1024;
    constexpr int NR = G::NR, STAGE = G::STAGE;
    constexpr int PPW = NPC / NW;                  // DMA instructions per wave and chunk
    static_assert(NPC % NW == 0 && NCH >= 2 && NB2 % 2 == 0, "chunk geometry");
    constexpr int OFF_STAGE = NR * SLOT, OFF_TAB = OFF_STAGE + NW * STAGE;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = blockIdx.x * (NW * 32) + wave * 32;     // this wave's first row
    float* const tab = reinterpret_cast<float*>(smem + OFF_TAB);   // [sc1 N1][sh1 N1][psc N1][psh N1][sc2 N2][sh2 N2]
    for (int i = tid; i < N1; i += NW * 64) {
        if constexpr (!TAIL) {
            tab[i] = a.sc1[i];
            tab[N1 + i] = a.sh1[i];
        }
        tab[2 * N1 + i] = a.psc[i];
        tab[3 * N1 + i] = a.psh[i];
    }
    for (int i = tid; i < N2; i += NW * 64) {
        tab[4 * N1 + i] = a.sc2[i];
        tab[4 * N1 + N2 + i] = a.sh2[i];
        if constexpr (FRONT) {
            tab[4 * N1 + 2 * N2 + i] = a.sc0[i];
            tab[4 * N1 + 3 * N2 + i] = a.sh0[i];
        }
    }
    // ---- the filter ring: this wave's DMA instructions of a chunk (piece = wave + i * NW: fixed per wave) ----
    const char* wsrc[PPW];
    int winc[PPW], wdst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + i * NW;
        if (piece < NP1) {                                 // rows of W1: [column n][K1], a chunk = 64 consecutive rows
            const int r = piece * RPP1 + lane / CPR1, c = lane % CPR1;
            wsrc[i] = reinterpret_cast<const char*>(a.w1) + (size_t)r * RBW1 + ((c ^ G::swz1(r)) << 4);
            winc[i] = 64 * RBW1;
        } else {                                           // 128-byte pieces of W2's rows: [column n2][64 k of the chunk]
            const int q = piece - NP1;
            const int r = q * 8 + (lane >> 3), c = lane & 7;
            wsrc[i] = reinterpret_cast<const char*>(a.w2) + (size_t)r * (N1 * 2) + ((c ^ G::swz128(r)) << 4);
            winc[i] = 128;
        }
        wdst[i] = piece * 1024;
    }
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);   // LDS address of the ring
    auto issue_chunk = [&](int cc, int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            ch_dma16(wsrc[i] + (size_t)cc * winc[i], __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT + wdst[i]));
    };
    constexpr int NPRE = FRONT ? G::NPRE : 0;                      // ring steps in front of chunk 0
    constexpr int TPS = G::TPS, TAPB = G::TAPB;
    const int r32 = lane & 31, h = lane >> 5;
    auto lds16 = [&](const char* p) -> u32x4 { return *reinterpret_cast<const u32x4*>(p); };
    auto lds16f = [&](const char* p) -> f32x4 { return *reinterpret_cast<const f32x4*>(p); };
    const int col_l = lane & 31, row_h = 4 * (lane >> 5);
    char* const stage = smem + OFF_STAGE + wave * STAGE;
    // ---- epilogue geometry: a lane holds 8 consecutive columns (c8) of row 8 * pass + r8 of a 32 x 64 block ----
    const int r8 = lane >> 3, c8 = lane & 7;
    // The z tile ([32 rows][64 channels] of 16 bits, 4 KB) is the FIRST HALF of the staging block: pass p writes z rows
    // 8p ... 8p + 7 = staging rows 4p ... 4p + 3, which pass p / 2 has read — LDS operations of one wave execute in order,
    // so nothing is overwritten before it was read.
    char* const ztile = stage;
    const int woff = (row_h * 64 + col_l) * 4;                     // + row * 256 + block * 128 per accumulator register
    const int roff = (r8 * 64 + c8 * 8) * 4;                       // + pass * 2048
    // rows past M are clamped to row M - 1 everywhere: such lanes compute that row's values from that row's operands and
    // store them to that row — duplicates of the same bytes — so that EVERY vector-memory operation is issued
    // unconditionally (a store under an exec branch makes the compiler's count of operations in flight path-dependent)
    const unsigned short* rrow_p[4];
    unsigned short* yrow[4];
    unsigned short* zrow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const size_t m = (size_t)min(mw + p * 8 + r8, a.M - 1);
        rrow_p[p] = NORES ? nullptr : a.res + m * a.res_ld + c8 * 8;
        yrow[p] = a.y + m * a.y_ld + c8 * 8;
        zrow[p] = a.z + m * a.z_ld + c8 * 8;
    }
    // the shortcut's first chunks: requested up front — with FRONT in front of the conv2 phase,
    // so that their HBM round trip runs under that phase's matrix work
    constexpr bool EARLY_RV = FRONT && D <= 64;              // (d = 128: the conv2 phase has no 16 registers to spare)
    u32x4 rv[RVS][4];
    auto load_rv = [&]() {
        if constexpr (!NORES) {
#pragma unroll
            for (int q = 0; q < RVS; ++q)
#pragma unroll
                for (int p = 0; p < 4; ++p) rv[q][p] = *reinterpret_cast<const u32x4*>(rrow_p[p] + q * 64);
        }
    };
    constexpr int RVL = NORES ? 0 : 4;             // shortcut loads a chunk issues / the prologue issues per register set
    u32x4 xa[KS1];
    if constexpr (TAIL) {
        issue_chunk(0, 0);
    } else if constexpr (!FRONT) {
        issue_chunk(0, 0);
        // ---- this wave's rows of x as A fragments (lane: row lane & 31, k 8 * (lane >> 5) ... + 7 of each 16-step) ----
        const int row = min(mw + r32, a.M - 1);
        const unsigned short* xp = a.x + (size_t)row * a.x_ld + h * 8;
#pragma unroll
        for (int s = 0; s < KS1; ++s) xa[s] = *reinterpret_cast<const u32x4*>(xp + s * 16);
    } else {
        // ======================= FRONT: conv2 (3x3 / 1, SAME) + BatchNorm + ReLU of this wave's 32 pixels =======================
        // filter tiles: tap t of pre-chunk pc sits at slot + (t % TPS) * TAPB as [d rows (columns of the GEMM)][RBW1 bytes],
        // pieces swizzled like W1's; its DMA pieces: piece = wave + i * NW of the slot's NPC (the last pre-chunk may be short)
        const char* fsrc[PPW];
        int fdst[PPW];
        bool flast_ok[PPW];                                        // does this piece exist in the LAST (short) pre-chunk
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + i * NW;
            const int ts = piece / (TAPB / 1024), pp = piece % (TAPB / 1024);   // tap inside the slot, piece inside the tap
            const int r = pp * RPP1 + lane / CPR1, c = lane % CPR1;
            fsrc[i] = reinterpret_cast<const char*>(a.w0) + (size_t)r * (9 * RBW1) + (size_t)ts * RBW1 + ((c ^ G::swz1(r)) << 4);
            fdst[i] = piece * 1024;
            flast_ok[i] = (NPRE - 1) * TPS + ts < 9;
        }
        auto issue_pre = [&](auto pcc, int slot) {
            constexpr int pc = decltype(pcc)::value;
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                if (pc + 1 < NPRE || flast_ok[i])                  // (wave-uniform; the last pre-chunk may hold fewer taps)
                    ch_dma16(fsrc[i] + (size_t)pc * TPS * RBW1, __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT + fdst[i]));
        };
        issue_pre(std::integral_constant<int, 0>{}, 0);
        // this lane's pixel (fragment layout: row r32) and which of its nine taps lie inside the image
        const int m = min(mw + r32, a.M - 1);
        const int n = gv_div(m, a.div_img);
        const int rem = m - n * (a.ih * a.iw);
        const int py = gv_div(rem, a.div_row);
        const int px = rem - py * a.iw;
        unsigned tapmask = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
            if (yy >= 0 && yy < a.ih && xx >= 0 && xx < a.iw) tapmask |= 1u << t;
        }
        const char* pm = reinterpret_cast<const char*>(a.x) + ((size_t)m * a.x_ld + h * 8) * 2;
        const char* zp = a.zeros + h * 16;
        const long long row_b = (long long)a.iw * a.x_ld * 2, pix_b = (long long)a.x_ld * 2;
        constexpr int FD = G::FD, FS = FD + 1;                     // lead of the fragment loads in taps, register sets
        u32x4 fa[FS][KS1];
        auto tap_load = [&](auto tc, u32x4 (&dst)[KS1]) {
            constexpr int t = decltype(tc)::value;
            const long long off = (t / 3 - 1) * row_b + (t % 3 - 1) * pix_b;           // (wave-uniform)
            const char* src = ((tapmask >> t) & 1u) ? pm + off : zp;
#pragma unroll
            for (int s = 0; s < KS1; ++s) dst[s] = *reinterpret_cast<const u32x4*>(src + s * 32);
        };
        ch_for_each([&](auto tc) { tap_load(tc, fa[decltype(tc)::value % FS]); }, std::make_integer_sequence<int, FD>{});
        if constexpr (EARLY_RV) load_rv();
        f32x16 acc0[NB2];
#pragma unroll
        for (int j = 0; j < NB2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc0[j][q] = 0.f;
        int f_row[NB2], f_sw[NB2];
#pragma unroll
        for (int j = 0; j < NB2; ++j) { f_row[j] = (j * 32 + r32) * RBW1; f_sw[j] = G::swz1(j * 32 + r32); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the constants table is written
        auto pre = [&](auto pcc) {
            constexpr int pc = decltype(pcc)::value;
            // the taps of pre-chunk pc have landed: their DMAs were issued at the top of pre-chunk pc - 1 (pc = 0: in front of
            // the first tap's loads); behind them came only the fragment loads of the taps that pre-chunk started
            constexpr int t_lo = pc * TPS, t_hi = (pc + 1) * TPS < 9 ? (pc + 1) * TPS : 9;
            if constexpr (pc == 0) ch_wait_vm<FD * KS1 + (EARLY_RV ? 4 * RVS : 0)>();
            else ch_wait_vm<G::front_loads((pc - 1) * TPS, pc * TPS)>();
            __builtin_amdgcn_s_barrier();
            if constexpr (pc + 1 < NPRE) issue_pre(std::integral_constant<int, pc + 1>{}, (pc + 1) % NR);
            else issue_chunk(0, NPRE % NR);
            const char* slot = smem + (pc % NR) * SLOT;
            ch_for_each([&](auto tc) {
                constexpr int t = t_lo + decltype(tc)::value;
                if constexpr (t + FD < 9) tap_load(std::integral_constant<int, t + FD>{}, fa[(t + FD) % FS]);
                const char* ft = slot + (t - t_lo) * TAPB;
#pragma unroll
                for (int s = 0; s < KS1; ++s) {
#pragma unroll
                    for (int j = 0; j < NB2; ++j) {
                        const u32x4 b = lds16(ft + f_row[j] + (((2 * s + h) ^ f_sw[j]) << 4));
                        acc0[j] = mfma16<T>(fa[t % FS][s], b, acc0[j]);
                    }
                }
            }, std::make_integer_sequence<int, t_hi - t_lo>{});
        };
        ch_for_each(pre, std::make_integer_sequence<int, NPRE>{});
        // ---- its epilogue: BatchNorm + ReLU, one rounding, and the transposition from accumulators (lane = column) to A
        //      fragments (lane = row), 64 columns at a time through the wave's staging block.  Rows of 256 bytes would put the
        //      32 lanes of a fragment read on the same banks: the 16-byte pieces of row r are permuted by r & 15 ----
        const float* t0 = tab + 4 * N1 + 2 * N2;
#pragma unroll
        for (int jj = 0; jj < NB2 / 2; ++jj) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_h + (r & 3) + 8 * (r >> 2), col = j * 32 + col_l;
                    *reinterpret_cast<float*>(stage + row * 256 + ((((col >> 2) ^ row) & 15) << 4) + (col & 3) * 4) = acc0[2 * jj + j][r];
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int col = s4 * 16 + h * 8;                   // this lane's 8 channels of the 64
                const f32x4 lo = lds16f(stage + r32 * 256 + ((((col >> 2)) ^ r32) & 15) * 16);
                const f32x4 hi = lds16f(stage + r32 * 256 + ((((col >> 2) + 1) ^ r32) & 15) * 16);
                const float* tt = t0 + jj * 64 + col;
                const f32x4 s_lo = *reinterpret_cast<const f32x4*>(tt), s_hi = *reinterpret_cast<const f32x4*>(tt + 4);
                const f32x4 h_lo = *reinterpret_cast<const f32x4*>(tt + N2), h_hi = *reinterpret_cast<const f32x4*>(tt + N2 + 4);
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const float sc[8] = {s_lo[0], s_lo[1], s_lo[2], s_lo[3], s_hi[0], s_hi[1], s_hi[2], s_hi[3]};
                const float sh[8] = {h_lo[0], h_lo[1], h_lo[2], h_lo[3], h_hi[0], h_hi[1], h_hi[2], h_hi[3]};
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = pack2<T>(v[2 * j], v[2 * j + 1]);
                xa[jj * 4 + s4] = o;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (!EARLY_RV) load_rv();
    f32x16 acc2[NB2];
#pragma unroll
    for (int j = 0; j < NB2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[j][q] = 0.f;
    // B fragment addresses inside a slot (lane: column lane & 31, k half h)
    int b1_row[2], b1_sw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { b1_row[j] = (j * 32 + r32) * RBW1; b1_sw[j] = G::swz1(j * 32 + r32); }
    int b2_off[NB2];
#pragma unroll
    for (int j = 0; j < NB2; ++j) b2_off[j] = W1C + (j * 32 + r32) * 128;
    const int sw128 = G::swz128(r32);                              // (swz128(j * 32 + r32) == swz128(r32))
    const int z_rd = r32 * 128;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the constants table is written

    // The chunk loop is FULLY UNROLLED (c is a compile-time constant): ring slots and table rows become immediate offsets,
    // and the compiler's wait-count pass sees straight-line code and counts the register loads in flight exactly (in a
    // rolled loop the shortcut prefetch is loop-carried and every use got s_waitcnt vmcnt(0)).
    auto chunk = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        constexpr int SET = RVS == 2 ? (c & 1) : 0;
        // chunk c of both filters has landed: this wave issued its DMAs at the top of chunk c - 1; behind them came 4 stores
        // and (where there still was a chunk to prefetch) 4 shortcut loads — everything older is complete.  Then every wave
        // is done with chunk c - 1, whose slot takes chunk c + 1.
        // (FRONT: chunk 0's DMAs went out at the top of the last pre-chunk, in front of that pre-chunk's tap loads)
        // (TAIL: no x fragments, no stores in the loop)
        if constexpr (c == 0) ch_wait_vm<(TAIL ? 0 : FRONT ? G::front_loads((NPRE - 1) * TPS, 9) : KS1) + (EARLY_RV ? 0 : RVL * RVS)>();
        else ch_wait_vm<(TAIL ? 0 : 4) + ((c - 1) + RVS < NCH ? RVL : 0)>();
        __builtin_amdgcn_s_barrier();
        if constexpr (c + 1 < NCH) issue_chunk(c + 1, (NPRE + c + 1) % NR);
        const char* slot = smem + ((NPRE + c) % NR) * SLOT;
        if constexpr (!TAIL) {
        // ---- GEMM 1: 32 rows x 64 columns, K1 deep ----
            f32x16 acc1[2];
    #pragma unroll
            for (int j = 0; j < 2; ++j)
    #pragma unroll
                for (int q = 0; q < 16; ++q) acc1[j][q] = 0.f;
    #pragma unroll
            for (int s = 0; s < KS1; ++s) {
    #pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 b = lds16(slot + b1_row[j] + (((2 * s + h) ^ b1_sw[j]) << 4));
                    acc1[j] = mfma16<T>(xa[s], b, acc1[j]);
                }
            }
            // ---- its epilogue: transpose through the wave's staging block, + bias + shortcut, one rounding, store; the rounded
            //      values through the next unit's BatchNorm + ReLU into the wave's z tile ----
    #pragma unroll
            for (int j = 0; j < 2; ++j)
    #pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float*>(stage + woff + ((r & 3) + 8 * (r >> 2)) * 256 + j * 128) = acc1[j][r];
            __builtin_amdgcn_wave_barrier();
        }
        {
            const float* t = tab + c * 64 + c8 * 8;
            float sc[8], sh[8], ps[8], ph[8];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(t + 4 * hh);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(t + N1 + 4 * hh);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(t + 2 * N1 + 4 * hh);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(t + 3 * N1 + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[4 * hh + e] = v0[e]; sh[4 * hh + e] = v1[e]; ps[4 * hh + e] = v2[e]; ph[4 * hh + e] = v3[e]; }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                u32x4 o;
                if constexpr (TAIL) {
                    o = rv[SET][p];                                // the stored y itself
                } else {
                    const f32x4 lo = lds16f(stage + roff + p * 2048);
                    const f32x4 hi = lds16f(stage + roff + p * 2048 + 16);
                    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                    if constexpr (!NORES) {
                        const u32x4 rq = rv[SET][p];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[2 * j] += from_bits<T>((unsigned short)(rq[j] & 0xffffu));
                            v[2 * j + 1] += from_bits<T>((unsigned short)(rq[j] >> 16));
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = pack2<T>(v[2 * j], v[2 * j + 1]);
                    *reinterpret_cast<u32x4*>(yrow[p] + c * 64) = o;
                }
                // relu(x * pscale + pshift) of the STORED value, rounded once (gv_conv2d_fwd_xpre's loader)
                u32x4 zq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float zl = fmaxf(from_bits<T>((unsigned short)(o[j] & 0xffffu)) * ps[2 * j] + ph[2 * j], 0.f);
                    const float zh = fmaxf(from_bits<T>((unsigned short)(o[j] >> 16)) * ps[2 * j + 1] + ph[2 * j + 1], 0.f);
                    zq[j] = (unsigned)to_bits<T>(zl) | ((unsigned)to_bits<T>(zh) << 16);
                }
                const int zr = p * 8 + r8;
                *reinterpret_cast<u32x4*>(ztile + zr * 128 + ((c8 ^ G::swz128(zr)) << 4)) = zq;
            }
        }
        if constexpr (c + RVS < NCH && !NORES) {                   // this register set is free: the shortcut RVS chunks ahead
#pragma unroll
            for (int p = 0; p < 4; ++p) rv[SET][p] = *reinterpret_cast<const u32x4*>(rrow_p[p] + (c + RVS) * 64);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- GEMM 2, partial sum over this chunk's 64 channels ----
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4 za = lds16(ztile + z_rd + (((2 * s + h) ^ sw128) << 4));
#pragma unroll
            for (int j = 0; j < NB2; ++j) {
                const u32x4 b = lds16(slot + b2_off[j] + (((2 * s + h) ^ sw128) << 4));
                acc2[j] = mfma16<T>(za, b, acc2[j]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    ch_for_each(chunk, std::make_integer_sequence<int, NCH>{});
    // ---- epilogue of GEMM 2: BatchNorm + ReLU, one rounding, whole 128-byte lines ----
    const float* t2 = tab + 4 * N1;
#pragma unroll
    for (int jj = 0; jj < NB2 / 2; ++jj) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<float*>(stage + woff + ((r & 3) + 8 * (r >> 2)) * 256 + j * 128) = acc2[2 * jj + j][r];
        __builtin_amdgcn_wave_barrier();
        const int col = jj * 64 + c8 * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(t2 + col + 4 * hh);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(t2 + N2 + col + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[4 * hh + e] = v0[e]; sh[4 * hh + e] = v1[e]; }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x4 lo = lds16f(stage + roff + p * 2048);
            const f32x4 hi = lds16f(stage + roff + p * 2048 + 16);
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
            store_chunk_lean<T>(zrow[p] + jj * 64, v, a.relu2 != 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    ch_wait_vm<0>();                                               // (nothing of this workgroup may still be landing in LDS)
}

template <typename T, int D, int NW, int RVS, bool FRONT = false, bool TAIL = false, int KX = 1>
int launch_chain(const ChainArgs& a, hipStream_t st) {
    using G = ChainGeom<D, KX>;
    constexpr int lds = G::template lds_bytes<NW>() - (TAIL ? G::NR * G::W1C : 0);
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    auto kern = &conv_chain_lp<T, D, NW, RVS, FRONT, TAIL, KX>;
    if (lds > 64 * 1024) {
        const bool ok = GV_BIG_LDS_OK(kern, lds);
        if (!ok) return GV_E_UNSUPPORTED;
    }
    const int nwg = gv_ceil_div(a.M, NW * 32);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(NW * 64), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T>
int launch_unit_d(int d, const ChainArgs& a, hipStream_t st) {
    switch (d) {
        case 64: return launch_chain<T, 64, 4, 2, true>(a, st);
        case 128: return launch_chain<T, 128, 8, 1, true>(a, st);
    }
    return GV_E_UNSUPPORTED;
}

template <typename T>
int launch_chain_d(int d, const ChainArgs& a, hipStream_t st, bool proj = false) {
    if (proj) {                                                // K = 2d, no shortcut operand: 117 KB, one workgroup of 8 waves
        if (d == 64) return launch_chain<T, 64, 8, 1, false, false, 2>(a, st);
        return GV_E_UNSUPPORTED;
    }
    switch (d) {
        case 64: return launch_chain<T, 64, 4, 2>(a, st);      // 68.5 KB of LDS: two workgroups of 4 waves per CU
        case 128: return launch_chain<T, 128, 8, 1>(a, st);    // 137 KB: one workgroup of 8 waves
    }
    return GV_E_UNSUPPORTED;
}

template <typename T>
int launch_tail_d(int d, const ChainArgs& a, hipStream_t st) {
    switch (d) {
        case 64: return launch_chain<T, 64, 4, 2, false, true>(a, st);
        case 128: return launch_chain<T, 128, 8, 2, false, true>(a, st);
        case 256: return launch_chain<T, 256, 8, 2, false, true>(a, st);
    }
    return GV_E_UNSUPPORTED;
}

int g_chain_debug = 0;

}  // namespace

namespace gvconv {

// gv_conv2d_fwd_xpre's class "1x1, cin = 4 * cout, BatchNorm + ReLU on every column, one 16-bit destination" (the conv1 of a
// ResNet-v2 identity unit reading the unit input through its folded pre-activation, nets/resnet_v2.py:75,83-84) as the TAIL
// form of the bottleneck launch: a tile configuration of its own (the convolution entry point offers it under the special
// tile index), bit for bit the register-staged kernels' result.
bool chain_tail_ok(const ConvArgs& a) {
    return a.xscale != nullptr && a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 && a.cin == 4 * a.cout &&
           (a.cout == 64 || a.cout == 128 || a.cout == 256) && a.res == nullptr && a.y2 == nullptr && a.split == 0 && a.relu &&
           a.relu_limit >= a.cout && a.x_ld % 8 == 0 && a.y_ld % 8 == 0 && gv_aligned16(a.x) && gv_aligned16(a.y) &&
           gv_aligned16(a.w) && a.st.mode == STAT_OFF && a.y_step == 0 && a.dil_shift == 0 && a.pool == 0;
}

int chain_tail_launch(int dtype, const ConvArgs& a, hipStream_t st) {
    if (!chain_tail_ok(a)) return GV_E_UNSUPPORTED;
    ChainArgs c;
    c.x = nullptr; c.w1 = nullptr; c.sc1 = c.sh1 = nullptr; c.y = nullptr;
    c.res = reinterpret_cast<const unsigned short*>(a.x);          // the unit input, read through the prefetch registers
    c.psc = a.xscale; c.psh = a.xshift;
    c.w2 = reinterpret_cast<const unsigned short*>(a.w); c.sc2 = a.scale; c.sh2 = a.shift;
    c.z = reinterpret_cast<unsigned short*>(a.y);
    c.M = a.M; c.x_ld = 0; c.res_ld = a.x_ld; c.y_ld = 0; c.z_ld = a.y_ld;
    c.relu2 = 1;
    c.dbg = g_chain_debug;
    c.w0 = nullptr; c.sc0 = c.sh0 = nullptr; c.zeros = nullptr; c.ih = c.iw = 0;
    if (dtype == GV_BF16) return launch_tail_d<__bf16>(a.cout, c, st);
    if (dtype == GV_F16) return launch_tail_d<_Float16>(a.cout, c, st);
    return GV_E_UNSUPPORTED;
}

}  // namespace gvconv

extern "C" void gv_bottleneck_chain_set_debug(int bits) { g_chain_debug = bits; }

extern "C" int gv_bottleneck_chain_fwd(const gv_chain_desc* d, const void* x, const void* w3_packed, const float* scale3,
                                       const float* shift3, const void* shortcut, void* y, const float* pre_scale,
                                       const float* pre_shift, const void* w1_packed, const float* scale1, const float* shift1,
                                       void* z, void* stream) {
    const bool proj = d && (d->flags & GV_CHAIN_PROJ) != 0;   // the shortcut is the second half of GEMM 1's K: no operand
    if (!d || !x || !w3_packed || !scale3 || !shift3 || (!shortcut && !proj) || (shortcut && proj) || !y || !pre_scale ||
        !pre_shift || !w1_packed || !scale1 || !shift1 || !z)
        return GV_E_BADARG;
    if (d->m <= 0 || d->d <= 0 || d->x_ld < d->d * (proj ? 2 : 1) || (!proj && d->res_ld < 4 * d->d) || d->y_ld < 4 * d->d ||
        d->z_ld < d->d)
        return GV_E_BADARG;
    if (d->dtype != GV_BF16 && d->dtype != GV_F16) return GV_E_UNSUPPORTED;
    if (d->d != 64 && (d->d != 128 || proj)) return GV_E_UNSUPPORTED;
    if ((d->x_ld | (proj ? 0 : d->res_ld) | d->y_ld | d->z_ld) % 8 != 0) return GV_E_UNSUPPORTED;
    if (!gv_aligned16(x) || !gv_aligned16(w3_packed) || !gv_aligned16(shortcut) || !gv_aligned16(y) || !gv_aligned16(w1_packed) ||
        !gv_aligned16(z))
        return GV_E_ALIGN;
    // 32-bit row offsets inside the kernel's size_t arithmetic are fine; the row count itself must fit an int
    ChainArgs a;
    a.x = (const unsigned short*)x; a.w1 = (const unsigned short*)w3_packed; a.sc1 = scale3; a.sh1 = shift3;
    a.res = (const unsigned short*)shortcut; a.y = (unsigned short*)y; a.psc = pre_scale; a.psh = pre_shift;
    a.w2 = (const unsigned short*)w1_packed; a.sc2 = scale1; a.sh2 = shift1; a.z = (unsigned short*)z;
    a.M = d->m; a.x_ld = d->x_ld; a.res_ld = d->res_ld; a.y_ld = d->y_ld; a.z_ld = d->z_ld;
    a.relu2 = (d->flags & GV_CONV_RELU2) ? 1 : 0;
    a.dbg = g_chain_debug;
    a.w0 = nullptr; a.sc0 = a.sh0 = nullptr; a.zeros = nullptr; a.ih = a.iw = 0;
    if (d->dtype == GV_BF16) return launch_chain_d<__bf16>(d->d, a, (hipStream_t)stream, proj);
    return launch_chain_d<_Float16>(d->d, a, (hipStream_t)stream, proj);
}

extern "C" int gv_bottleneck_unit_fwd(const gv_unit_desc* d, const void* x, const void* w2_packed, const float* scale2,
                                      const float* shift2, const void* w3_packed, const float* scale3, const float* shift3,
                                      const void* shortcut, void* y, const float* pre_scale, const float* pre_shift,
                                      const void* w1_packed, const float* scale1, const float* shift1, void* z, void* stream) {
    if (!d || !x || !w2_packed || !scale2 || !shift2 || !w3_packed || !scale3 || !shift3 || !shortcut || !y || !pre_scale ||
        !pre_shift || !w1_packed || !scale1 || !shift1 || !z)
        return GV_E_BADARG;
    if (d->nb <= 0 || d->ih <= 0 || d->iw <= 0 || d->d <= 0 || d->x_ld < d->d || d->res_ld < 4 * d->d || d->y_ld < 4 * d->d ||
        d->z_ld < d->d)
        return GV_E_BADARG;
    const int64_t M64 = (int64_t)d->nb * d->ih * d->iw;
    if (M64 > 0x7fffffff) return GV_E_UNSUPPORTED;
    if (d->dtype != GV_BF16 && d->dtype != GV_F16) return GV_E_UNSUPPORTED;
    if (d->d != 64 && d->d != 128) return GV_E_UNSUPPORTED;
    if ((d->x_ld | d->res_ld | d->y_ld | d->z_ld) % 8 != 0) return GV_E_UNSUPPORTED;
    if (!gv_aligned16(x) || !gv_aligned16(w2_packed) || !gv_aligned16(w3_packed) || !gv_aligned16(shortcut) || !gv_aligned16(y) ||
        !gv_aligned16(w1_packed) || !gv_aligned16(z))
        return GV_E_ALIGN;
    ChainArgs a;
    a.x = (const unsigned short*)x; a.w1 = (const unsigned short*)w3_packed; a.sc1 = scale3; a.sh1 = shift3;
    a.res = (const unsigned short*)shortcut; a.y = (unsigned short*)y; a.psc = pre_scale; a.psh = pre_shift;
    a.w2 = (const unsigned short*)w1_packed; a.sc2 = scale1; a.sh2 = shift1; a.z = (unsigned short*)z;
    a.M = (int)M64; a.x_ld = d->x_ld; a.res_ld = d->res_ld; a.y_ld = d->y_ld; a.z_ld = d->z_ld;
    a.relu2 = (d->flags & GV_CONV_RELU2) ? 1 : 0;
    a.dbg = g_chain_debug;
    a.w0 = (const unsigned short*)w2_packed; a.sc0 = scale2; a.sh0 = shift2;
    a.zeros = (const char*)gvconv::dma_zero_page();
    if (!a.zeros) return GV_E_UNSUPPORTED;
    a.ih = d->ih; a.iw = d->iw;
    a.div_img = gv_fast_div(d->ih * d->iw);
    a.div_row = gv_fast_div(d->iw);
    if (d->dtype == GV_BF16) return launch_unit_d<__bf16>(d->d, a, (hipStream_t)stream);
    return launch_unit_d<_Float16>(d->d, a, (hipStream_t)stream);
}
