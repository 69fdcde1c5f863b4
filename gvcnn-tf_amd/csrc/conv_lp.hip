// conv_lp.hip — implicit-GEMM convolution on 16-bit storage (GV_BF16 / GV_F16 tensors, fp32 accumulate):
// the path of configs c3-c5.  Activations, residual, outputs and the packed filter are
// 16-bit in HBM; products run on v_mfma_f32_32x32x16_{bf16,f16}; the BN fold (scale/shift), residual
// add and ReLU are applied to the fp32 accumulator before the single rounding to the storage type.
//
// Structure (4 or 8 waves, wave tile TM x TN MFMA tiles of 32x32, k-tile = 32):
//   * LDS row = 64 B (32 values) + 16 B pad = 5 sixteen-byte slots (odd): the ds_read_b128 fragment reads
//     (lane (r = lane&31, h = lane>>5) reads k = 16s+8h .. +7 of row r) hit 16 distinct 4-bank groups in
//     every 16-lane service group;
//   * a loader lane owns one 16-byte chunk (8 channels of one filter tap) per slot; its element offset
//     advances by 32 channels per k-tile and is re-derived from (r, s) only when the chunk moves to another
//     tap, loads are unconditional (padding taps read offset 0 and are zeroed by a select before the LDS
//     write), so the compiler counts vmcnt and keeps two k-tiles of global loads in flight; rows R and
//     R+4 share an 8-lane ds_write_b128 group (conflict-free for 80-byte rows);
//   * fragments are double buffered in registers; the k-tile barrier sits between the two halves of a
//     tile's MFMAs;
//   * GENERIC variant (cin % 8 != 0: the stems' cin = 3) gathers element-wise, optionally straight from the
//     fp32 network input (GV_CONV_X_F32), so no separate cast pass over the images exists.
#include <type_traits>

#include "conv_common.h"
#include "conv_lp_epi.h"

namespace {

constexpr int KT = 32;                      // k-tile depth
constexpr int RB = 2 * KT + 16;             // LDS row bytes

// FDB: fragments double buffered in registers.  The 128x128 tile (2x2 accumulators per wave) reaches three workgroups
// per CU only with ONE fragment set; the other two waves of the SIMD cover the ds_read latency instead.
template <int WM, int WN, int TM, int TN>
constexpr bool lp_fdb() { return !(WM * WN == 4 && TM * TN == 4); }

// XPRE: the input is read as relu(x * xscale[c] + xshift[c]) (fp32 arithmetic on the staged registers, rounded once
// more to T): the pre-activation of a ResNet-v2 unit (nets/resnet_v2.py:75) applied by its CONSUMER, so that the unit
// before it stores the sum once instead of the sum and its pre-activation.  1x1 / unpadded launches only.
// STATS: train-mode BatchNorm sums of the stored tensor folded into the epilogue (conv_stats.h).
template <typename T, int WM, int WN, int TM, int TN, bool GENERIC, bool XF32, bool XPRE = false, int STATS = 0>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? (TM * TN <= 4 && STATS != gvconv::STAT_BWD ? 3 : 2) : 1)) void conv_igemm_lp(const ConvArgs a) {
    constexpr bool FDB = lp_fdb<WM, WN, TM, TN>();
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    static_assert(GENERIC || !XF32, "fp32 input only on the gather path");
    static_assert(!XPRE || !GENERIC, "pre-activation on load: vector loader only");
    constexpr int NT = WM * WN * 64;                 // threads
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int A_SLOTS = (BM * 4 + NT - 1) / NT;  // (row, chunk) slots: 8 values each
    constexpr int B_SLOTS = (BN * 4 + NT - 1) / NT;
    constexpr int NMF = TM * TN * 2;                 // MFMAs per k-tile
    constexpr int HALF = NMF / 2;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* sA = smem_raw;                             // [2][BM][RB]
    char* sB = smem_raw + 2 * BM * RB;               // [2][BN][RB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;

    const int lid = gv_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % a.tiles_n;
    const int tile_m = lid / a.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    constexpr bool HAS_SUMS = gvconv::stat_has(STATS);
    int st_b0 = 0;                                   // STATS: image of the tile's first pixel
    if constexpr (HAS_SUMS) st_b0 = m0 / a.st.hw;

    const int q = tid & 3;                           // this thread's 16-byte chunk of every row it loads
    // slot -> row: an 8-lane ds_write_b128 group covers rows R and R+4 (80-byte rows: 16 banks apart mod 32)
    auto slot_row = [](int idx) -> int {
        const int g8 = idx >> 3;
        return (g8 >> 2) * 8 + (g8 & 3) + 4 * ((idx >> 2) & 1);
    };
    const unsigned short* xs = reinterpret_cast<const unsigned short*>(a.x);

    int a_img[A_SLOTS], a_iy0[A_SLOTS], a_ix0[A_SLOTS];
    const int ohow = a.oh * a.ow;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int row = slot_row(tid + i * NT);
        const int m = m0 + row;
        if (row < BM && m < a.M) {
            const int n = gv_div(m, a.y_div_img);
            const int rem = m - n * ohow;
            const int oy = gv_div(rem, a.y_div_row);
            const int ox = rem - oy * a.ow;
            a_img[i] = n * a.ih;
            a_iy0[i] = oy * a.stride - a.pad_t;
            a_ix0[i] = ox * a.stride - a.pad_l;
        } else {
            a_img[i] = 0;
            a_iy0[i] = -(1 << 28);
            a_ix0[i] = 0;
        }
    }
    // gather path: k -> (offset of tap (r, s, c) from the window origin, r, s) looked up in LDS instead of two
    // integer divisions per element; entries past K carry r = 0x7fff (never inside the image)
    int2* ktab = reinterpret_cast<int2*>(smem_raw + (2 * BM + 2 * BN) * RB);
    int a_base[A_SLOTS];
    if constexpr (GENERIC) {
        for (int k = tid; k < a.Kpad; k += NT) {
            int2 t;
            if (k < a.K) {
                const int rs = k / a.cin, c = k - rs * a.cin;
                const int r = rs / a.kw, s_ = rs - r * a.kw;
                t.x = (r * a.iw + s_) * a.x_ld + c;
                t.y = (r << 16) | s_;
            } else {
                t.x = 0;
                t.y = 0x7fff << 16;
            }
            ktab[k] = t;
        }
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i)
            a_base[i] = a_iy0[i] > -(1 << 27) ? ((a_img[i] + a_iy0[i]) * a.iw + a_ix0[i]) * a.x_ld : 0;
        __syncthreads();
    }
    // The epilogue's per-column constants go to LDS behind everything else (main buffers + gather / pre-activation table,
    // or the per-wave staging blocks where those are larger): requested here, written after the first tile's loads are
    // issued, read back as 16-byte LDS reads (see conv_dma.hip: loaded from global memory inside the epilogue, every
    // read-back block waits out a memory round trip, which the short-K launches cannot hide)
    static_assert(NT >= BN, "one thread per tile column");
    float* sstab;
    {
        const int main_b = (2 * BM + 2 * BN) * RB + (GENERIC ? a.Kpad * 8 : 0) + (XPRE ? a.cin * 8 : 0);
        const int epi_b = WM * WN * EpiGeom<TN>::BYTES;
        sstab = reinterpret_cast<float*>(smem_raw + ((main_b > epi_b ? main_b : epi_b) + 15) / 16 * 16);
    }
    float ss_v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool ss_dual = a.y2 != nullptr && a.split == 0;
    if (tid < BN) {
        const int cc = min(n0 + tid, a.cout - 1);
        ss_v[0] = a.scale[cc];
        ss_v[1] = a.shift[cc];
        if (ss_dual) { ss_v[2] = a.scale2[cc]; ss_v[3] = a.shift2[cc]; }
    }
    // XPRE: (scale, shift) of every input channel, read back 8 channels (this thread's chunk) at a time
    float2* xss = reinterpret_cast<float2*>(smem_raw + (2 * BM + 2 * BN) * RB);
    if constexpr (XPRE) {
        for (int c = tid; c < a.cin; c += NT) xss[c] = make_float2(a.xscale[c], a.xshift[c]);
        __syncthreads();
    }
    int xc[2] = {0, 0};                               // first channel of the chunk staged in register set 0 / 1
    const char* b_ptr[B_SLOTS];
    bool b_ok[B_SLOTS];
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
        const int row = slot_row(tid + i * NT);
        const int n = n0 + row;
        b_ok[i] = (row < BN);
        // rows past cout re-read the last filter: their accumulator columns are never stored
        const int nc = n < a.cout ? n : a.cout - 1;
        b_ptr[i] = (const char*)a.w + ((size_t)nc * a.Kpad + 8 * q) * 2;
    }

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    u32x4 ra[2][A_SLOTS];                             // tile t is staged in register set t & 1
    u32x4 rb[2][B_SLOTS];
    bool ra_ok[2][A_SLOTS];
    // filter tap / channel of this thread's chunk in the NEXT tile to load
    int fc = 8 * q, fs = 0, fr = 0;
    unsigned a_off[A_SLOTS];                          // element offset of the chunk to load (0 when the tap is padding)
    bool a_ok[A_SLOTS];
    auto locate = [&]() {                             // (fr, fs, fc) -> per-slot offset / validity
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int iyn = a_iy0[i] + fr;
            const int ixn = a_ix0[i] + fs;
            const int dmask = (1 << a.dil_shift) - 1; // zero-dilated input: see conv_bf16s.hip
            const int iy = iyn >> a.dil_shift;
            const int ix = ixn >> a.dil_shift;
            const bool ok = iyn >= 0 && ixn >= 0 && ((iyn | ixn) & dmask) == 0 && iy < a.ih && ix < a.iw && fr < a.kh;
            a_ok[i] = ok;
            a_off[i] = ok ? ((unsigned)(a_img[i] + iy) * (unsigned)a.iw + (unsigned)ix) * (unsigned)a.x_ld + (unsigned)fc : 0u;
        }
    };
    if constexpr (!GENERIC) {
        while (fc >= a.cin) { fc -= a.cin; if (++fs == a.kw) { fs = 0; ++fr; } }
        locate();
    }

    auto load_tile = [&](auto rsc, int kt) {
        constexpr int RS = decltype(rsc)::value;
        if constexpr (XPRE) xc[RS] = min(fc, a.cin - 8);       // (a chunk past the last channel is zeroed by its mask)
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if constexpr (!GENERIC) {
                // unconditional load (padding taps read offset 0.. of the tensor and are zeroed by the select)
                v = *reinterpret_cast<const u32x4*>(xs + a_off[i]);
                ra_ok[RS][i] = a_ok[i];
            } else {
                ra_ok[RS][i] = true;
                unsigned short e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kt < a.ktiles)                            // (uniform) no gather work for prefetches past the end
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kt * KT + 8 * q + j;
                    const int2 t = ktab[k];                   // same k across a row group: LDS broadcast
                    const int iy = a_iy0[i] + (t.y >> 16);
                    const int ix = a_ix0[i] + (t.y & 0xffff);
                    e[j] = 0;
                    if ((unsigned)iy < (unsigned)a.ih && (unsigned)ix < (unsigned)a.iw) {
                        const int off = a_base[i] + t.x;
                        if constexpr (XF32) e[j] = to_bits<T>(a.x[off]);
                        else e[j] = xs[off];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (unsigned)e[2 * j] | ((unsigned)e[2 * j + 1] << 16);
            }
            ra[RS][i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int ktc = kt < a.ktiles ? kt : a.ktiles - 1;          // prefetch past the end re-reads the last tile
            rb[RS][i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + (size_t)ktc * (KT * 2));
        }
    };
    auto advance_tap = [&]() {
        if constexpr (!GENERIC) {
            fc += KT;
            if (fc >= a.cin) {                        // the chunk leaves this filter tap: new window position
                do { fc -= a.cin; if (++fs == a.kw) { fs = 0; ++fr; } } while (fc >= a.cin);
                locate();
            } else {                                  // same tap, next 32 channels
#pragma unroll
                for (int i = 0; i < A_SLOTS; ++i) a_off[i] += a_ok[i] ? (unsigned)KT : 0u;
            }
        }
    };
    auto store_tile = [&](auto rsc, int buf) {
        constexpr int RS = decltype(rsc)::value;
        f32x4 xp[4];                                      // XPRE: (scale, shift) x 8 channels, the same for every slot
        if constexpr (XPRE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xp[j] = reinterpret_cast<const f32x4*>(xss + xc[RS])[j];
        }
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int idx = tid + i * NT;
            if (A_SLOTS * NT == BM * 4 || idx < BM * 4) {
                u32x4 v = ra[RS][i];
                if constexpr (XPRE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 p = xp[j];                    // scale, shift of channels 2j and 2j + 1
                        const float lo = fmaxf(from_bits<T>((unsigned short)(v[j] & 0xffffu)) * p[0] + p[1], 0.f);
                        const float hi = fmaxf(from_bits<T>((unsigned short)(v[j] >> 16)) * p[2] + p[3], 0.f);
                        v[j] = (unsigned)to_bits<T>(lo) | ((unsigned)to_bits<T>(hi) << 16);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = ra_ok[RS][i] ? v[j] : 0u;
                *reinterpret_cast<u32x4*>(sA + buf * BM * RB + slot_row(idx) * RB + 16 * q) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int idx = tid + i * NT;
            if (B_SLOTS * NT == BN * 4 || b_ok[i])
                *reinterpret_cast<u32x4*>(sB + buf * BN * RB + slot_row(idx) * RB + 16 * q) = rb[RS][i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * RB + 16 * (lane >> 5);
    const char* a_frag = sA + (wm * TM * 32) * RB + frag_off;
    const char* b_frag = sB + (wn * TN * 32) * RB + frag_off;

    u32x4 fa[FDB ? 2 : 1][TM][2], fb[FDB ? 2 : 1][TN][2];
    auto read_frags = [&](auto setc, int buf) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[S][i][s] = *reinterpret_cast<const u32x4*>(a_frag + buf * BM * RB + i * 32 * RB + s * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[S][j][s] = *reinterpret_cast<const u32x4*>(b_frag + buf * BN * RB + j * 32 * RB + s * 32);
        }
    };
    auto mfma_range = [&](auto setc, auto loc, auto hic) {
        constexpr int S = decltype(setc)::value;
        constexpr int LO = decltype(loc)::value, HI = decltype(hic)::value;
#pragma unroll
        for (int t = LO; t < HI; ++t) {
            const int s = t / (TM * TN);
            const int ij = t % (TM * TN);
            const int i = ij / TN, j = ij % TN;
            acc[i][j] = mfma16<T>(fa[S][i][s], fb[S][j][s], acc[i][j]);
        }
    };
    using IH = std::integral_constant<int, HALF>;
    using IN = std::integral_constant<int, NMF>;

    auto step = [&](auto setc, int kt) {
        constexpr int S = decltype(setc)::value;
        using IS = std::integral_constant<int, S>;
        using IO = std::integral_constant<int, S ^ 1>;
        const int buf = kt & 1;
        mfma_range(IS{}, I0{}, IH{});
        store_tile(IO{}, buf ^ 1);                             // register set S^1 holds tile kt+1
        advance_tap();
        load_tile(IO{}, kt + 3);                               // refill it; tile kt+2 stays in flight in set S
        __syncthreads();
        read_frags(IO{}, buf ^ 1);
        mfma_range(IS{}, IH{}, IN{});
    };

    load_tile(I0{}, 0);
    store_tile(I0{}, 0);
    advance_tap();
    load_tile(I1{}, 1);
    advance_tap();
    load_tile(I0{}, 2);
    if (tid < BN) {
        sstab[tid] = ss_v[0];
        sstab[BN + tid] = ss_v[1];
        if (ss_dual) { sstab[2 * BN + tid] = ss_v[2]; sstab[3 * BN + tid] = ss_v[3]; }
    }
    __syncthreads();
    read_frags(I0{}, 0);
    int kt = 0;
    if constexpr (FDB) {
        for (; kt + 2 < a.ktiles; kt += 2) {
            step(I0{}, kt);
            step(I1{}, kt + 1);
        }
        if (a.ktiles - kt == 2) {
            step(I0{}, kt);
            mfma_range(I1{}, I0{}, IN{});
        } else {
            mfma_range(I0{}, I0{}, IN{});
        }
    } else {
        // one fragment set: hand tile kt+1 to LDS, multiply tile kt, barrier, fetch tile kt+1's fragments
        auto step1 = [&](auto rsc, int k) {
            const int buf = k & 1;
            store_tile(rsc, buf ^ 1);
            advance_tap();
            load_tile(rsc, k + 3);
            mfma_range(I0{}, I0{}, IN{});
            __syncthreads();
            read_frags(I0{}, buf ^ 1);
        };
        for (; kt + 2 < a.ktiles; kt += 2) {
            step1(I1{}, kt);
            step1(I0{}, kt + 1);
        }
        if (a.ktiles - kt == 2) step1(I1{}, kt);
        mfma_range(I0{}, I0{}, IN{});
    }

    __syncthreads();                                  // every wave is done with the main-loop buffers
    if constexpr (HAS_SUMS) {                         // the sums table (in the freed main-loop space where it fits)
        if (!(a.st.dbg & 16384)) {
        gvconv::stat_table_init<STATS>(a.st, smem_raw, tid, NT, BN, n0, a.cout, st_b0);
        __syncthreads();
        }
    }
    lp_epilogue_staged<T, TM, TN, STATS>(a, acc, m0, n0, wm, wn, lane,
                                         reinterpret_cast<float*>(smem_raw + wave * EpiGeom<TN>::BYTES), 32,
                                         (a.dbg & 512) ? nullptr : sstab, BN,      // dbg 512: constants from global memory (A/B)
                                         smem_raw, st_b0 * (HAS_SUMS ? a.st.hw : 0));
    if constexpr (HAS_SUMS) {
        if (!(a.st.dbg & 16384)) __syncthreads();     // every lane's runs are in the table
        const int last = (m0 + BM < a.M ? m0 + BM : a.M) - 1;
        gvconv::stat_publish<STATS>(a.st, smem_raw, tid, BN, n0, a.cout, st_b0, min(last / a.st.hw - st_b0 + 1, a.st.slots));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Halo-tiled 3x3 / stride-1 convolution for the few-channel stem layers (cin = 32, cout = 32 or 64:
// Conv2d_2a_3x3 / Conv2d_2b_3x3).  The implicit-GEMM kernel above fetches every input element once per filter tap
// (9x) from L2, which — not HBM, not the matrix pipe — bounds these layers.  Here a workgroup owns a 32-pixel-wide
// column strip of one image and walks down it 4 output rows at a time: the (4+2) x (32+2) pixel halo is staged ONCE
// in LDS (80-byte pixels: the 32 consecutive pixels a wave's MFMA rows map to are conflict-free), every tap's A
// fragment is a ds_read_b128 at a shifted pixel, and the whole filter (9 taps x 2 k-steps x TN fragments) lives in
// REGISTERS for the life of the workgroup.  The next tile's halo is loaded into registers under this tile's
// MFMAs.  Epilogue: the staged 16-byte path above.
// RPW: output rows per wave and tile (wave w owns rows w and w + 4): with two, the two barriers, the halo hand-over and the
// filter fragment reads of a tile are shared by twice the MFMAs and the halo overlap drops from 6/4 to 10/8 input rows
// per output row.
// CIN: 32 (Conv2d_2a / 2b forward) or 64 (the data gradient of Conv2d_2b: 64 -> 32 channels over the padded map; filter of
// 9 x 64 values per output channel in LDS).
// POOL (GV_CONV_MAXPOOL3S2; Conv2d_2b_3x3 -> MaxPool_3a_3x3, nets/inception_v3.py:111-113): the 3x3 / 2 VALID max pool of the
// output is taken inside the workgroup and only the pooled tensor is written (a quarter of the bytes; the pool's launch,
// which re-read all of them, disappears).  A strip advances 30 columns (15 pooled ones; the same number of strips as
// 32-column strips for the 109- and 147-wide maps).  After the scale / shift / ReLU each wave reduces its rows
// horizontally on the way out of its staging block (pooled pixel p = max over pixels 2p .. 2p+2; rounding to the storage
// type is monotone, so max-then-round equals the pool of the rounded tensor bit for bit) into an LDS image of the tile's
// 8 rows; after a barrier all threads take the vertical maxima on the packed words (values are >= 0 or -0.0: signed
// 16-bit max orders them as the floats) — pooled rows 4k .. 4k+2 of tile k at once, row 4k+3 when the first row of tile
// k+1 exists (the partial maximum waits in a register of the thread that owns the chunk).
template <typename T, int TN, int RPW, int CIN = 32, int STATS = 0, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_lp(const ConvArgs a) {
    static_assert(!POOL || (TN == 2 && RPW == 2 && CIN == 32 && STATS == 0), "the pooled form: Conv2d_2b's class");
    constexpr int TH = 4 * RPW, TW = 32, HH = TH + 2, HW = TW + 2, PB = CIN * 2 + 16;   // halo pixel + 16 B pad
    constexpr int PSTEP = 30, PPX = 15, XROW = 2 * PPX * 32 * 2;             // POOL: columns per strip, pooled pixels, bytes
                                                                              // of one reduced row [column tile][pixel][32]
    constexpr int CPP = CIN / 8, KS = CIN / 16;                               // 16-byte chunks / MFMA k-steps per pixel
    constexpr int NCH = HH * HW * CPP;                                        // 16-byte chunks of one halo
    constexpr int SL = (NCH + 255) / 256;
    constexpr bool BREG = TN == 1 && CIN == 32;                               // filter fragments in registers (TN = 2: 144 VGPRs
                                                                              // of filter spill and run Conv2d_2b 0.295 -> 0.45 ms)
    constexpr int WB = 9 * CIN * 2 + 16;                                      // LDS filter row: K = 9 * CIN values + pad
    constexpr int SW = 32 + 4;                                                // staging row: one 32-column tile (+ pad)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // CIN = 64: the per-wave staging blocks ALIAS the halo (one more barrier per tile, between the MFMAs and the epilogue):
    // 67 instead of 85 KB, i.e. two resident workgroups
    constexpr bool ALIAS = CIN == 64 || (TN == 2 && RPW == 2);               // (64 columns, two rows per wave: 65 instead of 84 KB)
    static_assert(!ALIAS || 4 * 32 * SW * 4 <= HH * HW * PB, "staging fits the halo");
    char* sH = smem_raw;                                                      // [HH*HW][PB]
    float* stage = reinterpret_cast<float*>(smem_raw + (ALIAS ? 0 : HH * HW * PB)) + (threadIdx.x >> 6) * (32 * SW);
    char* sW = smem_raw + HH * HW * PB + (ALIAS ? 0 : 4 * 32 * SW * 4);       // [32*TN][WB]   (!BREG)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tiles_x = POOL ? (a.pw + PPX - 1) / PPX : (a.ow + TW - 1) / TW;
    const int n = blockIdx.x / tiles_x;
    const int ox0 = (blockIdx.x % tiles_x) * (POOL ? PSTEP : TW);
    const int oh_run = POOL ? 2 * a.ph + 1 : a.oh;                            // output rows somebody needs
    char* sX = sW + 32 * TN * WB;                                             // [8][XROW]   (POOL)
    u32x4 carry = {0u, 0u, 0u, 0u};
    const unsigned short* xs = reinterpret_cast<const unsigned short*>(a.x);
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(a.w);
    const unsigned short* res = reinterpret_cast<const unsigned short*>(a.res);
    unsigned short* y = reinterpret_cast<unsigned short*>(a.y);

    // the filter: B fragment of tap t, k-step c, column tile j = 8 values k = t*32 + c*16 + 8*lh .. of row n = j*32 + li
    u32x4 fb[BREG ? 9 : 1][KS][TN];
    if constexpr (BREG) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < KS; ++c)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = min(j * 32 + li, a.cout - 1);
                    fb[t][c][j] = *reinterpret_cast<const u32x4*>(wp + (size_t)col * a.Kpad + t * CIN + c * 16 + 8 * lh);
                }
    } else {
        constexpr int FCH = 9 * CPP;                                          // 16-byte chunks per filter row
        for (int idx = tid; idx < 32 * TN * FCH; idx += 256) {
            const int row = idx / FCH, ch = idx - row * FCH;
            const int col = min(row, a.cout - 1);
            *reinterpret_cast<u32x4*>(sW + row * WB + ch * 16) =
                *reinterpret_cast<const u32x4*>(wp + (size_t)col * a.Kpad + ch * 8);
        }
    }

    // epilogue constants of this lane's 8 output channels of column tile j (read-back layout: 4 lanes per 32 columns)
    constexpr int RPP = 16;
    const int rrow = lane >> 2, col8 = (lane & 3) * 8;
    float sc[TN][8], sh[TN][8];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = min(j * 32 + col8 + e, a.cout - 1);
            sc[j][e] = a.scale[c];
            sh[j][e] = a.shift[c];
        }
    const bool vec = (a.y_ld % 8 == 0) && ((((uintptr_t)y) & 15) == 0) &&
                     (res == nullptr || ((a.res_ld % 8 == 0) && ((((uintptr_t)res) & 15) == 0)));

    // STATS: the train-mode BatchNorm sums of the tensor this launch stores (conv_stats.h): the whole strip is one image
    gvconv::StatStrip<STATS == 0 ? gvconv::STAT_FWD : STATS, TN> sstat;
    const int sgrp = STATS != 0 ? n % a.st.G : 0;
    if constexpr (STATS != 0) sstat.init(a.st, sgrp, col8, a.cout);

    u32x4 hr[SL];
    auto fetch = [&](int oy0) {
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            const int idx = tid + k * 256;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (idx < NCH) {
                const int pix = idx / CPP, ch = idx % CPP;
                const int hy = pix / HW, hx = pix - hy * HW;
                const int iy = oy0 + hy - a.pad_t, ix = ox0 + hx - a.pad_l;
                if ((unsigned)iy < (unsigned)a.ih && (unsigned)ix < (unsigned)a.iw)
                    v = *reinterpret_cast<const u32x4*>(xs + ((size_t)(n * a.ih + iy) * a.iw + ix) * a.x_ld + ch * 8);
            }
            hr[k] = v;
        }
    };
    fetch(0);
    for (int oy0 = 0; oy0 < oh_run; oy0 += TH) {
        __syncthreads();                                   // previous tile: fragment reads and staging done
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            const int idx = tid + k * 256;
            if (idx < NCH) *reinterpret_cast<u32x4*>(sH + (idx / CPP) * PB + (idx % CPP) * 16) = hr[k];
        }
        __syncthreads();
        if (oy0 + TH < oh_run) fetch(oy0 + TH);            // in flight under the MFMAs below
        // STAT_BWD: the z chunks this tile's epilogue will need, requested here so that they too travel under the MFMAs
        // (fetched inside the epilogue every row-chunk waited out a memory round trip: Conv2d_2a's data gradient 0.145 ->
        // 0.29 ms instead of -> 0.19)
        u32x4 zpre[STATS == gvconv::STAT_BWD ? RPW * TN : 1][2];
        if constexpr (STATS == gvconv::STAT_BWD) {
#pragma unroll
            for (int qj = 0; qj < RPW * TN; ++qj) {
                const int q = qj / TN, j = qj % TN;
                const int oy = oy0 + wave + 4 * q;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = pass * 16 + rrow;
                    const bool ok = oy < a.oh && ox0 + row < a.ow && j * 32 + col8 + 8 <= a.cout;
                    const size_t m = (size_t)(n * a.oh + (ok ? oy : 0)) * a.ow + (ok ? ox0 + row : 0);
                    zpre[qj][pass] = ok ? *reinterpret_cast<const u32x4*>(sstat.zb[j] + m * sstat.zld) : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        f32x16 acc[RPW][TN];
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int r = t / 3, s_ = t - r * 3;
            const char* ap = sH + ((wave + r) * HW + li + s_) * PB + 16 * lh;
#pragma unroll
            for (int c = 0; c < KS; ++c) {
                u32x4 fa[RPW];
#pragma unroll
                for (int q = 0; q < RPW; ++q) fa[q] = *reinterpret_cast<const u32x4*>(ap + q * 4 * HW * PB + c * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    u32x4 b;
                    if constexpr (BREG) b = fb[t][c][j];
                    else b = *reinterpret_cast<const u32x4*>(sW + (j * 32 + li) * WB + (t * CIN + c * 16 + 8 * lh) * 2);
#pragma unroll
                    for (int q = 0; q < RPW; ++q) acc[q][j] = mfma16<T>(fa[q], b, acc[q][j]);
                }
            }
        }
        if constexpr (ALIAS) __syncthreads();              // every wave has read its fragments: the halo becomes staging
        if constexpr (POOL) {
            const int pp = lane >> 2;                      // this lane's pooled pixel of the strip (15: none)
#pragma unroll
            for (int qj = 0; qj < RPW * TN; ++qj) {
                const int q = qj / TN, j = qj % TN;
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[(4 * lh + (r & 3) + 8 * (r >> 2)) * SW + li] = acc[q][j][r];
                __builtin_amdgcn_wave_barrier();
                if (pp < PPX) {
                    float mx[8];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int row = 2 * pp + t;
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * SW + col8);
                        const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * SW + col8 + 4);
                        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v[e] = fmaxf(v[e] * sc[j][e] + sh[j][e], 0.f);
                            mx[e] = t == 0 ? v[e] : fmaxf(mx[e], v[e]);
                        }
                    }
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(mx[2 * e], mx[2 * e + 1]);
                    *reinterpret_cast<u32x4*>(sX + (wave + 4 * q) * XROW + (j * PPX + pp) * 64 + (lane & 3) * 16) = o;
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();                               // the tile's 8 reduced rows are in sX
            auto pkmax = [](u32x4 p, u32x4 q_) {
                u32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) asm("v_pk_max_i16 %0, %1, %2" : "=v"(r[e]) : "v"(p[e]), "v"(q_[e]));
                return r;
            };
            const int kt = oy0 / TH;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int c = tid + it * 256;              // chunk (pooled row of the tile, column tile, pixel, 8 channels)
                if (c >= 4 * 2 * PPX * 4) continue;
                const int rsel = c / (2 * PPX * 4), cc = c - rsel * (2 * PPX * 4);
                const int pc = ox0 / 2 + (cc % (PPX * 4)) / 4;
                const char* xp = sX + cc * 16;
                unsigned short* yp = y + ((size_t)n * a.ph * a.pw + pc) * a.y_ld + (cc / (PPX * 4)) * 32 + (cc & 3) * 8;
                if (rsel < 3) {
                    const int pr = 4 * kt + rsel;
                    const u32x4 m = pkmax(pkmax(*reinterpret_cast<const u32x4*>(xp + (2 * rsel) * XROW),
                                                *reinterpret_cast<const u32x4*>(xp + (2 * rsel + 1) * XROW)),
                                          *reinterpret_cast<const u32x4*>(xp + (2 * rsel + 2) * XROW));
                    if (pr < a.ph && pc < a.pw) *reinterpret_cast<u32x4*>(yp + (size_t)pr * a.pw * a.y_ld) = m;
                } else {
                    const int pr = 4 * kt - 1;             // the row whose third input row is this tile's first
                    if (kt > 0 && pr < a.ph && pc < a.pw)
                        *reinterpret_cast<u32x4*>(yp + (size_t)pr * a.pw * a.y_ld) =
                            pkmax(carry, *reinterpret_cast<const u32x4*>(xp));
                    carry = pkmax(*reinterpret_cast<const u32x4*>(xp + 6 * XROW), *reinterpret_cast<const u32x4*>(xp + 7 * XROW));
                }
            }
            continue;
        }
        // transposing epilogue (see lp_epilogue_staged): accumulators -> private LDS block -> 8 channels per lane
#pragma unroll
        for (int qj = 0; qj < RPW * TN; ++qj) {
            const int q = qj / TN, j = qj % TN;
            const int oy = oy0 + wave + 4 * q;
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[(4 * lh + (r & 3) + 8 * (r >> 2)) * SW + li] = acc[q][j][r];
            __builtin_amdgcn_wave_barrier();
            const int colj = j * 32 + col8;
            const int nvalid = min(8, a.cout - colj);
#pragma unroll
            for (int pass = 0; pass < 32 / RPP; ++pass) {
                const int row = pass * RPP + rrow;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * SW + col8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * SW + col8 + 4);
                if (oy >= a.oh || nvalid <= 0 || ox0 + row >= a.ow) continue;
                const size_t m = (size_t)(n * a.oh + oy) * a.ow + ox0 + row;
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[j][e] + sh[j][e];
                if (res) {
                    const unsigned short* rp = res + m * a.res_ld + colj;
                    if (vec && nvalid == 8) {
                        const u32x4 rv = *reinterpret_cast<const u32x4*>(rp);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[2 * q] += from_bits<T>((unsigned short)(rv[q] & 0xffffu));
                            v[2 * q + 1] += from_bits<T>((unsigned short)(rv[q] >> 16));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (e < nvalid) v[e] += from_bits<T>(rp[e]);
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (colj + e < a.relu_limit) ? fmaxf(v[e], 0.f) : v[e];
                }
                if constexpr (STATS != 0) {               // sums of the values exactly as stored below
                    if (nvalid == 8 && !(a.st.dbg & 8192)) {
                        float rr[8];
                        unsigned zq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                        for (int e = 0; e < 8; ++e) rr[e] = from_bits<T>(to_bits<T>(v[e]));
                        if constexpr (STATS == gvconv::STAT_BWD) {
                            const u32x4 zz = zpre[qj][pass];
                            zq[0] = zz[0]; zq[1] = zz[1]; zq[2] = zz[2]; zq[3] = zz[3];
                        }
                        if (j == 0) sstat.template add<T, 0>(rr, zq);
                        else sstat.template add<T, (TN > 1 ? 1 : 0)>(rr, zq);
                    }
                }
                store_chunk<T>(y + m * a.y_ld + colj, v, nvalid, vec);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (STATS != 0) sstat.finish(a.st, stage, lane, sgrp, a.cout);
}

// ---------------------------------------------------------------------------------------------------------------
// The 3-channel stems (Inception Conv2d_1a 3x3/2, ResNet conv1 7x7/2 with explicit pad) straight from the fp32
// images (GV_CONV_X_F32).  The gather path of conv_igemm_lp issues one 4-byte global load, one table lookup and a
// bounds test per (pixel, k) element.  Here a workgroup owns a 32-pixel-wide column strip of one image and walks down
// it 4 output rows at a time: the input patch ((3*2 + KW) rows x (31*2 + KW) pixels x 3 channels) is loaded with
// coalesced row reads, rounded to the storage type once and kept in LDS; k is re-ordered row-wise — filter row r
// owns KR = 16 (3x3) or 24 (7x7) slots of which KW*3 are real — so the 8 values of an MFMA fragment are 8
// CONSECUTIVE patch elements: four ds_read_b32 (lane stride 12 B: conflict-free), no per-element addressing at all.
// The filter is re-ordered the same way into LDS once per workgroup.
// POOL (GV_CONV_MAXPOOL3S2 / GV_CONV_MAXPOOL3S2_SAME; ResNet-v2's conv1 -> pool1, nets/resnet_v2.py:178-181): as in
// conv3x3_halo_lp — strips advance 30 columns, each wave reduces its row horizontally on the way out of its staging block,
// the vertical maxima are taken on the packed words after a barrier, the row that needs the next tile's first row waits in
// a register.  Here the values are signed (conv1 has a bias and no activation), so the reduced rows hold ORDER KEYS
// (bits ^ 0x7fff where the sign is set: signed 16-bit order = float order, -0 < +0), a missing row or column (TF's SAME
// pads (0, 1) on an even map: the window is clipped) is the lowest key, and the store turns keys back into values.
// POOL staging (round 6): a wave's 32 x 32 block leaves the accumulators ALREADY scaled, shifted and rounded to the storage
// type (one channel per lane there: one scale and one shift register per column block instead of 8 + 8), as 16-bit values in
// rows of 96 bytes — rounding is monotonic, so the maximum of the rounded values is the rounded maximum the fp32 form took —
// and the horizontal reduce is three 16-byte reads and two packed maxima on order keys per lane.  12 KB of staging instead
// of 18 and ~40 registers fewer: THREE workgroups per CU (LDS 49.6 KB each, <= 168 registers) instead of two; the kernel
// waits on its patch loads, not on arithmetic.
template <typename T, int TN, int KW, int STATS = 0, bool POOL = false>
__global__ __launch_bounds__(256, POOL ? 3 : 2) void conv_stem_patch_lp(const ConvArgs a) {
    static_assert(!POOL || (TN == 2 && STATS == 0), "the pooled form: 64 output channels, no BatchNorm sums");
    constexpr int PSTEP = 30, PPX = 15, XROW = 2 * PPX * 32 * 2;
    constexpr int SROW = 96;                                // POOL: bytes of a staged row (32 x 16 bit + pad: conflict-free reads)
    constexpr int KR = KW == 3 ? 16 : 24;                   // k slots per filter row (multiple of 8)
    constexpr int NG = (KW * KR / 8 + 1) / 2 * 2;           // 8-value groups, padded to whole 16-deep k-steps
    constexpr int PR = 3 * 2 + KW + 1;                      // patch rows (+1 zero row for the padding group)
    constexpr int PC = 31 * 2 + KW;                         // patch pixels per row
    constexpr int PITCH = (PC * 3 * 2 + 16 + 3) / 4 * 4;    // patch row bytes (+ slack for the last fragment)
    constexpr int NEL = (PR - 1) * PC * 3;                  // patch elements loaded per tile
    constexpr int SL = (NEL + 255) / 256;
    constexpr int WB = NG * 16 + 16;                        // LDS filter row bytes
    constexpr int SW = 32 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* sP = smem_raw;                                                      // [PR][PITCH]
    constexpr int STAGE_B = POOL ? 32 * SROW : 32 * SW * 4;                   // a wave's staging block
    float* stage = reinterpret_cast<float*>(smem_raw + PR * PITCH + (threadIdx.x >> 6) * STAGE_B);
    char* sW = smem_raw + PR * PITCH + 4 * STAGE_B;                           // [32*TN][WB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tiles_x = POOL ? (a.pw + PPX - 1) / PPX : (a.ow + 31) / 32;
    const int n = blockIdx.x / tiles_x;
    const int ox0 = (blockIdx.x % tiles_x) * (POOL ? PSTEP : 32);
    const int oh_run = POOL ? min(a.oh, 2 * a.ph + 1) : a.oh;                 // output rows somebody needs
    char* sX = sW + 32 * TN * WB;                                             // [4][XROW]   (POOL)
    u32x4 carry = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(a.w);
    unsigned short* y = reinterpret_cast<unsigned short*>(a.y);

    for (int idx = tid; idx < 32 * TN * NG * 8; idx += 256) {               // filter in the row-wise k order
        const int row = idx / (NG * 8), kk = idx - row * (NG * 8);
        const int r = kk / KR, j = kk - r * KR;
        unsigned short v = 0;
        if (row < a.cout && r < KW && j < KW * 3) v = wp[(size_t)row * a.Kpad + r * KW * 3 + j];
        *reinterpret_cast<unsigned short*>(sW + row * WB + kk * 2) = v;
    }
    for (int idx = tid; idx < PR * PITCH / 4; idx += 256)                    // zero row + the slack at every row end
        reinterpret_cast<unsigned*>(sP)[idx] = 0u;                           // (read against zero filter slots)

    const int rrow = lane >> 2, col8 = (lane & 3) * 8;
    float sc[TN][8], sh[TN][8];
    float sc_l[TN], sh_l[TN];                               // POOL: the constants of this lane's accumulator column
    bool relu_l[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if constexpr (POOL) {
            const int c = min(j * 32 + li, a.cout - 1);
            sc_l[j] = a.scale[c];
            sh_l[j] = a.shift[c];
            relu_l[j] = a.relu && j * 32 + li < a.relu_limit;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = min(j * 32 + col8 + e, a.cout - 1);
                sc[j][e] = a.scale[c];
                sh[j][e] = a.shift[c];
            }
        }
    }
    (void)sc_l; (void)sh_l; (void)relu_l; (void)sc; (void)sh;
    const bool vec = (a.y_ld % 8 == 0) && ((((uintptr_t)y) & 15) == 0);

    gvconv::StatStrip<gvconv::STAT_FWD, TN> sstat;          // STATS (forward sums only: this kernel has no data-gradient use)
    const int sgrp = STATS != 0 ? n % a.st.G : 0;
    if constexpr (STATS != 0) sstat.init(a.st, sgrp, col8, a.cout);

    // the patch, one element per thread and step (a wave's load is 256 contiguous bytes of a patch row); what does not depend
    // on the tile — the element's patch row, its column test, its offsets in the image and in LDS — is worked out once
    // (12-byte loads, a pixel per thread, were measured too: fewer instructions, but a wave's load then spans 768 bytes at
    // a 12-byte lane pitch and Conv2d_1a — at the HBM roofline with these loads — fell from 4.5 to 3.1 TB/s)
    float pr_[SL];
    int p_goff[SL], p_info[SL];                             // element offset from the tile's first patch row; LDS offset | row << 16 | ok << 24 | live << 25
#pragma unroll
    for (int k = 0; k < SL; ++k) {
        const int idx = tid + k * 256;
        const int prow = idx / (PC * 3), e = idx - prow * (PC * 3);
        const int px = e / 3, ch = e - px * 3;
        const int ix = ox0 * 2 - a.pad_l + px;
        const bool live = idx < NEL, ok = live && (unsigned)ix < (unsigned)a.iw;
        p_goff[k] = (prow * a.iw + ix) * a.x_ld + ch;
        p_info[k] = (prow * PITCH + e * 2) | (prow << 16) | ((ok ? 1 : 0) << 24) | ((live ? 1 : 0) << 25);
    }
    auto fetch = [&](int oy0) {
        const int iy0 = oy0 * 2 - a.pad_t;
        const ptrdiff_t row0 = ((ptrdiff_t)n * a.ih + iy0) * (ptrdiff_t)a.iw * a.x_ld;   // (may lie in front of the image: only offsets of rows inside it are read)
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            const int iy = iy0 + ((p_info[k] >> 16) & 0xff);
            float v = 0.f;
            if (((p_info[k] >> 24) & 1) && (unsigned)iy < (unsigned)a.ih) v = a.x[row0 + p_goff[k]];
            pr_[k] = v;
        }
    };
    // order key <-> value of the two 16-bit elements of a word (an involution: the sign bit stays)
    auto keyw = [](unsigned w) { const unsigned sg = (w >> 15) & 0x00010001u; return w ^ ((sg << 15) - sg); };
    auto pkmax = [](u32x4 p, u32x4 q_) {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) asm("v_pk_max_i16 %0, %1, %2" : "=v"(r[e]) : "v"(p[e]), "v"(q_[e]));
        return r;
    };
    // POOL: this thread's chunk of a pooled row (column tile, pooled pixel, 8 channels): threads [0, 120) take the row a
    // tile completes, threads [120, 240) the one that waits for the next tile
    const int pcc = tid % (2 * PPX * 4), prs = tid / (2 * PPX * 4);
    const int ppc = ox0 / 2 + (pcc % (PPX * 4)) / 4;
    unsigned short* ypool = y + ((size_t)n * a.ph * a.pw + ppc) * a.y_ld + (pcc / (PPX * 4)) * 32 + (pcc & 3) * 8;
    // GV_CONV_POOL_ACT2: the pooled value, as it would be stored, goes through a second per-channel affine (+ ReLU) on its way
    // out — ResNet-v2's first `preact` BatchNorm + ReLU (nets/resnet_v2.py:75 behind :181), whose only input is pool1
    const bool post = POOL && a.scale2 != nullptr;
    float psc[8], psh[8];                                   // (16 registers: the kernel has them to spare at three workgroups per CU)
    if (post) {
        const int cb = (pcc / (PPX * 4)) * 32 + (pcc & 3) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            psc[e] = a.scale2[min(cb + e, a.cout - 1)];
            psh[e] = a.shift2[min(cb + e, a.cout - 1)];
        }
    }
    auto put_pooled = [&](int prow, u32x4 k) {
        if (prow < a.ph && ppc < a.pw) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = keyw(k[e]);
            if (post) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo = from_bits<T>((unsigned short)(o[e] & 0xffffu)) * psc[2 * e] + psh[2 * e];
                    float hi = from_bits<T>((unsigned short)(o[e] >> 16)) * psc[2 * e + 1] + psh[2 * e + 1];
                    if (a.relu2) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                    o[e] = pack2<T>(lo, hi);
                }
            }
            *reinterpret_cast<u32x4*>(ypool + (size_t)prow * a.pw * a.y_ld) = o;
        }
    };
    fetch(0);
    for (int oy0 = 0; oy0 < oh_run; oy0 += 4) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SL; ++k)
            if ((p_info[k] >> 25) & 1) *reinterpret_cast<unsigned short*>(sP + (p_info[k] & 0xffff)) = to_bits<T>(pr_[k]);
        __syncthreads();
        if (oy0 + 4 < oh_run) fetch(oy0 + 4);
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NG / 2; ++c) {
            const int g = 2 * c + lh;                      // this half-wave's 8-value group
            const int r = g / (KR / 8), q = g - r * (KR / 8);   // filter row, 8-slot group inside it
            const int prow = r < KW ? 2 * wave + r : PR - 1;    // the padding group reads the zero row
            const char* ap = sP + prow * PITCH + li * 12 + q * 16;
            u32x4 fa;
#pragma unroll
            for (int d = 0; d < 4; ++d) fa[d] = *reinterpret_cast<const unsigned*>(ap + 4 * d);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const u32x4 b = *reinterpret_cast<const u32x4*>(sW + (j * 32 + li) * WB + g * 16);
                acc[j] = mfma16<T>(fa, b, acc[j]);
            }
        }
        const int oy = oy0 + wave;
        if constexpr (POOL) {
            const int pp = lane >> 2;                      // this lane's pooled pixel of the strip (15: none)
            char* stage_b = reinterpret_cast<char*>(stage);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {             // the value as it would be stored: row = pixel, column = channel
                    float v = acc[j][r] * sc_l[j] + sh_l[j];
                    if (relu_l[j]) v = fmaxf(v, 0.f);
                    *reinterpret_cast<unsigned short*>(stage_b + (4 * lh + (r & 3) + 8 * (r >> 2)) * SROW + li * 2) = to_bits<T>(v);
                }
                __builtin_amdgcn_wave_barrier();
                if (pp < PPX) {
                    const u32x4 lowest = {0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
                    u32x4 mx = lowest;
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int row = 2 * pp + t;
                        u32x4 k = *reinterpret_cast<const u32x4*>(stage_b + row * SROW + (lane & 3) * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) k[e] = keyw(k[e]);
                        mx = pkmax(mx, ox0 + row < a.ow ? k : lowest);
                    }
                    *reinterpret_cast<u32x4*>(sX + wave * XROW + (j * PPX + pp) * 64 + (lane & 3) * 16) = oy < a.oh ? mx : lowest;
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();                               // the tile's 4 reduced rows are in sX
            const int kt = oy0 >> 2;
            if (prs < 2) {
                const char* xp = sX + pcc * 16;
                const u32x4 x2 = *reinterpret_cast<const u32x4*>(xp + 2 * XROW);
                if (prs == 0) {
                    put_pooled(2 * kt, pkmax(pkmax(*reinterpret_cast<const u32x4*>(xp), *reinterpret_cast<const u32x4*>(xp + XROW)), x2));
                } else {
                    if (kt > 0) put_pooled(2 * kt - 1, pkmax(carry, *reinterpret_cast<const u32x4*>(xp)));
                    carry = pkmax(x2, *reinterpret_cast<const u32x4*>(xp + 3 * XROW));
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[(4 * lh + (r & 3) + 8 * (r >> 2)) * SW + li] = acc[j][r];
            __builtin_amdgcn_wave_barrier();
            const int colj = j * 32 + col8;
            const int nvalid = min(8, a.cout - colj);
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = pass * 16 + rrow;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * SW + col8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * SW + col8 + 4);
                if (oy >= a.oh || nvalid <= 0 || ox0 + row >= a.ow) continue;
                const size_t m = (size_t)(n * a.oh + oy) * a.ow + ox0 + row;
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = v[e] * sc[j][e] + sh[j][e];
                    if (a.relu && colj + e < a.relu_limit) v[e] = fmaxf(v[e], 0.f);
                }
                if constexpr (STATS != 0) {               // sums of the values exactly as stored below
                    if (nvalid == 8 && !(a.st.dbg & 8192)) {
                        float rr[8];
                        const unsigned zq[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                        for (int e = 0; e < 8; ++e) rr[e] = from_bits<T>(to_bits<T>(v[e]));
                        if (j == 0) sstat.template add<T, 0>(rr, zq);
                        else sstat.template add<T, (TN > 1 ? 1 : 0)>(rr, zq);
                    }
                }
                store_chunk<T>(y + m * a.y_ld + colj, v, nvalid, vec);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (POOL) {
        // the row that was waiting for a tile that does not exist (SAME: its window ends with the map)
        if (prs == 1) put_pooled(2 * ((oh_run + 3) >> 2) - 1, carry);
    }
    if constexpr (STATS != 0) sstat.finish(a.st, stage, lane, sgrp, a.cout);
}

// [kh][kw][cin][cout] fp32 -> [cout][Kpad] T, k = (r*kw+s)*cin + c, zero filled to a multiple of 32
template <typename T>
__global__ void pack_filter_lp(const float* __restrict__ w, int K, int Kpad, int cout,
                               unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)cout * Kpad) return;
    const int n = (int)(i / Kpad);
    const int k = (int)(i - (int64_t)n * Kpad);
    out[i] = to_bits<T>(k < K ? w[(size_t)k * cout + n] : 0.f);
}

// The same packing for MANY filters in one launch (the training step re-packs every filter, forward and data-gradient
// form, after each update): block -> job through a table; `flipped` packs the data-gradient filter
// W'[r',s',co,ci] = W[kh-1-r', kw-1-s', ci, co] straight from the forward HWIO variable.
template <typename T>
__global__ __launch_bounds__(256) void pack_filters_batched_lp(const gv_pack_job* __restrict__ jobs,
                                                               const int* __restrict__ block_job) {
    const gv_pack_job j = jobs[block_job[blockIdx.x]];
    const int co_p = j.flipped ? j.cin : j.cout;                 // rows of this job in the packed image
    const int ci_p = j.flipped ? j.cout : j.cin;                 // this job's channels per tap on the k axis
    const int K = j.kh * j.kw * ci_p;
    const int Kpad = (K + KT - 1) / KT * KT;
    if (!j.flipped && co_p % 16 == 0) {
        // The forward image is the TRANSPOSE of the HWIO variable (out[n][k] = w[k][n]): with one thread per output element
        // a wave reads 64 elements 4*cout bytes apart.  The job's blocks — co_p * Kpad / 256 of them, exactly — take 16 x 16
        // tiles instead: 16 rows of 64 contiguous source bytes in, through LDS, 16 rows of 32 contiguous bytes out.
        __shared__ float tile[16][17];
        const int tiles_k = Kpad / 16, bl = (int)blockIdx.x - j.first_block;
        const int n0 = (bl / tiles_k) * 16, k0 = (bl % tiles_k) * 16;
        const int a = threadIdx.x >> 4, b = threadIdx.x & 15;
        const size_t wld = j.w_ld > 0 ? (size_t)j.w_ld : (size_t)j.cout;
        tile[a][b] = k0 + a < K ? j.w[(size_t)(k0 + a) * wld + n0 + b] : 0.f;
        __syncthreads();
        reinterpret_cast<unsigned short*>(j.out)[(size_t)(n0 + a) * Kpad + k0 + b] = to_bits<T>(tile[b][a]);
        return;
    }
    const int64_t i = (int64_t)(blockIdx.x - j.first_block) * 256 + threadIdx.x;
    if (i >= (int64_t)co_p * Kpad) return;
    const int n = (int)(i / Kpad);
    const int k = (int)(i - (int64_t)n * Kpad);
    float v = 0.f;
    int64_t dst = i;
    const size_t wld = j.w_ld > 0 ? (size_t)j.w_ld : (size_t)j.cout;       // row stride of the HWIO source
    if (k < K) {
        if (!j.flipped) {
            v = j.w[(size_t)k * wld + n];
        } else {
            const int tap = k / ci_p, co = k - tap * ci_p;
            const int r = tap / j.kw, sx = tap - r * j.kw;
            // source tap of packed tap (r, sx): the flipped window — of the whole filter, or (sub_step 2) of the taps one
            // parity class of a stride-2 data gradient uses
            const int sr = j.sub_step == 2 ? j.sub_r0 + 2 * (j.kh - 1 - r) : j.kh - 1 - r;
            const int ss = j.sub_step == 2 ? j.sub_s0 + 2 * (j.kw - 1 - sx) : j.kw - 1 - sx;
            const int skw = j.sub_step == 2 ? j.src_kw : j.kw;
            v = j.w[((size_t)(sr * skw + ss) * j.cin + n) * wld + co];
            if (j.k_total > 0) {                                 // one member of a wider fused filter: its column range
                const int kt = j.kh * j.kw * j.k_total;
                dst = (int64_t)n * ((kt + KT - 1) / KT * KT) + (int64_t)tap * j.k_total + j.k_off + co;
            }
        }
    } else if (j.flipped && j.k_total > 0) {
        return;                                                  // (the fused image's own padding was zeroed once)
    }
    reinterpret_cast<unsigned short*>(j.out)[dst] = to_bits<T>(v);
}

struct TileCfg { int bm, bn; };
constexpr TileCfg kTiles[] = {{128, 128}, {128, 64}, {64, 64}, {128, 96}, {64, 128}, {128, 32},
                              {256, 128}, {128, 256}, {256, 64},    // these three: 8 waves
                              {128, 192}, {64, 192},                // one n-tile for the many 192-channel layers
                              {256, 192}};                          // 8 waves: least operand traffic per flop (Conv2d_4a)
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <typename T, int WM, int WN, int TM, int TN, bool GENERIC, bool XF32, bool XPRE = false, int STATS = 0>
int launch_one(const ConvArgs& a, int64_t nwg, size_t lds, hipStream_t st) {
    if (lds > 160 * 1024) return GV_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
        const bool ok = GV_BIG_LDS_OK((&conv_igemm_lp<T, WM, WN, TM, TN, GENERIC, XF32, XPRE, STATS>), 160 * 1024);
        if (!ok) return GV_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL((conv_igemm_lp<T, WM, WN, TM, TN, GENERIC, XF32, XPRE, STATS>), dim3((unsigned)nwg), dim3(WM * WN * 64), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

// the tiles the pre-activation-on-load loader is instantiated for: 128 / 256 rows x 64 / 128 / 256 columns (the conv1 of a
// ResNet unit has 64 ... 512 output channels)
constexpr bool xpre_cfg(int wm, int wn, int tm, int tn) {
    return (wm == 2 && wn == 2 && tm == 2 && (tn == 2 || tn == 1)) || (wm == 4 && wn == 2 && tm == 2 && (tn == 2 || tn == 1)) ||
           (wm == 2 && wn == 4 && tm == 2 && tn == 2);
}

template <typename T, int WM, int WN, int TM, int TN>
int launch_cfg(const ConvArgs& a0, bool generic, bool xf32, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    ConvArgs a = a0;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    const int tiles_m = gv_ceil_div(a.M, BM);
    const int64_t nwg = (int64_t)tiles_m * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    const size_t lds_main = (size_t)(2 * BM + 2 * BN) * RB + ((generic || xf32) ? (size_t)a.Kpad * 8 : 0) +
                            (a.xscale ? (size_t)a.cin * 8 : 0);
    const size_t lds_epi = (size_t)(WM * WN) * EpiGeom<TN>::BYTES;
    size_t lds = ((lds_main > lds_epi ? lds_main : lds_epi) + 15) / 16 * 16 + 16 * BN;   // + the epilogue's constants
    if (a.st.mode != gvconv::STAT_OFF) {              // BatchNorm sums in the epilogue: vector loader, plain epilogue
        if (a.xscale || xf32 || generic || !gvconv::stat_tile_ok(a.st, BM, a.cout, TM * 32)) return GV_E_UNSUPPORTED;
        a.st.slots = gvconv::stat_rows(BM, a.st.hw, a.st.G);
        a.st.fold = gvconv::stat_slots(BM, a.st.hw) > a.st.G ? 1 : 0;
        // the tables live in the main-loop buffers the epilogue has freed, behind the waves' staging blocks, where they
        // fit in front of the epilogue's constants; else behind everything (more LDS per workgroup)
        const size_t tab = gvconv::stat_lds_bytes(a.st.mode, a.st.slots, BN), ss_off = ((lds_main > lds_epi ? lds_main : lds_epi) + 15) / 16 * 16;
        if (lds_epi + tab <= ss_off) {
            a.st.lds_off = (int)lds_epi;
        } else {
            a.st.lds_off = (int)lds;
            lds += tab;
        }
        if (a.st.mode == gvconv::STAT_FWD) return launch_one<T, WM, WN, TM, TN, false, false, false, gvconv::STAT_FWD>(a, nwg, lds, st);
        return launch_one<T, WM, WN, TM, TN, false, false, false, gvconv::STAT_BWD>(a, nwg, lds, st);
    }
    if (a.xscale) {
        if constexpr (xpre_cfg(WM, WN, TM, TN)) {
            if (gvconv::lp_epilogue_lean_ok(a)) return launch_one<T, WM, WN, TM, TN, false, false, true, gvconv::STAT_LEAN>(a, nwg, lds, st);
            return launch_one<T, WM, WN, TM, TN, false, false, true>(a, nwg, lds, st);
        } else {
            return GV_E_UNSUPPORTED;
        }
    }
    if (xf32) return launch_one<T, WM, WN, TM, TN, true, true>(a, nwg, lds, st);
    if (generic) return launch_one<T, WM, WN, TM, TN, true, false>(a, nwg, lds, st);
    if (gvconv::lp_epilogue_lean_ok(a)) return launch_one<T, WM, WN, TM, TN, false, false, false, gvconv::STAT_LEAN>(a, nwg, lds, st);
    return launch_one<T, WM, WN, TM, TN, false, false>(a, nwg, lds, st);
}

template <typename T>
int launch_t(int cfg, const ConvArgs& a, bool generic, bool xf32, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_cfg<T, 2, 2, 2, 2>(a, generic, xf32, st);
        case 1: return launch_cfg<T, 2, 2, 2, 1>(a, generic, xf32, st);
        case 2: return launch_cfg<T, 2, 2, 1, 1>(a, generic, xf32, st);
        case 3: return launch_cfg<T, 4, 1, 1, 3>(a, generic, xf32, st);
        case 4: return launch_cfg<T, 2, 2, 1, 2>(a, generic, xf32, st);
        case 5: return launch_cfg<T, 4, 1, 1, 1>(a, generic, xf32, st);
        case 6: return launch_cfg<T, 4, 2, 2, 2>(a, generic, xf32, st);
        case 7: return launch_cfg<T, 2, 4, 2, 2>(a, generic, xf32, st);
        case 8: return launch_cfg<T, 4, 2, 2, 1>(a, generic, xf32, st);
        case 9: return launch_cfg<T, 2, 2, 2, 3>(a, generic, xf32, st);
        case 10: return launch_cfg<T, 2, 2, 1, 3>(a, generic, xf32, st);
        case 11: return launch_cfg<T, 4, 2, 2, 3>(a, generic, xf32, st);
    }
    return GV_E_UNSUPPORTED;
}

// (STATS: the instantiation that also produces the train-mode BatchNorm sums of its output, conv_stats.h)
template <typename T, int RPW, int STATS>
int launch_halo_r(const ConvArgs& a, hipStream_t st) {
    const int tiles_x = (a.ow + 31) / 32;
    const dim3 grid((unsigned)(a.nb * tiles_x));
    const size_t halo = (size_t)(4 * RPW + 2) * 34 * 80;
    if (a.cout <= 32) {
        const size_t lds = halo + 4 * 32 * (32 + 4) * 4;
        hipLaunchKernelGGL((conv3x3_halo_lp<T, 1, RPW, 32, STATS>), grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = halo + (RPW == 2 ? 0 : 4 * 32 * (32 + 4) * 4) + 64 * (288 * 2 + 16);   // (RPW = 2: staging aliases the halo)
        const bool ok = GV_BIG_LDS_OK((&conv3x3_halo_lp<T, 2, RPW, 32, STATS>), 160 * 1024);
        if (!ok) return GV_E_UNSUPPORTED;
        hipLaunchKernelGGL((conv3x3_halo_lp<T, 2, RPW, 32, STATS>), grid, dim3(256), lds, st, a);
    }
    GV_LAUNCH_CHECK();
    return GV_OK;
}

// GV_CONV_MAXPOOL3S2: 27 KB halo (staging aliased) + 38 KB filter + 15 KB of reduced rows = 80 448 B: two per CU
template <typename T>
int launch_halo_pool(const ConvArgs& a, hipStream_t st) {
    const int tiles_x = (a.pw + 14) / 15;
    const size_t lds = (size_t)10 * 34 * 80 + 64 * (288 * 2 + 16) + 8 * (2 * 15 * 32 * 2);
    const bool ok = GV_BIG_LDS_OK((&conv3x3_halo_lp<T, 2, 2, 32, 0, true>), 160 * 1024);
    if (!ok) return GV_E_UNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_halo_lp<T, 2, 2, 32, 0, true>), dim3((unsigned)(a.nb * tiles_x)), dim3(256), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T, int STATS>
int launch_halo64(const ConvArgs& a, hipStream_t st) {                       // 64 input channels, <= 32 output channels
    const int tiles_x = (a.ow + 31) / 32;
    const size_t lds = (size_t)6 * 34 * (64 * 2 + 16) + 32 * (9 * 64 * 2 + 16);      // (staging aliases the halo)
    const bool ok = GV_BIG_LDS_OK((&conv3x3_halo_lp<T, 1, 1, 64, STATS>), 160 * 1024);
    if (!ok) return GV_E_UNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_halo_lp<T, 1, 1, 64, STATS>), dim3((unsigned)(a.nb * tiles_x)), dim3(256), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T, int STATS>
int launch_halo_s(const ConvArgs& a, hipStream_t st) {
    if (a.cin == 64) return launch_halo64<T, STATS>(a, st);
    // two rows per wave wherever the map has them: Conv2d_2a 0.225 -> 0.150 ms.  With the 64-column filter in LDS the taller
    // halo would cost the second resident workgroup (84 KB: Conv2d_2b 0.295 -> 0.385 ms), so that form lets the staging
    // blocks alias the halo (65 KB): 0.335 -> 0.283 ms.  Debug bit 1024: one row per wave (A/B)
    return (a.oh >= 8 && !(a.dbg & 1024)) ? launch_halo_r<T, 2, STATS>(a, st) : launch_halo_r<T, 1, STATS>(a, st);
}

template <typename T>
int launch_halo(const ConvArgs& a, hipStream_t st) {
    if (a.pool) return launch_halo_pool<T>(a, st);
    if (a.st.mode == gvconv::STAT_FWD) return launch_halo_s<T, gvconv::STAT_FWD>(a, st);
    // The BACKWARD sums are not folded into this kernel: they need z next to every chunk of dy it stores, and one wave per
    // SIMD with its loads retiring in order has nothing to hide 290 MB of extra reads behind — measured on the data
    // gradients of Conv2d_2a / 2b: 0.145 -> 0.29 ms fetched in the epilogue, 0.34 ms prefetched in front of the MFMAs,
    // against 0.14 ms for the separate sums pass it would replace (the code path exists: STATS = STAT_BWD compiles)
    if (a.st.mode == gvconv::STAT_BWD) return GV_E_UNSUPPORTED;
    return launch_halo_s<T, 0>(a, st);
}

template <typename T, int TN, int KW, int STATS>
int launch_stem_one(const ConvArgs& a, hipStream_t st) {
    constexpr int KR = KW == 3 ? 16 : 24, NG = (KW * KR / 8 + 1) / 2 * 2, PR = 3 * 2 + KW + 1, PC = 31 * 2 + KW;
    constexpr int PITCH = (PC * 3 * 2 + 16 + 3) / 4 * 4;
    const size_t lds = (size_t)PR * PITCH + 4 * 32 * 36 * 4 + (size_t)32 * TN * (NG * 16 + 16);
    hipLaunchKernelGGL((conv_stem_patch_lp<T, TN, KW, STATS>), dim3((unsigned)(a.nb * ((a.ow + 31) / 32))), dim3(256), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T, int KW>
int launch_stem_pool(const ConvArgs& a, hipStream_t st) {
    constexpr int KR = KW == 3 ? 16 : 24, NG = (KW * KR / 8 + 1) / 2 * 2, PR = 3 * 2 + KW + 1, PC = 31 * 2 + KW;
    constexpr int PITCH = (PC * 3 * 2 + 16 + 3) / 4 * 4;
    const size_t lds = (size_t)PR * PITCH + 4 * 32 * 96 + (size_t)32 * 2 * (NG * 16 + 16) + 4 * (2 * 15 * 32 * 2);   // (16-bit staging rows)
    hipLaunchKernelGGL((conv_stem_patch_lp<T, 2, KW, 0, true>), dim3((unsigned)(a.nb * ((a.pw + 14) / 15))), dim3(256), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T>
int launch_stem(const ConvArgs& a, hipStream_t st) {
    if (a.pool) return a.kw == 3 ? launch_stem_pool<T, 3>(a, st) : launch_stem_pool<T, 7>(a, st);
    if (a.st.mode == gvconv::STAT_BWD) return GV_E_UNSUPPORTED;             // (a first layer has no data gradient)
    if (a.st.mode == gvconv::STAT_FWD) {
        if (a.kw == 3) return a.cout <= 32 ? launch_stem_one<T, 1, 3, gvconv::STAT_FWD>(a, st) : launch_stem_one<T, 2, 3, gvconv::STAT_FWD>(a, st);
        return a.cout <= 32 ? launch_stem_one<T, 1, 7, gvconv::STAT_FWD>(a, st) : launch_stem_one<T, 2, 7, gvconv::STAT_FWD>(a, st);
    }
    if (a.kw == 3) return a.cout <= 32 ? launch_stem_one<T, 1, 3, 0>(a, st) : launch_stem_one<T, 2, 3, 0>(a, st);
    return a.cout <= 32 ? launch_stem_one<T, 1, 7, 0>(a, st) : launch_stem_one<T, 2, 7, 0>(a, st);
}

}  // namespace

namespace gvconv {

// configurations: [0, kNumTiles) register-staged tiles, kNumTiles = the strip / halo kernels of the stem layers,
// then the LDS-DMA tiles of conv_dma.hip
int lp_num_cfgs() { return kNumTiles + 1 + dma_lp_num_cfgs(); }
int lp_special_cfg() { return kNumTiles; }

// the 3-channel stems read from the fp32 images: square 3x3 or 7x7 window, stride 2, <= 64 output channels
bool lp_stem_ok(const ConvArgs& a, bool xf32) {
    return xf32 && a.cin == 3 && a.kh == a.kw && (a.kw == 3 || a.kw == 7) && a.stride == 2 && a.cout <= 64 &&
           a.dil_shift == 0 && a.split == 0 && a.y2 == nullptr && a.res == nullptr &&
           (int64_t)a.nb * a.ih * a.iw * a.x_ld < 0x7fffffffll;
}

// the halo kernel's layer class: 3x3 / stride 1, 32 input channels in 16-byte aligned pixels, <= 64 output channels,
// plain epilogue (Conv2d_2a_3x3, Conv2d_2b_3x3)
bool lp_halo_ok(const ConvArgs& a, bool generic) {
    return !generic && a.kh == 3 && a.kw == 3 && a.stride == 1 && ((a.cin == 32 && a.cout <= 64) || (a.cin == 64 && a.cout <= 32)) &&
           a.cout % 8 == 0 &&
           a.dil_shift == 0 && a.split == 0 && a.y2 == nullptr && a.oh == a.ih + 2 * a.pad_t - 2 &&
           a.ow == a.iw + 2 * a.pad_l - 2;
}

bool lp_xpre_cfg_ok(int cfg) { return cfg == 0 || cfg == 1 || cfg == 6 || cfg == 7 || cfg == 8; }
int lp_xpre_pick(int /*M*/, int N) { return N <= 64 ? 1 : 0; }

int lp_pick_tile(int M, int N, int /*K*/) {
    int best = 0;
    double best_cost = 1e30;
    const int order[] = {0, 3, 1, 5};                 // 128 x {128, 96, 64, 32}
    const double pen[] = {1.00, 1.02, 1.05, 1.20};
    for (int t = 0; t < 4; ++t) {
        const int bn = kTiles[order[t]].bn;
        const double cost = (double)gv_ceil_div(N, bn) * bn * pen[t];
        if (cost < best_cost) { best_cost = cost; best = order[t]; }
    }
    const int64_t blocks = (int64_t)gv_ceil_div(M, 128) * gv_ceil_div(N, kTiles[best].bn);
    if (blocks < 1024) {
        if (kTiles[best].bn == 128) best = 4;
        else if (kTiles[best].bn == 64) best = 2;
    }
    return best;
}

// GV_CONV_MAXPOOL3S2: the halo kernel's 32 -> 64 channel form with a plain ReLU epilogue into aligned 16-byte chunks
bool lp_halo_pool_ok(const ConvArgs& a, bool generic) {
    return lp_halo_ok(a, generic) && a.cin == 32 && a.cout == 64 && a.relu && a.relu_limit >= a.cout && a.res == nullptr &&
           a.st.mode == STAT_OFF && a.y_step == 0 && a.xscale == nullptr && a.oh >= 3 && a.ow >= 3 && a.y_ld % 8 == 0 &&
           (((uintptr_t)a.y) & 15) == 0;
}

// ... and the strip kernel of the 3-channel stems at 64 output channels (either pool geometry)
bool lp_stem_pool_ok(const ConvArgs& a, bool xf32) {
    return lp_stem_ok(a, xf32) && a.cout == 64 && a.st.mode == STAT_OFF && a.y_step == 0 && a.xscale == nullptr &&
           a.oh >= 3 && a.ow >= 3 && a.y_ld % 8 == 0 && (((uintptr_t)a.y) & 15) == 0;
}

int lp_launch(int dtype, int cfg, const ConvArgs& a0, bool generic, bool xf32, hipStream_t st) {
    ConvArgs a = a0;
    a.Kpad = (a.K + KT - 1) / KT * KT;
    a.ktiles = a.Kpad / KT;
    if (a.pool && (cfg != kNumTiles || !((a.pool == 1 && lp_halo_pool_ok(a, generic)) || lp_stem_pool_ok(a, xf32))))
        return GV_E_UNSUPPORTED;
    if (a.pool && a.scale2 != nullptr && !lp_stem_pool_ok(a, xf32)) return GV_E_UNSUPPORTED;   // GV_CONV_POOL_ACT2: the stem strip kernel only
    if (cfg == kNumTiles) {
        if (a.y_step != 0) return GV_E_UNSUPPORTED;              // (the strip / halo kernels have no two-level output stride)
        // BatchNorm sums: one segment over every output column (these kernels own whole images: no slot table)
        if (a.st.mode != STAT_OFF && !stat_strip_ok_host(a.st, a.cout)) return GV_E_UNSUPPORTED;
        if (lp_stem_ok(a, xf32)) {
            if (dtype == GV_BF16) return launch_stem<__bf16>(a, st);
            if (dtype == GV_F16) return launch_stem<_Float16>(a, st);
            return GV_E_UNSUPPORTED;
        }
        if (!lp_halo_ok(a, generic)) return GV_E_UNSUPPORTED;
        if (dtype == GV_BF16) return launch_halo<__bf16>(a, st);
        if (dtype == GV_F16) return launch_halo<_Float16>(a, st);
        return GV_E_UNSUPPORTED;
    }
    if (cfg > kNumTiles) {
        if (!dma_lp_ok(a, generic, xf32)) return GV_E_UNSUPPORTED;
        return dma_lp_launch(dtype, cfg - kNumTiles - 1, a, st);
    }
    if (dtype == GV_BF16) return launch_t<__bf16>(cfg, a, generic, xf32, st);
    if (dtype == GV_F16) return launch_t<_Float16>(cfg, a, generic, xf32, st);
    return GV_E_UNSUPPORTED;
}

int64_t lp_packed_bytes(int kh, int kw, int cin, int cout) {
    const int64_t K = (int64_t)kh * kw * cin;
    return (int64_t)cout * ((K + KT - 1) / KT * KT) * 2;
}

int lp_pack_filter(const float* w_hwio, int kh, int kw, int cin, int cout, int dtype, void* out, hipStream_t st) {
    const int K = kh * kw * cin;
    const int Kpad = (K + KT - 1) / KT * KT;
    const dim3 grid((unsigned)gv_ceil_div((int64_t)cout * Kpad, 256));
    if (dtype == GV_BF16)
        hipLaunchKernelGGL(pack_filter_lp<__bf16>, grid, dim3(256), 0, st, w_hwio, K, Kpad, cout, (unsigned short*)out);
    else if (dtype == GV_F16)
        hipLaunchKernelGGL(pack_filter_lp<_Float16>, grid, dim3(256), 0, st, w_hwio, K, Kpad, cout, (unsigned short*)out);
    else
        return GV_E_UNSUPPORTED;
    GV_LAUNCH_CHECK();
    return GV_OK;
}

int lp_pack_filters_batched(const gv_pack_job* jobs_dev, const int* block_job_dev, int nblocks, int dtype,
                            hipStream_t st) {
    if (dtype == GV_BF16)
        hipLaunchKernelGGL(pack_filters_batched_lp<__bf16>, dim3((unsigned)nblocks), dim3(256), 0, st, jobs_dev, block_job_dev);
    else if (dtype == GV_F16)
        hipLaunchKernelGGL(pack_filters_batched_lp<_Float16>, dim3((unsigned)nblocks), dim3(256), 0, st, jobs_dev, block_job_dev);
    else
        return GV_E_UNSUPPORTED;
    GV_LAUNCH_CHECK();
    return GV_OK;
}

}  // namespace gvconv
