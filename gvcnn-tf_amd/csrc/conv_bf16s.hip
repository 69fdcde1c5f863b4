// conv_bf16s.hip — the same implicit-GEMM convolution as conv_igemm.hip, with the fp32 operands split
// into NP bf16 "planes" so the products run on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16,
// 16x the fp32 MFMA rate per instruction):
//
//     a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (exact for fp32 a)
//     a*b ~= a0*b0 + a0*b1 + a1*b0 + a1*b1 + a0*b2 + a2*b0                             (NP = 3: 6 MFMAs)
//
// Every bf16 x bf16 product is exact in the fp32 accumulator; the dropped terms (a1*b2, a2*b1, a2*b2)
// are <= 2^-24 relative to the product, i.e. the result has fp32-level accuracy (it is NOT bitwise the
// fp32 MFMA result: the summation order inside an MFMA differs).  NP = 2 keeps 3 products (2^-16),
// NP = 1 is plain bf16 compute on fp32 storage.  Activations stay fp32 in HBM; the split is done by
// the loader between the global load and the LDS write, filters are split once at pack time.
//
// Structure (4 waves, wave tile TM x TN MFMA tiles of 32x32, k-tile = 16):
//   * LDS row = NP planes x 32 B (16 bf16) + 16 B pad -> an odd number of 16-byte slots: ds_write_b128
//     from the loader (rows R, R+2, R+4, R+6 per 8-lane group) and ds_read_b128 of the MFMA fragments
//     (lane (r = lane&31, h = lane>>5) reads k = 8h..8h+7 of row r) are both conflict-free;
//   * two k-tiles of global loads are in flight per workgroup (two register staging sets, each
//     refilled right after it is written to LDS); fragments are double buffered in registers; the k-tile barrier sits between the two halves of a tile's MFMAs so the next tile's
//     fragment reads are covered by this wave's own MFMAs;
//   * epilogue, XCD-aware tile order, split/dual outputs: shared with the fp32 kernel (conv_common.h).
#include <type_traits>

#include "conv_common.h"
#include "conv_x3_epi.h"

namespace {

using gvconv::ConvArgs;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KT = 16;                      // k-tile depth

// (plane of A, plane of B) of the t-th product, small terms first
__host__ __device__ constexpr int prod_count(int np) { return np == 3 ? 6 : (np == 2 ? 3 : 1); }
__host__ __device__ constexpr int prod_pa(int np, int t) {
    return np == 3 ? (t == 0 ? 2 : (t == 2 || t == 3 ? 1 : 0)) : (np == 2 ? (t == 0 ? 1 : 0) : 0);
}
__host__ __device__ constexpr int prod_pb(int np, int t) {
    return np == 3 ? (t == 1 ? 2 : (t == 2 || t == 4 ? 1 : 0)) : (np == 2 ? (t == 1 ? 1 : 0) : 0);
}

// 8 consecutive fp32 -> NP x (8 bf16 packed in 16 bytes)
template <int NP>
__device__ __forceinline__ void split8(f32x4 lo, f32x4 hi, bool ok, u32x4 (&out)[NP]) {
    float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = ok ? x[j] : 0.f;          // zero padding taps (loads are unconditional)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bf16x2 pr = {(__bf16)x[2 * q], (__bf16)x[2 * q + 1]};
            out[p][q] = __builtin_bit_cast(unsigned, pr);
            if (p + 1 < NP) {
                x[2 * q] -= (float)pr[0];
                x[2 * q + 1] -= (float)pr[1];
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, int NP, bool GENERIC>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemm_bf16s(const ConvArgs a) {
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    constexpr int NT = WM * WN * 64;                 // threads
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int RB = NP * 32 + 16;                 // LDS row bytes
    constexpr int A_SLOTS = (BM * 2 + NT - 1) / NT;  // (row, half) slots: 8 fp32 each
    constexpr int B_SLOTS = (BN * 2 + NT - 1) / NT;
    constexpr int NMF = TM * TN * prod_count(NP);              // MFMAs per k-tile
    constexpr int HALF = (NMF + 1) / 2;

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

    // slot -> (row, half): an 8-lane ds_write group covers rows R, R+2, R+4, R+6 x both halves
    auto slot_row = [](int idx) -> int {
        const int g = idx >> 3;
        return (g >> 1) * 8 + (g & 1) + 2 * ((idx >> 1) & 3);
    };

    int a_img[A_SLOTS], a_iy0[A_SLOTS], a_ix0[A_SLOTS];
    const int ohow = a.oh * a.ow;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int idx = tid + i * NT;
        const int row = slot_row(idx);
        const int m = m0 + row;
        if (row < BM && m < a.M) {
            const int n = m / ohow;
            const int rem = m - n * ohow;
            const int oy = rem / a.ow;
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
    const char* b_ptr[B_SLOTS];
    bool b_ok[B_SLOTS];
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
        const int idx = tid + i * NT;
        const int row = slot_row(idx);
        const int n = n0 + row;
        b_ok[i] = (row < BN);
        // packed filter: [n][k-tile][plane][16 bf16].  Rows past cout re-read the last filter: their
        // accumulator columns are never stored, so no zeroing (and no divergent branch) is needed.
        const int nc = n < a.cout ? n : a.cout - 1;
        b_ptr[i] = (const char*)a.w + (size_t)nc * a.ktiles * (NP * 32) + 16 * (idx & 1);
    }

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    f32x4 ra[2][A_SLOTS][2];                          // tile t is staged in register set t & 1
    u32x4 rb[2][B_SLOTS][NP];
    bool aok[2][A_SLOTS];
    int fr = 0, fs = 0, fc = 0;                       // filter tap / channel base of the NEXT tile to load
    unsigned a_off[A_SLOTS];                          // element offset of the slot's 8 values (0 when the tap is padding)
    bool a_ok[A_SLOTS];
    auto locate = [&]() {                             // (fr, fs, fc) -> per-slot offset / validity
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int half = (tid + i * NT) & 1;
            const int iyn = a_iy0[i] + fr;
            const int ixn = a_ix0[i] + fs;
            // zero-dilated input (data gradient of a strided conv): only positions that are multiples of
            // the dilation exist, at index >> dil_shift
            const int dmask = (1 << a.dil_shift) - 1;
            const int iy = iyn >> a.dil_shift;
            const int ix = ixn >> a.dil_shift;
            const bool ok = iyn >= 0 && ixn >= 0 && ((iyn | ixn) & dmask) == 0 && iy < a.ih && ix < a.iw && fr < a.kh;
            a_ok[i] = ok;
            a_off[i] = ok ? ((unsigned)(a_img[i] + iy) * (unsigned)a.iw + (unsigned)ix) * (unsigned)a.x_ld +
                                (unsigned)(fc + 8 * half)
                          : 0u;
        }
    };
    if constexpr (!GENERIC) locate();

    auto load_tile = [&](auto rsc, int kt) {
        constexpr int RS = decltype(rsc)::value;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int half = (tid + i * NT) & 1;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!GENERIC) {
                // unconditional loads (padding taps and rows past M read offset 0.. and are zeroed later).  No
                // divergent branch => the compiler can count vmcnt and leave the younger tile's loads in flight.
                aok[RS][i] = a_ok[i];
                const float* p = a.x + a_off[i];
                if (!(a.dbg & 16)) {                          // (timing ablation 16: no A loads)
                    v0 = *reinterpret_cast<const f32x4*>(p);
                    v1 = *reinterpret_cast<const f32x4*>(p + 4);
                }
            } else {
                aok[RS][i] = true;
                if (kt < a.ktiles)                            // (uniform) no gather work for prefetches past the end
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = kt * KT + 8 * half + j;
                    const int2 t = ktab[k];                   // same k across the rows of a tile: LDS broadcast
                    const int iy = a_iy0[i] + (t.y >> 16);
                    const int ix = a_ix0[i] + (t.y & 0xffff);
                    float e = 0.f;
                    if ((unsigned)iy < (unsigned)a.ih && (unsigned)ix < (unsigned)a.iw) e = a.x[a_base[i] + t.x];
                    if (j < 4) v0[j] = e; else v1[j - 4] = e;
                }
            }
            ra[RS][i][0] = v0;
            ra[RS][i][1] = v1;
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int ktc = kt < a.ktiles ? kt : a.ktiles - 1;      // prefetch past the end re-reads the last tile
                rb[RS][i][p] = *reinterpret_cast<const u32x4*>(b_ptr[i] + (size_t)ktc * (NP * 32) + p * 32);
            }
        }
    };
    auto advance_tap = [&]() {
        if constexpr (!GENERIC) {
            fc += KT;
            if (fc >= a.cin) {                        // next filter tap: new window position
                fc = 0;
                if (++fs == a.kw) { fs = 0; ++fr; }
                locate();
            } else {                                  // same tap, next 16 channels
#pragma unroll
                for (int i = 0; i < A_SLOTS; ++i) a_off[i] += a_ok[i] ? (unsigned)KT : 0u;
            }
        }
    };
    auto store_tile = [&](auto rsc, int buf) {
        constexpr int RS = decltype(rsc)::value;
        if (a.dbg & 32) {                                      // timing ablation 32: no LDS writes (operands stay live)
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) asm volatile("" ::"v"(ra[RS][i][0]), "v"(ra[RS][i][1]));
#pragma unroll
            for (int i = 0; i < B_SLOTS; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(rb[RS][i][p]));
            return;
        }
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int idx = tid + i * NT;
            if (A_SLOTS * NT == BM * 2 || idx < BM * 2) {
                u32x4 pl[NP];
                if (a.dbg & 8) {                              // timing ablation: no plane split (wrong values)
#pragma unroll
                    for (int p = 0; p < NP; ++p) pl[p] = __builtin_bit_cast(u32x4, ra[RS][i][p & 1]);
                } else {
                    split8<NP>(ra[RS][i][0], ra[RS][i][1], aok[RS][i], pl);
                }
                char* dst = sA + buf * BM * RB + slot_row(idx) * RB + 16 * (idx & 1);
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(dst + p * 32) = pl[p];
            }
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i) {
            const int idx = tid + i * NT;
            if (B_SLOTS * NT == BN * 2 || b_ok[i]) {
                char* dst = sB + buf * BN * RB + slot_row(idx) * RB + 16 * (idx & 1);
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(dst + p * 32) = rb[RS][i][p];
            }
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

    u32x4 fa[2][TM][NP], fb[2][TN][NP];
    auto read_frags = [&](auto setc, int buf) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                fa[S][i][p] = *reinterpret_cast<const u32x4*>(a_frag + buf * BM * RB + i * 32 * RB + p * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                fb[S][j][p] = *reinterpret_cast<const u32x4*>(b_frag + buf * BN * RB + j * 32 * RB + p * 32);
    };
    // MFMAs [lo, hi) of the flat list (product-major: small terms of every tile first)
    auto mfma_range = [&](auto setc, auto loc, auto hic) {
        constexpr int S = decltype(setc)::value;
        constexpr int LO = decltype(loc)::value, HI = decltype(hic)::value;
#pragma unroll
        for (int t = LO; t < HI; ++t) {
            const int pr = t / (TM * TN);
            const int ij = t % (TM * TN);
            const int i = ij / TN, j = ij % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, fa[S][i][prod_pa(NP, pr)]), __builtin_bit_cast(bf16x8, fb[S][j][prod_pb(NP, pr)]),
                acc[i][j], 0, 0, 0);
        }
    };
    using IH = std::integral_constant<int, HALF>;
    using IN = std::integral_constant<int, NMF>;

    // one k-tile with a successor: first half of its MFMAs, hand tile kt+1 to LDS, barrier, fetch the
    // successor's fragments into the other register set, second half of the MFMAs
    auto step = [&](auto setc, int kt) {
        constexpr int S = decltype(setc)::value;
        using IS = std::integral_constant<int, S>;
        using IO = std::integral_constant<int, S ^ 1>;
        const int buf = kt & 1;
        mfma_range(IS{}, I0{}, IH{});
        store_tile(IO{}, buf ^ 1);                             // register set S^1 holds tile kt+1
        advance_tap();
        load_tile(IO{}, kt + 3);                               // refill it (clamped past the end); tile kt+2 stays in flight in set S
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
    __syncthreads();
    read_frags(I0{}, 0);
    int kt = 0;
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

    if (a.y_p3 | a.y2_p3) {                           // a three-plane destination: the LDS-staged epilogue
        __syncthreads();                              // every wave is done with the main-loop buffers
        x3_epilogue_staged<TM, TN>(a, acc, m0, n0, wm, wn, lane,
                                   reinterpret_cast<float*>(smem_raw + wave * X3EpiGeom<TN>::BYTES));
    } else {
        gvconv::conv_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Halo-tiled 3x3 / stride-1 convolution for the few-channel stem layers on fp32 storage (cin = 32, cout <= 64:
// Conv2d_2a_3x3 / Conv2d_2b_3x3), GV_MATH_BF16X3.  The implicit-GEMM kernel above fetches AND SPLITS every input
// element once per filter tap (9x); with 32 or 64 output channels that split costs as many VALU cycles as the
// MFMAs it feeds, and the LDS-DMA kernel (conv_dma.hip) would move 9x the input through L2 -> LDS for 32 columns of
// arithmetic.  Here a workgroup owns a 30-pixel-wide column strip of one image (a 32-pixel halo row: exactly two
// 8-channel loader chunks per thread) and 32 output channels and walks down the strip 4 output rows (one per wave) at a
// time over a ROLLING RING of 10 halo rows in LDS:
//   * every input row is fetched and split into its three bf16 planes ONCE per strip (208-byte pixels: conflict-free
//     ds_read_b128 fragments at any tap shift); the 4 new rows of the next tile are split and written into the ring
//     slots the previous tile released WHILE this tile's MFMAs run (same instruction stream, no staging phase), and
//     the 4 rows after those are fetched into registers under the rest of the tile: one barrier per tile;
//   * every tap's A fragments are ds_read_b128 at a shifted pixel of the ring, one k-step ahead of their MFMAs;
//   * the packed filter ([n][k-tile][plane][16], 55 KB for 32 columns) is LDS resident for the life of the workgroup.
// Accumulation order (tap, channel half, plane product) is that of the implicit-GEMM kernels: bitwise the same result.
// RW output rows per wave: 1 = 30-pixel strips (32-pixel halo rows: exactly 2 loader chunks per thread, 4 rows per tile),
// 2 = 16-pixel strips (18-pixel halo rows, an MFMA block = 2 output rows x 16 pixels, 8 rows per tile, 18-row ring):
// the narrow form where it saves >= 20 % of the MFMA rows on the image width (narrow images); at 109 columns (4 x 32 = 128
// rows against 7 x 16 = 112) the two measured equal — the chip holds a higher clock with fewer MFMAs per second.
constexpr int halo_tw(int rw) { return rw == 1 ? 30 : 16; }
// ABL: timing ablations (gv_conv2d_set_debug): 4 no epilogue, 16 no fetch, 32 no split / ring stores.
// PLAIN: 32 | cout, 16-byte aligned rows, no residual, output < 4 GiB: the epilogue is branch-free (buffer stores drop
// the lanes past the strip / image), so it too issues between the MFMAs; otherwise the general epilogue.
template <int ABL, bool PLAIN, int RW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_halo_x3(const ConvArgs a) {
    constexpr int TH = 4 * RW, TW = halo_tw(RW), HW = RW == 1 ? 32 : 18, PB = 3 * 64 + 16;  // halo pixel: 3 planes x 32 ch + pad
    constexpr int R = 2 * TH + 2;                                   // ring rows: TH + 2 in use + TH being written
    constexpr int ROWB = RW == 1 ? HW * PB : (HW * PB + 255) / 256 * 256;   // RW 2: a fragment spans two ring rows, whose
                                                                            // banks interleave when the pitch is 0 mod 256 B
    constexpr int NCH = TH * HW * 4;                                // 8-channel chunks of the TH new rows of a tile
    constexpr int SL = (NCH + 255) / 256;                           // per thread: 2 (RW 1), 3 with the last one 1/4 full (RW 2)
    constexpr int DUMP = RW == 1 ? 0 : 64 * 16 * 3;                 // where the idle lanes of the last chunk store
    constexpr int WB = 18 * 96 + 16;                                // LDS filter row: 18 k-tiles x 3 planes
    constexpr int SW = 32 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* sH = smem_raw;                                                          // [R][HW][PB]
    float* stage = reinterpret_cast<float*>(smem_raw + R * ROWB) + (threadIdx.x >> 6) * (32 * SW);
    char* sW = smem_raw + R * ROWB + 4 * 32 * SW * 4;                             // [32][WB]
    char* sDump = sW + 32 * WB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tiles_x = (a.ow + TW - 1) / TW;
    const int nct = (a.cout + 31) / 32;
    // column tile fastest, and consecutive LOGICAL ids on one XCD (hardware deals consecutive workgroup ids round-robin over
    // the 8 XCDs): the two column tiles of a strip then share an L2 and the second one's halo rows hit it
    const int lid = gv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int ct = lid % nct;
    const int strip = lid / nct;
    const int n = strip / tiles_x;
    const int ox0 = (strip % tiles_x) * TW;
    const int co0 = ct * 32;

    for (int idx = tid; idx < 32 * 108; idx += 256) {      // 108 chunks of 16 B per packed filter row
        const int row = idx / 108, ch = idx - row * 108;
        const int col = min(co0 + row, a.cout - 1);
        *reinterpret_cast<u32x4*>(sW + row * WB + ch * 16) =
            *reinterpret_cast<const u32x4*>((const char*)a.w + (size_t)col * a.ktiles * 96 + ch * 16);
    }

    const int rrow = lane >> 3, col4 = (lane & 7) * 4;      // read-back layout of the epilogue: 8 lanes per 32 columns
    const int colg = co0 + col4;                            // (a store instruction writes whole 128-byte lines)
    const int nvalid = min(4, a.cout - colg);
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = min(colg + e, a.cout - 1);
        sc[e] = a.scale[c];
        sh[e] = a.shift[c];
    }
    const bool vec = (a.y_ld % 4 == 0) && ((((uintptr_t)a.y) & 15) == 0) &&
                     (a.res == nullptr || ((a.res_ld % 4 == 0) && ((((uintptr_t)a.res) & 15) == 0)));
    const int ox_end = min(a.ow, ox0 + TW);

    // loader slots: chunk idx = tid + k*256 of a 4-row group -> (row in group, halo column, 8-channel chunk); every
    // thread owns exactly SL chunks, loads are unconditional (clamped addresses) and padding is zeroed in the split
    int l_hy[SL], l_goff[SL], l_loff[SL];
    bool l_ok[SL];
#pragma unroll
    for (int k = 0; k < SL; ++k) {
        const bool live = tid + k * 256 < NCH;             // (the idle lanes repeat the previous chunk's load)
        const int idx = live ? tid + k * 256 : tid + (k - 1) * 256;
        const int pix = idx >> 2, ch = idx & 3;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int ix = ox0 + hx - a.pad_l;
        l_hy[k] = live ? hy : -1;
        l_ok[k] = (unsigned)ix < (unsigned)a.iw;
        l_goff[k] = min(max(ix, 0), a.iw - 1) * a.x_ld + ch * 8;
        l_loff[k] = live ? hx * PB + ch * 16 : (tid & 63) * 16;
    }
    const float* ximg = a.x + (size_t)n * a.ih * a.iw * a.x_ld;
    const int rowpitch = a.iw * a.x_ld;

    f32x4 hr[SL][2], hn[SL][2];                             // rows being split now / rows in flight for the next tile
    bool hok[SL], hnok[SL];
    // halo rows [h0, h0 + TH) -> registers (halo row h is input row h - pad_t; outside the image: zeros)
    auto fetch_one = [&](int k, int h0, f32x4 (&dst)[SL][2], bool (&dok)[SL]) {
        const int iy = h0 + max(l_hy[k], 0) - a.pad_t;
        const float* p = ximg + (size_t)min(max(iy, 0), a.ih - 1) * rowpitch + l_goff[k];
        dst[k][0] = *reinterpret_cast<const f32x4*>(p);
        dst[k][1] = *reinterpret_cast<const f32x4*>(p + 4);
        dok[k] = l_ok[k] && (unsigned)iy < (unsigned)a.ih;
    };
    // registers -> ring slots of halo rows [h0, h0 + TH): split once into the three planes
    auto put_one = [&](int k, int h0) {
        u32x4 pl[3];
        split8<3>(hr[k][0], hr[k][1], hok[k], pl);
        char* dst = l_hy[k] >= 0 ? sH + ((h0 + l_hy[k] + 2 * R) % R) * ROWB + l_loff[k] : sDump + l_loff[k];
        const int ps = l_hy[k] >= 0 ? 64 : 1024;          // plane p of a pixel at +64*p
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * ps) = pl[p];
    };
    // scale / shift / residual / ReLU and store of read-back pass `pass` (8 output pixels x 32 channels) of row oy
    float rlo[4];                                           // ReLU as max(v, rlo): 0 where it applies, -inf elsewhere
#pragma unroll
    for (int e = 0; e < 4; ++e) rlo[e] = (a.relu && colg + e < a.relu_limit) ? 0.f : -__builtin_inff();
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.y, 0, PLAIN ? (int)(unsigned)((size_t)a.M * a.y_ld * 4) : 0, 0x00020000);
    auto finish_pass = [&](int pass, int oy0, f32x4 v4) {       // oy0: the wave's first output row of that tile
        const int mrow = pass * 8 + rrow;                   // MFMA row -> (output row, column in the strip)
        const int oy = RW == 1 ? oy0 : oy0 + (mrow >> 4);
        const int row = RW == 1 ? mrow : (mrow & 15);
        float v[4] = {v4[0], v4[1], v4[2], v4[3]};
        if constexpr (PLAIN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], rlo[e]);
            const unsigned m = (unsigned)((n * a.oh + oy) * a.ow + ox0 + row);
            const unsigned off = (oy < a.oh && ox0 + row < ox_end) ? (m * (unsigned)a.y_ld + (unsigned)colg) * 4u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), yrs, off, 0, 0);
        } else {
            if (oy >= a.oh || nvalid <= 0 || ox0 + row >= ox_end) return;
            const size_t m = (size_t)(n * a.oh + oy) * a.ow + ox0 + row;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (a.res) {
                const float* rp = a.res + m * a.res_ld + colg;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nvalid) v[e] += rp[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], rlo[e]);
            float* yp = a.y + m * a.y_ld + colg;
            if (vec && nvalid == 4) {
                *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nvalid) yp[e] = v[e];
            }
        }
    };

    const int ntiles = (a.oh + TH - 1) / TH;
#pragma unroll
    for (int g = 0; g < 2; ++g) {                           // halo rows 2-TH..1 (those below 0 land in free slots), 2..TH+1
#pragma unroll
        for (int k = 0; k < SL; ++k) fetch_one(k, TH * g + 2 - TH, hr, hok);
#pragma unroll
        for (int k = 0; k < SL; ++k) put_one(k, TH * g + 2 - TH);
    }
#pragma unroll
    for (int k = 0; k < SL; ++k) fetch_one(k, TH + 2, hr, hok);   // split and stored under tile 0

    const int a_lane = (RW == 1 ? li : (li & 15)) * PB + 16 * lh;
    const int a_row = RW == 1 ? wave : 2 * wave + (li >> 4);    // this lane's output row within the tile
    const char* b_lane = sW + li * WB + 16 * lh;
    float accv[16];                                         // the previous tile's accumulators: stored under this tile
#pragma unroll
    for (int r = 0; r < 16; ++r) accv[r] = 0.f;
    int rbase = 0;                                          // (TH t) % R
    for (int t = 0; t < ntiles; ++t, rbase = rbase + TH >= R ? rbase + TH - R : rbase + TH) {
        __syncthreads();                                    // ring rows of this tile written; previous tile's reads done
        const int oy = t * TH + RW * wave;
        const int hput = TH * t + TH + 2;                   // rows past the last tile's need go to released slots: harmless
        const char* ar[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int sl = rbase + a_row + r;               // (oy + r) % R without a division
            ar[r] = sH + (sl >= R ? sl - R : sl) * ROWB + a_lane;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        u32x4 fa[2][3], fb[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            fa[0][p] = *reinterpret_cast<const u32x4*>(ar[0] + p * 64);
            fb[0][p] = *reinterpret_cast<const u32x4*>(b_lane + p * 32);
        }
        // One wave per SIMD has nobody to hide anything behind, so everything but the MFMAs is dealt out in pieces, one
        // per k-step, each fenced into its k-step (sched_barrier) to issue in the shadow of that step's six MFMAs:
        //   all steps   the fragment reads of k-step ks+1, in FRONT of the MFMAs of k-step ks
        //   0 .. 5      the PREVIOUS tile's epilogue: accumulators -> per-wave LDS block -> 4 x (8 pixels x 128 bytes)
        //   0 .. 9      split (one element pair per step) and ring stores of the 4 rows the next tile needs
        //   6, 7        fetch of the 4 halo rows the tile after next needs: AFTER the stores, because vmcnt retires in
        //               issue order and the wait for these loads (next tile's first split) must not wait on stores
        float px[SL][8];
        u32x4 ppl[SL][3];
        f32x4 sv[2];
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {                   // k-step = (tap, channel half)
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 18) {
                const int t1 = (ks + 1) >> 1, c1 = (ks + 1) & 1;
                const int r1 = t1 / 3, s1 = t1 - r1 * 3;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[nxt][p] = *reinterpret_cast<const u32x4*>(ar[r1] + s1 * PB + p * 64 + c1 * 32);
                    fb[nxt][p] = *reinterpret_cast<const u32x4*>(b_lane + (ks + 1) * 96 + p * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cur][prod_pa(3, q)]),
                                                              __builtin_bit_cast(bf16x8, fb[cur][prod_pb(3, q)]), acc, 0, 0, 0);
            if (ks >= 6 && ks < 6 + SL && !(ABL & 16)) fetch_one(ks - 6, hput + TH, hn, hnok);
            if (ks < SL * 5 && !(ABL & 32)) {               // unit u = ks / 5: pieces 0..3 split a pair each, piece 4 stores
                const int u = ks / 5, pc = ks - u * 5;
                if (pc < 4) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v = pc < 2 ? hr[u][0][2 * pc + e] : hr[u][1][2 * (pc - 2) + e];
                        px[u][2 * pc + e] = hok[u] ? v : 0.f;
                    }
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const bf16x2 pr = {(__bf16)px[u][2 * pc], (__bf16)px[u][2 * pc + 1]};
                        unsigned word = __builtin_bit_cast(unsigned, pr);
                        asm volatile("" : "+v"(word));     // pins this piece into its k-step (else it sinks to the store)
                        ppl[u][p][pc] = word;
                        if (p < 2) {
                            px[u][2 * pc] -= (float)pr[0];
                            px[u][2 * pc + 1] -= (float)pr[1];
                        }
                    }
                } else {
                    const int sl = rbase + TH + 2 + l_hy[u];        // (hput + l_hy) % R
                    char* dst = l_hy[u] >= 0 ? sH + (sl >= R ? sl - R : sl) * ROWB + l_loff[u] : sDump + l_loff[u];
                    const int ps = l_hy[u] >= 0 ? 64 : 1024;
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * ps) = ppl[u][p];
                }
            }
            if (!(ABL & 4) && t > 0) {
                if (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) stage[(4 * lh + (r & 3) + 8 * (r >> 2)) * SW + li] = accv[r];
                }
                if (ks >= 1 && ks < 5) {                    // read back pass ks-1 now, finish it one k-step later:
                    if (ks == 1) __builtin_amdgcn_wave_barrier();      // its wait then passes over the younger fragment reads
                    sv[ks & 1] = *reinterpret_cast<const f32x4*>(stage + ((ks - 1) * 8 + rrow) * SW + col4);
                }
                if (ks >= 2 && ks < 6) finish_pass(ks - 2, oy - TH, sv[(ks - 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL & 4) {
            float tsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) tsum += acc[r];
            if (tsum == 1.2345e-30f) a.y[0] = tsum;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[r] = acc[r];
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            hr[k][0] = hn[k][0];
            hr[k][1] = hn[k][1];
            hok[k] = hnok[k];
        }
    }
    if (!(ABL & 4)) {                                       // the last tile's epilogue
        const int oy = (ntiles - 1) * TH + RW * wave;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[(4 * lh + (r & 3) + 8 * (r >> 2)) * SW + li] = accv[r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
            finish_pass(pass, oy, *reinterpret_cast<const f32x4*>(stage + (pass * 8 + rrow) * SW + col4));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x32_bf16 (one k-step = one filter tap: 32 channels): the 32 x 32 wave tile is
// 2 x 2 blocks of 16 x 16, the LDS images keep a plane's 32 channels contiguous (64 bytes; halo pixels 192 bytes with
// the 8-channel chunks XOR-swizzled by (column >> 1) & 3, filter rows [tap][plane][32] at a 1760-byte pitch: both
// conflict-free for the 16-lane groups of ds_read_b128 at every tap shift).  Same bytes read from LDS per FLOP and the
// same cycles per FLOP as the 32x32x16 form; the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS (7)).
// POOL (GV_CONV_MAXPOOL3S2 on fp32 storage; Conv2d_2b_3x3 -> MaxPool_3a_3x3, nets/inception_v3.py:111-113): the 3x3 / 2
// VALID max pool of the output is taken inside the workgroup and only the pooled tensor is written — a quarter of the
// bytes, and the pool's launch (which re-read all of them from HBM: the largest tensor of the fp32 plan) disappears.  A
// strip advances 28 columns = 14 pooled ones (30 are computed as before: the same number of strips for 109 columns); a
// finished row (BatchNorm + ReLU applied) is pooled HORIZONTALLY on its way out of the wave's staging block into a ten-row
// LDS ring of [14 pooled pixels][32 channels], and two tiles later — behind the tile loop's own barrier, no extra one —
// 224 threads take the two pooled rows a tile completed: bitwise the two launches' values.  (Whole rows in a six-row ring
// with a barrier in the middle of the tile cost the convolution +15 %.)
template <int ABL, bool PLAIN, bool POOL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_halo_x3_k32(const ConvArgs a) {
    constexpr int TH = 4, TW = 30, HW = 32, PB = 192;
    constexpr int R = 2 * TH + 2;
    constexpr int ROWB = HW * PB;
    constexpr int SL = 2;
    constexpr int WB = 9 * 192 + 32;
    constexpr int SW = 32 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* sH = smem_raw;                                                          // [R][HW][plane][32 ch]
    float* stage = reinterpret_cast<float*>(smem_raw + R * ROWB) + (threadIdx.x >> 6) * (32 * SW);
    char* sW = smem_raw + R * ROWB + 4 * 32 * SW * 4;                             // [32][WB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
    constexpr int PSTEP = POOL ? 28 : TW, PPX = 14;
    const int tiles_x = POOL ? (a.pw + PPX - 1) / PPX : (a.ow + TW - 1) / TW;
    const int nct = (a.cout + 31) / 32;
    const int lid = gv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int ct = lid % nct;
    const int strip = lid / nct;
    const int n = strip / tiles_x;
    const int ox0 = (strip % tiles_x) * PSTEP;
    float* pimg = reinterpret_cast<float*>(smem_raw + R * ROWB + 4 * 32 * SW * 4 + 32 * WB);   // POOL: [10 rows][14 pooled px][32 ch]
    const int co0 = ct * 32;

    for (int idx = tid; idx < 32 * 108; idx += 256) {      // packed [n][k-tile 16][plane][16] -> LDS [n][tap][plane][32]
        const int row = idx / 108, ch = idx - row * 108;
        const int col = min(co0 + row, a.cout - 1);
        const int kt = ch / 6, rem = ch - kt * 6;
        *reinterpret_cast<u32x4*>(sW + row * WB + (kt >> 1) * 192 + (rem >> 1) * 64 + (kt & 1) * 32 + (rem & 1) * 16) =
            *reinterpret_cast<const u32x4*>((const char*)a.w + (size_t)col * a.ktiles * 96 + ch * 16);
    }

    const int rrow = lane >> 3, col4 = (lane & 7) * 4;
    const int colg = co0 + col4;
    const int nvalid = min(4, a.cout - colg);
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = min(colg + e, a.cout - 1);
        sc[e] = a.scale[c];
        sh[e] = a.shift[c];
    }
    float scp[2] = {0.f, 0.f}, shp[2] = {0.f, 0.f}, rlop[2] = {0.f, 0.f};     // POOL: this lane's two channels of the accumulator layout
    if constexpr (POOL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = min(co0 + 16 * j + l16, a.cout - 1);
            scp[j] = a.scale[c];
            shp[j] = a.shift[c];
            rlop[j] = (a.relu && c < a.relu_limit) ? 0.f : -__builtin_inff();
        }
    }
    const bool vec = (a.y_ld % 4 == 0) && ((((uintptr_t)a.y) & 15) == 0) &&
                     (a.res == nullptr || ((a.res_ld % 4 == 0) && ((((uintptr_t)a.res) & 15) == 0)));
    const int ox_end = min(a.ow, ox0 + TW);

    int l_hy[SL], l_goff[SL], l_loff[SL];
    bool l_ok[SL];
#pragma unroll
    for (int k = 0; k < SL; ++k) {
        const int idx = tid + k * 256;
        const int pix = idx >> 2, ch = idx & 3;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int ix = ox0 + hx - a.pad_l;
        l_hy[k] = hy;
        l_ok[k] = (unsigned)ix < (unsigned)a.iw;
        l_goff[k] = min(max(ix, 0), a.iw - 1) * a.x_ld + ch * 8;
        l_loff[k] = hx * PB + ((ch ^ ((hx >> 1) & 3)) * 16);
    }
    const float* ximg = a.x + (size_t)n * a.ih * a.iw * a.x_ld;
    const int rowpitch = a.iw * a.x_ld;

    f32x4 hr[SL][2], hn[SL][2];
    bool hok[SL], hnok[SL];
    auto fetch_one = [&](int k, int h0, f32x4 (&dst)[SL][2], bool (&dok)[SL]) {
        const int iy = h0 + l_hy[k] - a.pad_t;
        const float* p = ximg + (size_t)min(max(iy, 0), a.ih - 1) * rowpitch + l_goff[k];
        dst[k][0] = *reinterpret_cast<const f32x4*>(p);
        dst[k][1] = *reinterpret_cast<const f32x4*>(p + 4);
        dok[k] = l_ok[k] && (unsigned)iy < (unsigned)a.ih;
    };
    auto put_one = [&](int k, int h0) {
        u32x4 pl[3];
        split8<3>(hr[k][0], hr[k][1], hok[k], pl);
        char* dst = sH + ((h0 + l_hy[k] + 2 * R) % R) * ROWB + l_loff[k];
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * 64) = pl[p];
    };
    float rlo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rlo[e] = (a.relu && colg + e < a.relu_limit) ? 0.f : -__builtin_inff();
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.y, 0, PLAIN ? (int)(unsigned)((size_t)a.M * a.y_ld * 4) : 0, 0x00020000);
    auto finish_pass = [&](int pass, int oy, f32x4 v4) {
        const int row = pass * 8 + rrow;
        float v[4] = {v4[0], v4[1], v4[2], v4[3]};
        if constexpr (PLAIN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], rlo[e]);
            const unsigned m = (unsigned)((n * a.oh + oy) * a.ow + ox0 + row);
            const unsigned off = (oy < a.oh && ox0 + row < ox_end) ? (m * (unsigned)a.y_ld + (unsigned)colg) * 4u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), yrs, off, 0, 0);
        } else {
            if (oy >= a.oh || nvalid <= 0 || ox0 + row >= ox_end) return;
            const size_t m = (size_t)(n * a.oh + oy) * a.ow + ox0 + row;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (a.res) {
                const float* rp = a.res + m * a.res_ld + colg;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nvalid) v[e] += rp[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], rlo[e]);
            float* yp = a.y + m * a.y_ld + colg;
            if (vec && nvalid == 4) {
                *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nvalid) yp[e] = v[e];
            }
        }
    };

    // POOL: half h of the wave's row oy (7 pooled pixels x 8 four-channel groups per half): max over pixels 2p .. 2p+2 of the
    // finished values (BatchNorm + ReLU went in with the accumulators) from the wave's staging block into the ring
    auto hpass = [&](int h, int oy) {
        const int pp = lane >> 3;
        if (pp >= 7) return;
        const int ppx = h * 7 + pp;
        // (no masks: a pooled pixel inside the pooled map only reads convolution pixels inside the convolution's map)
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + (2 * ppx) * SW + col4);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + (2 * ppx + 1) * SW + col4);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(stage + (2 * ppx + 2) * SW + col4);
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v0[e], v1[e]), v2[e]);
        const int slot = oy - (oy / 10) * 10;
        *reinterpret_cast<f32x4*>(pimg + (slot * PPX + ppx) * 32 + col4) = m;
    };
    // POOL: the pooled rows 2T-1 and 2T, complete once tile T's rows are in the ring (14 pooled pixels x 8 four-channel groups each)
    auto pool_rows = [&](int T) {
        if (tid >= 2 * PPX * 8) return;
        const int pyl = tid / (PPX * 8), rem = tid - pyl * (PPX * 8), ppx = rem >> 3, c4 = (rem & 7) * 4;
        const int py = 2 * T - 1 + pyl, px = PPX * (strip % tiles_x) + ppx;
        if (py < 0 || py >= a.ph || px >= a.pw || co0 + c4 >= a.cout) return;
        f32x4 m = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int row = 2 * py + dy, slot = row - (row / 10) * 10;
            const f32x4 v = *reinterpret_cast<const f32x4*>(pimg + (slot * PPX + ppx) * 32 + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
        float* yp = a.y + ((size_t)(n * a.ph + py) * a.pw + px) * a.y_ld + co0 + c4;
        const int nv = min(4, a.cout - co0 - c4);
        if (vec && nv == 4) {
            *reinterpret_cast<f32x4*>(yp) = m;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nv) yp[e] = m[e];
        }
    };
    const int ntiles = (a.oh + TH - 1) / TH;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int k = 0; k < SL; ++k) fetch_one(k, TH * g + 2 - TH, hr, hok);
#pragma unroll
        for (int k = 0; k < SL; ++k) put_one(k, TH * g + 2 - TH);
    }
#pragma unroll
    for (int k = 0; k < SL; ++k) fetch_one(k, TH + 2, hr, hok);

    // fragment addresses: A block i (pixels 16 i .. 16 i + 15 of the wave's row) at tap column s, B block j
    int aoff[3][2];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = l16 + 16 * i + s_;
            aoff[s_][i] = q * PB + ((lg ^ ((q >> 1) & 3)) * 16);
        }
    const char* b_lane[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_lane[j] = sW + (l16 + 16 * j) * WB + lg * 16;

    float accv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) accv[r] = 0.f;
    int rbase = 0;
    for (int t = 0; t < ntiles; ++t, rbase = rbase + TH >= R ? rbase + TH - R : rbase + TH) {
        __syncthreads();
        const int oy = t * TH + wave;
        const int hput = TH * t + TH + 2;
        const char* ar[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int sl = rbase + wave + r;
            ar[r] = sH + (sl >= R ? sl - R : sl) * ROWB;
        }
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 fa[2][3], fb[2][2][3];                         // A: by half-step parity; B: by tap parity, block j
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            fa[0][p] = *reinterpret_cast<const u32x4*>(ar[0] + aoff[0][0] + p * 64);
            fb[0][0][p] = *reinterpret_cast<const u32x4*>(b_lane[0] + p * 64);
            fb[0][1][p] = *reinterpret_cast<const u32x4*>(b_lane[1] + p * 64);
        }
        float px[SL][8];
        u32x4 ppl[SL][3];
        f32x4 sv[2];
#pragma unroll
        for (int hs = 0; hs < 18; ++hs) {                   // half-step = (tap, A block i): 12 MFMAs of 16 cycles
            const int tap = hs >> 1, i = hs & 1, tp = tap & 1;
            if (hs + 1 < 18) {
                const int tap1 = (hs + 1) >> 1, i1 = (hs + 1) & 1;
                const int r1 = tap1 / 3, s1 = tap1 - r1 * 3;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fa[i1][p] = *reinterpret_cast<const u32x4*>(ar[r1] + aoff[s1][i1] + p * 64);
            }
            if (tap + 1 < 9) {                              // the next tap's B block i
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[tp ^ 1][i][p] = *reinterpret_cast<const u32x4*>(b_lane[i] + (tap + 1) * 192 + p * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i][prod_pa(3, q)]),
                                                                       __builtin_bit_cast(bf16x8, fb[tp][j][prod_pb(3, q)]),
                                                                       acc[i][j], 0, 0, 0);
            const int ks = hs;
            if (ks >= 6 && ks < 6 + SL && !(ABL & 16)) fetch_one(ks - 6, hput + TH, hn, hnok);
            if (ks < SL * 5 && !(ABL & 32)) {
                const int u = ks / 5, pc = ks - u * 5;
                if (pc < 4) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v = pc < 2 ? hr[u][0][2 * pc + e] : hr[u][1][2 * (pc - 2) + e];
                        px[u][2 * pc + e] = hok[u] ? v : 0.f;
                    }
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const bf16x2 pr = {(__bf16)px[u][2 * pc], (__bf16)px[u][2 * pc + 1]};
                        unsigned word = __builtin_bit_cast(unsigned, pr);
                        asm volatile("" : "+v"(word));
                        ppl[u][p][pc] = word;
                        if (p < 2) {
                            px[u][2 * pc] -= (float)pr[0];
                            px[u][2 * pc + 1] -= (float)pr[1];
                        }
                    }
                } else {
                    const int sl = rbase + TH + 2 + l_hy[u];
                    char* dst = sH + (sl >= R ? sl - R : sl) * ROWB + l_loff[u];
#pragma unroll
                    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * 64) = ppl[u][p];
                }
            }
            if (!(ABL & 4) && t > 0) {
                if (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {          // acc[i][j][v]: pixel 16 i + 4 lg + v, channel 16 j + l16
                        const int jc = (r >> 2) & 1;
                        stage[(16 * (r >> 3) + 4 * lg + (r & 3)) * SW + 16 * jc + l16] =
                            POOL ? fmaxf(accv[r] * scp[jc] + shp[jc], rlop[jc]) : accv[r];
                    }
                }
                if constexpr (POOL) {
                    if (ks == 1) {
                        __builtin_amdgcn_wave_barrier();
                        if (t >= 2) pool_rows(t - 2);       // tile t-2's rows: in the ring since the barrier at this tile's top
                    }
                    if (ks == 2 || ks == 3) hpass(ks - 2, oy - TH);
                } else {
                    if (ks >= 1 && ks < 5) {
                        if (ks == 1) __builtin_amdgcn_wave_barrier();
                        sv[ks & 1] = *reinterpret_cast<const f32x4*>(stage + ((ks - 1) * 8 + rrow) * SW + col4);
                    }
                    if (ks >= 2 && ks < 6) finish_pass(ks - 2, oy - TH, sv[(ks - 1) & 1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL & 4) {
            float tsum = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) tsum += acc[i][j][v];
            if (tsum == 1.2345e-30f) a.y[0] = tsum;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) accv[r] = acc[r >> 3][(r >> 2) & 1][r & 3];
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            hr[k][0] = hn[k][0];
            hr[k][1] = hn[k][1];
            hok[k] = hnok[k];
        }
    }
    if (!(ABL & 4)) {
        const int oy = (ntiles - 1) * TH + wave;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jc = (r >> 2) & 1;
            stage[(16 * (r >> 3) + 4 * lg + (r & 3)) * SW + 16 * jc + l16] = POOL ? fmaxf(accv[r] * scp[jc] + shp[jc], rlop[jc]) : accv[r];
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (POOL) {
            // (a wave still in the last iteration's pool_rows reads ring slots this tile's rows are about to take: in the
            // loop the next iteration's barrier stands between the two)
            __syncthreads();
            hpass(0, oy);
            hpass(1, oy);
            __syncthreads();
            if (ntiles >= 2) pool_rows(ntiles - 2);
            pool_rows(ntiles - 1);
        } else {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass)
                finish_pass(pass, oy, *reinterpret_cast<const f32x4*>(stage + (pass * 8 + rrow) * SW + col4));
        }
    }
}

int launch_halo_x3(const ConvArgs& a, hipStream_t st) {
    // strips of 30 (one output row per wave) or 16 pixels (two): whichever covers the width with fewer MFMA rows
    // (measured equal at 12 % fewer rows — Conv2d_2a/2b, 109 columns — so the narrow form needs a 20 % saving)
    const int rw = gv_ceil_div(a.ow, 16) * 16 * 5 <= gv_ceil_div(a.ow, 30) * 32 * 4 ? 2 : 1;
    const int tw = halo_tw(rw), hw = rw == 1 ? 32 : 18, ring = 8 * rw + 2;
    const int tiles_x = (a.ow + tw - 1) / tw, nct = (a.cout + 31) / 32;
    const size_t lds = (size_t)ring * (rw == 1 ? hw * 208 : (hw * 208 + 255) / 256 * 256) + 4 * 32 * 36 * 4 + 32 * (18 * 96 + 16) +
                       (rw == 1 ? 0 : 64 * 16 * 3);
    const dim3 grid((unsigned)(a.nb * tiles_x * nct));
    const bool plain = a.cout % 32 == 0 && a.y_ld % 4 == 0 && ((((uintptr_t)a.y) & 15) == 0) && a.res == nullptr &&
                       (uint64_t)a.M * a.y_ld * 4 < 0xffffffffull;
#define GV_HALO_LAUNCH(B, P, W)                                                                                     \
    {                                                                                                               \
        const bool ok = GV_BIG_LDS_OK((&conv3x3_halo_x3<B, P, W>), 160 * 1024);                                   \
        if (!ok) return GV_E_UNSUPPORTED;                                                                           \
        hipLaunchKernelGGL((conv3x3_halo_x3<B, P, W>), grid, dim3(256), lds, st, a);                                \
        GV_LAUNCH_CHECK();                                                                                          \
        return GV_OK;                                                                                               \
    }
#define GV_HALO_K32(B, P)                                                                                           \
    {                                                                                                               \
        const bool ok = GV_BIG_LDS_OK((&conv3x3_halo_x3_k32<B, P>), 160 * 1024);                                   \
        if (!ok) return GV_E_UNSUPPORTED;                                                                           \
        hipLaunchKernelGGL((conv3x3_halo_x3_k32<B, P>), grid, dim3(256), lds32, st, a);                             \
        GV_LAUNCH_CHECK();                                                                                          \
        return GV_OK;                                                                                               \
    }
    const size_t lds32 = (size_t)10 * 32 * 192 + 4 * 32 * 36 * 4 + 32 * (9 * 192 + 32);
    if (a.pool) {                                     // conv -> max pool 3x3 / 2 VALID: the 30-pixel strip form only
        if (rw != 1 || a.pool != 1 || a.res != nullptr || a.cout % 4 != 0) return GV_E_UNSUPPORTED;
        const dim3 pgrid((unsigned)(a.nb * ((a.pw + 13) / 14) * nct));
        const size_t ldsp = lds32 + (size_t)10 * 14 * 32 * 4;
        const bool ok = GV_BIG_LDS_OK((&conv3x3_halo_x3_k32<0, false, true>), 160 * 1024);
        if (!ok) return GV_E_UNSUPPORTED;
        hipLaunchKernelGGL((conv3x3_halo_x3_k32<0, false, true>), pgrid, dim3(256), ldsp, st, a);
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    if (rw == 1 && !(a.dbg & 8)) {                    // (debug bit 8: the 32x32x16 form, for A/B timing)
        if (a.dbg & (4 | 16 | 32)) {
            if (!plain) return GV_E_UNSUPPORTED;
            switch (a.dbg & 52) {
                case 4: GV_HALO_K32(4, true)
                case 52: GV_HALO_K32(52, true)
            }
            return GV_E_UNSUPPORTED;
        }
        if (plain) GV_HALO_K32(0, true)
        GV_HALO_K32(0, false)
    }
    if (a.dbg & (4 | 16 | 32)) {                      // timing experiments only
        if (!plain || rw != 1) return GV_E_UNSUPPORTED;
        switch (a.dbg & 52) {
            case 4: GV_HALO_LAUNCH(4, true, 1)
            case 48: GV_HALO_LAUNCH(48, true, 1)
            case 52: GV_HALO_LAUNCH(52, true, 1)
        }
        return GV_E_UNSUPPORTED;
    }
    if (rw == 2) {
        if (plain) GV_HALO_LAUNCH(0, true, 2)
        GV_HALO_LAUNCH(0, false, 2)
    }
    if (plain) GV_HALO_LAUNCH(0, true, 1)
    GV_HALO_LAUNCH(0, false, 1)
#undef GV_HALO_LAUNCH
#undef GV_HALO_K32
}

// ---------------------------------------------------------------------------------------------------------------
// The 3-channel stems on fp32 storage (Inception Conv2d_1a 3x3/2, ResNet conv1 7x7/2), GV_MATH_BF16X3: the strip
// kernel of conv_lp.hip (conv_stem_patch_lp) with the input patch split ONCE into its three bf16 planes while it is
// written to LDS and the six plane products per k-step.  k is re-ordered row-wise (filter row r owns KR slots of
// which KW*3 are real) so the 8 values of a fragment are consecutive patch elements: four ds_read_b32 per plane.
template <int TN, int KW>
__global__ __launch_bounds__(256, 2) void conv_stem_patch_x3(const ConvArgs a) {
    constexpr int KR = KW == 3 ? 16 : 24;
    constexpr int NG = (KW * KR / 8 + 1) / 2 * 2;
    constexpr int PR = 3 * 2 + KW + 1, PC = 31 * 2 + KW;
    constexpr int PITCH = (PC * 3 * 2 + 16 + 3) / 4 * 4;    // one plane of one patch row
    constexpr int NEL = (PR - 1) * PC * 3;
    constexpr int SL = (NEL + 255) / 256;
    constexpr int WB = NG * 16 + 16;                        // one plane of one filter row
    constexpr int SW = 32 + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* sP = smem_raw;                                                      // [3][PR][PITCH]
    float* stage = reinterpret_cast<float*>(smem_raw + 3 * PR * PITCH) + (threadIdx.x >> 6) * (32 * SW);
    char* sW = smem_raw + 3 * PR * PITCH + 4 * 32 * SW * 4;                   // [3][32*TN][WB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tiles_x = (a.ow + 31) / 32;
    const int n = blockIdx.x / tiles_x;
    const int ox0 = (blockIdx.x % tiles_x) * 32;
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(a.w);   // [cout][k-tile][plane][16]

    for (int idx = tid; idx < 32 * TN * NG * 8; idx += 256) {               // filter planes in the row-wise k order
        const int row = idx / (NG * 8), kk = idx - row * (NG * 8);
        const int r = kk / KR, j = kk - r * KR;
        const bool ok = row < a.cout && r < KW && j < KW * 3;
        const int k = r * KW * 3 + j;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<unsigned short*>(sW + (p * 32 * TN + row) * WB + kk * 2) =
                ok ? wp[((size_t)row * a.ktiles + k / 16) * 48 + p * 16 + (k & 15)] : (unsigned short)0;
    }
    for (int idx = tid; idx < 3 * PR * PITCH / 4; idx += 256) reinterpret_cast<unsigned*>(sP)[idx] = 0u;

    const int rrow = lane >> 3, col4 = (lane & 7) * 4;      // read-back: 8 lanes x 16 bytes = one pixel's 32 channels, so a
    float sc[TN][4], sh[TN][4];                             // store instruction writes whole 128-byte lines
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = min(j * 32 + col4 + e, a.cout - 1);
            sc[j][e] = a.scale[c];
            sh[j][e] = a.shift[c];
        }
    const bool vec = (a.y_ld % 4 == 0) && ((((uintptr_t)a.y) & 15) == 0);

    float pr_[SL];
    auto fetch = [&](int oy0) {
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            const int idx = tid + k * 256;
            float v = 0.f;
            if (idx < NEL) {
                const int prow = idx / (PC * 3), e = idx - prow * (PC * 3);
                const int px = e / 3, ch = e - px * 3;
                const int iy = oy0 * 2 - a.pad_t + prow, ix = ox0 * 2 - a.pad_l + px;
                if ((unsigned)iy < (unsigned)a.ih && (unsigned)ix < (unsigned)a.iw)
                    v = a.x[((size_t)(n * a.ih + iy) * a.iw + ix) * a.x_ld + ch];
            }
            pr_[k] = v;
        }
    };
    fetch(0);
    for (int oy0 = 0; oy0 < a.oh; oy0 += 4) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SL; ++k) {
            const int idx = tid + k * 256;
            if (idx < NEL) {
                const int prow = idx / (PC * 3), e = idx - prow * (PC * 3);
                float x = pr_[k];
#pragma unroll
                for (int p = 0; p < 3; ++p) {                 // a = a0 + a1 + a2, exact (see the file header)
                    const __bf16 h = (__bf16)x;
                    *reinterpret_cast<unsigned short*>(sP + (p * PR + prow) * PITCH + e * 2) =
                        __builtin_bit_cast(unsigned short, h);
                    x -= (float)h;
                }
            }
        }
        __syncthreads();
        if (oy0 + 4 < a.oh) fetch(oy0 + 4);
        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NG / 2; ++c) {
            const int g = 2 * c + lh;
            const int r = g / (KR / 8), q = g - r * (KR / 8);
            const int prow = r < KW ? 2 * wave + r : PR - 1;
            u32x4 fa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const char* ap = sP + (p * PR + prow) * PITCH + li * 12 + q * 16;
#pragma unroll
                for (int d = 0; d < 4; ++d) fa[p][d] = *reinterpret_cast<const unsigned*>(ap + 4 * d);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                u32x4 fb[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[p] = *reinterpret_cast<const u32x4*>(sW + (p * 32 * TN + j * 32 + li) * WB + g * 16);
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[prod_pa(3, t)]),
                                                                     __builtin_bit_cast(bf16x8, fb[prod_pb(3, t)]),
                                                                     acc[j], 0, 0, 0);
            }
        }
        const int oy = oy0 + wave;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[(4 * lh + (r & 3) + 8 * (r >> 2)) * SW + li] = acc[j][r];
            __builtin_amdgcn_wave_barrier();
            const int colj = j * 32 + col4;
            const int nvalid = min(4, a.cout - colj);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + rrow;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * SW + col4);
                if (oy >= a.oh || nvalid <= 0 || ox0 + row >= a.ow) continue;
                const size_t m = (size_t)(n * a.oh + oy) * a.ow + ox0 + row;
                float v[4] = {lo[0], lo[1], lo[2], lo[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] * sc[j][e] + sh[j][e];
                    if (a.relu && colj + e < a.relu_limit) v[e] = fmaxf(v[e], 0.f);
                }
                float* yp = a.y + m * a.y_ld + colj;
                if (vec && nvalid == 4) {
                    *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (e < nvalid) yp[e] = v[e];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int TN, int KW>
int launch_stem_x3_one(const ConvArgs& a, hipStream_t st) {
    constexpr int KR = KW == 3 ? 16 : 24, NG = (KW * KR / 8 + 1) / 2 * 2, PR = 3 * 2 + KW + 1, PC = 31 * 2 + KW;
    constexpr int PITCH = (PC * 3 * 2 + 16 + 3) / 4 * 4;
    const size_t lds = (size_t)3 * PR * PITCH + 4 * 32 * 36 * 4 + (size_t)3 * 32 * TN * (NG * 16 + 16);
    if (lds > 64 * 1024) {
        const bool ok = GV_BIG_LDS_OK((&conv_stem_patch_x3<TN, KW>), 160 * 1024);
        if (!ok) return GV_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL((conv_stem_patch_x3<TN, KW>), dim3((unsigned)(a.nb * ((a.ow + 31) / 32))), dim3(256), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

int launch_stem_x3(const ConvArgs& a, hipStream_t st) {
    if (a.kw == 3) return a.cout <= 32 ? launch_stem_x3_one<1, 3>(a, st) : launch_stem_x3_one<2, 3>(a, st);
    return a.cout <= 32 ? launch_stem_x3_one<1, 7>(a, st) : launch_stem_x3_one<2, 7>(a, st);
}

// [kh][kw][cin][cout] fp32 -> [cout][k-tile][plane][16 bf16]
template <int NP>
__global__ void pack_filter_bf16s(const float* __restrict__ w, int K, int ktiles, int cout,
                                  unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)cout * ktiles * KT;
    if (i >= total) return;
    const int n = (int)(i / (ktiles * KT));
    const int k = (int)(i - (int64_t)n * ktiles * KT);
    float x = (k < K) ? w[(size_t)k * cout + n] : 0.f;
    unsigned short* dst = out + ((size_t)n * ktiles + k / KT) * (NP * 16) + (k % KT);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const __bf16 h = (__bf16)x;
        dst[p * 16] = __builtin_bit_cast(unsigned short, h);
        x -= (float)h;
    }
}

struct TileCfg { int bm, bn; };
constexpr TileCfg kTiles[] = {{128, 128}, {128, 64}, {64, 64}, {128, 96}, {64, 128}, {128, 32},
                              {256, 128}, {128, 256}, {256, 64},    // these three: 8 waves
                              {128, 192}, {64, 192}};               // one n-tile for the many 192-channel layers
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int WM, int WN, int TM, int TN, int NP>
int launch_cfg(const ConvArgs& a0, bool generic, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    ConvArgs a = a0;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    const int tiles_m = gv_ceil_div(a.M, BM);
    const int64_t nwg = (int64_t)tiles_m * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    size_t lds = (size_t)(2 * BM + 2 * BN) * (NP * 32 + 16) + (generic ? (size_t)a.Kpad * 8 : 0);
    if (a.y_p3 | a.y2_p3) {
        const size_t epi = (size_t)(WM * WN) * X3EpiGeom<TN>::BYTES;
        lds = lds > epi ? lds : epi;
    }
    if (generic) {
        if (lds > 64 * 1024) {
            const bool ok = GV_BIG_LDS_OK((&conv_igemm_bf16s<WM, WN, TM, TN, NP, true>), 160 * 1024);
            if (!ok) return GV_E_UNSUPPORTED;
        }
        hipLaunchKernelGGL((conv_igemm_bf16s<WM, WN, TM, TN, NP, true>), dim3((unsigned)nwg), dim3(WM * WN * 64), lds, st, a);
    } else {
        if (lds > 64 * 1024) {
            const bool ok = GV_BIG_LDS_OK((&conv_igemm_bf16s<WM, WN, TM, TN, NP, false>), 160 * 1024);
            if (!ok) return GV_E_UNSUPPORTED;
        }
        hipLaunchKernelGGL((conv_igemm_bf16s<WM, WN, TM, TN, NP, false>), dim3((unsigned)nwg), dim3(WM * WN * 64), lds, st, a);
    }
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <int NP>
int launch_np(int cfg, const ConvArgs& a, bool generic, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_cfg<2, 2, 2, 2, NP>(a, generic, st);
        case 1: return launch_cfg<2, 2, 2, 1, NP>(a, generic, st);
        case 2: return launch_cfg<2, 2, 1, 1, NP>(a, generic, st);
        case 3: return launch_cfg<4, 1, 1, 3, NP>(a, generic, st);
        case 4: return launch_cfg<2, 2, 1, 2, NP>(a, generic, st);
        case 5: return launch_cfg<4, 1, 1, 1, NP>(a, generic, st);
        case 6: return launch_cfg<4, 2, 2, 2, NP>(a, generic, st);
        case 7: return launch_cfg<2, 4, 2, 2, NP>(a, generic, st);
        case 8: return launch_cfg<4, 2, 2, 1, NP>(a, generic, st);
        case 9: return launch_cfg<2, 2, 2, 3, NP>(a, generic, st);
        case 10: return launch_cfg<2, 2, 1, 3, NP>(a, generic, st);
    }
    return GV_E_UNSUPPORTED;
}

}  // namespace

namespace gvconv {

// + the halo-tiled stem kernel (3 planes only) + the wave-specialised kernel's GEMM mode (conv_ws_x3.hip; 3 planes only)
int bf16s_num_cfgs() { return kNumTiles + 1 + wsg_x3_num_cfgs(); }
int bf16s_special_cfg() { return kNumTiles; }

// the 3-channel stems: square 3x3 or 7x7 window, stride 2, <= 64 output channels, plain epilogue
bool bf16s_stem_ok(int planes, const ConvArgs& a) {
    return planes == 3 && !a.y_p3 && a.cin == 3 && a.kh == a.kw && (a.kw == 3 || a.kw == 7) && a.stride == 2 && a.cout <= 64 &&
           a.dil_shift == 0 && a.split == 0 && a.y2 == nullptr && a.res == nullptr &&
           (int64_t)a.nb * a.ih * a.iw * a.x_ld < 0x7fffffffll;
}

// the halo kernel's layer class: 3x3 / stride 1, 32 input channels, <= 64 output channels, plain epilogue
bool bf16s_halo_ok(int planes, const ConvArgs& a, bool generic) {
    return planes == 3 && !generic && !a.y_p3 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.cin == 32 && a.cout <= 64 &&
           a.dil_shift == 0 && a.split == 0 && a.y2 == nullptr && a.oh == a.ih + 2 * a.pad_t - 2 &&
           a.ow == a.iw + 2 * a.pad_l - 2;
}

// GV_CONV_MAXPOOL3S2 on fp32 storage: the halo kernel's class in its 30-pixel strip form, VALID pool, no residual
bool bf16s_halo_pool_ok(int planes, const ConvArgs& a, bool generic) {
    return a.pool == 1 && bf16s_halo_ok(planes, a, generic) && a.res == nullptr && a.cout % 4 == 0 && a.oh >= 3 && a.ow >= 3 &&
           !(gv_ceil_div(a.ow, 16) * 16 * 5 <= gv_ceil_div(a.ow, 30) * 32 * 4);
}

int bf16s_pick_tile(int /*planes*/, int M, int N, int /*K*/) {
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

int bf16s_launch(int planes, int cfg, const ConvArgs& a0, bool generic, hipStream_t st) {
    ConvArgs a = a0;
    a.Kpad = (a.K + KT - 1) / KT * KT;
    a.ktiles = a.Kpad / KT;
    if (cfg == kNumTiles) {
        if (a.pool) return bf16s_halo_pool_ok(planes, a, generic) ? launch_halo_x3(a, st) : GV_E_UNSUPPORTED;
        if (bf16s_stem_ok(planes, a)) return launch_stem_x3(a, st);
        return bf16s_halo_ok(planes, a, generic) ? launch_halo_x3(a, st) : GV_E_UNSUPPORTED;
    }
    if (cfg > kNumTiles) return planes == 3 && !generic ? wsg_x3_launch(cfg - kNumTiles - 1, a0, st) : GV_E_UNSUPPORTED;
    switch (planes) {
        case 3: return launch_np<3>(cfg, a, generic, st);
        case 2: return launch_np<2>(cfg, a, generic, st);
        case 1: return launch_np<1>(cfg, a, generic, st);
    }
    return GV_E_UNSUPPORTED;
}

int64_t bf16s_packed_bytes(int kh, int kw, int cin, int cout, int planes) {
    const int64_t K = (int64_t)kh * kw * cin;
    return (int64_t)cout * ((K + KT - 1) / KT) * planes * 32;
}

int bf16s_pack_filter(const float* w_hwio, int kh, int kw, int cin, int cout, int planes, void* out,
                      hipStream_t st) {
    const int K = kh * kw * cin;
    const int ktiles = (K + KT - 1) / KT;
    const int64_t total = (int64_t)cout * ktiles * KT;
    const dim3 grid((unsigned)gv_ceil_div(total, 256));
    switch (planes) {
        case 3: hipLaunchKernelGGL(pack_filter_bf16s<3>, grid, dim3(256), 0, st, w_hwio, K, ktiles, cout, (unsigned short*)out); break;
        case 2: hipLaunchKernelGGL(pack_filter_bf16s<2>, grid, dim3(256), 0, st, w_hwio, K, ktiles, cout, (unsigned short*)out); break;
        case 1: hipLaunchKernelGGL(pack_filter_bf16s<1>, grid, dim3(256), 0, st, w_hwio, K, ktiles, cout, (unsigned short*)out); break;
        default: return GV_E_UNSUPPORTED;
    }
    GV_LAUNCH_CHECK();
    return GV_OK;
}

}  // namespace gvconv
