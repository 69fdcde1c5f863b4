// conv_slab.hip — stride-1 convolution on 16-bit storage whose INPUT is staged once per 16-channel slab and re-used
// by every filter tap (no im2col gather through L2 -> LDS).
//
// Why: the LDS-DMA implicit-GEMM kernel (conv_dma.hip) moves every input element kh*kw times from L2 into LDS.  On the
// 16-bit MFMA that traffic, not the matrix pipe, sets the time of every k > 1 layer of the backbones (measured 6.4-6.7
// TB/s L2 -> LDS chip-wide on Conv2d_4a, the 5x5 / 3x3 layers of Mixed_5 and the 1x7 / 7x1 layers of Mixed_6: the
// LDS-DMA ceiling of MI355X_MICROARCH.md).  Here a workgroup owns BM consecutive output pixels (row-major inside the
// batch: m = (n*OH + oh)*OW + ow) and all BN output channels of a column tile, and walks K as (slab, tap):
//   * A ("the slab"): the S padded input rows those BM pixels need (G0 .. G0+S-1 in the virtual row numbering
//     G = n*Hp + padded_row, Hp = OH + kh - 1: consecutive images' rows are consecutive, so a tile may span images),
//     all Wp = OW + kw - 1 columns, 16 channels = 32 bytes per pixel, padding from a zero page.  Fetched ONCE per slab
//     by LDS-DMA (double buffered: slab s+1 lands while slab s computes).  The fragment of tap (r, s) for output pixel m
//     is the 32 bytes at pixel base(m) + r*Wp + s of that image — a ds_read_b128 at a shifted address, no gather.
//   * B: the 16 x BN filter slice of (tap, slab), a 3-stage LDS ring exactly as in conv_dma.hip.
// k-step order is (slab, tap) instead of (tap, channel): the same products, summed in another order.
// LDS images are [row][32 bytes] with the two 16-byte halves of a row swapped where (row >> 3) & 1 (conv_dma.hip's
// conflict-free pattern for ds_read_b128); one global_load_lds_dwordx4 fills 32 rows, the swap is applied to the source.
//
// Pipeline per k-step j: s_waitcnt vmcnt (this wave's part of k-step j+1 has landed; the instructions issued during
// k-step j-1 may stay in flight) -> s_barrier -> issue the slab share and the filter slice of k-step j+3 -> read the
// fragments of k-step j+1 into the other register set -> MFMAs of k-step j.
#include <type_traits>

#include "conv_common.h"
#include "conv_lp_epi.h"

namespace gvconv {
const void* dma_zero_page();
}

namespace {

struct SlabGeo {
    int Wp, Hp;            // padded width / height of one image (stride 1: OW + kw - 1, OH + kh - 1)
    int tpi;               // tiles per image (tiles never span images), or 0: tiles over the flattened batch
    int a_stage;           // bytes of one slab stage (a multiple of waves * 1 KiB)
    unsigned magic_wp, magic_hp;   // ceil(2^32 / d): exact quotients for the small dividends used here
    int nslab, taps, cg;   // cin / 16, kh * kw, cin / 16
};

__device__ __forceinline__ void slab_dma16(const char* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void slab_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void slab_wait_vm_rt(int n) {       // at most n vector-memory operations stay in flight
    switch (n) {
        case 0: slab_wait_vm<0>(); break;
        case 1: slab_wait_vm<1>(); break;
        case 2: slab_wait_vm<2>(); break;
        case 3: slab_wait_vm<3>(); break;
        case 4: slab_wait_vm<4>(); break;
        case 5: slab_wait_vm<5>(); break;
        case 6: slab_wait_vm<6>(); break;
        case 7: slab_wait_vm<7>(); break;
        default: slab_wait_vm<8>(); break;
    }
}

constexpr int SLAB_AUW_MAX = 8;      // slab DMA instructions per wave (table rows in LDS)
constexpr int SLAB_AP_MAX = 4;       // of which at most this many are issued in one k-step

template <typename T, int WM, int TM, int TN>
__global__ __launch_bounds__(WM * 64) void conv_slab(const ConvArgs a, const SlabGeo g) {
    constexpr int NT = WM * 64;
    constexpr int BM = WM * TM * 32, BN = TN * 32, BST = 3;
    constexpr int UB = BN / 32;                                    // filter DMA instructions per k-step (32 rows x 32 B each)
    constexpr int UBW = (UB + WM - 1) / WM;
    constexpr int B_STAGE = BN * 32;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                               // [2][a_stage]
    char* sB = smem + 2 * g.a_stage;                               // [BST][BN][32]
    unsigned* a_tab = reinterpret_cast<unsigned*>(sB + BST * B_STAGE);   // [SLAB_AUW_MAX][NT] source offsets of the slab units

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = gv_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % a.tiles_n;
    const int tile_m = lid / a.tiles_n;
    const int n0 = tile_n * BN;
    const int ohow = a.oh * a.ow;
    int m0, m_end;
    if (g.tpi > 0) {
        const int img = tile_m / g.tpi;
        m0 = img * ohow + (tile_m - img * g.tpi) * BM;
        m_end = (img + 1) * ohow;
    } else {
        m0 = tile_m * BM;
        m_end = a.M;
    }
    const int m_last = min(m0 + BM, m_end) - 1;
    int G0, npix;
    {
        const int nf = m0 / ohow, ohf = (m0 - nf * ohow) / a.ow;
        const int nl = m_last / ohow, ohl = (m_last - nl * ohow) / a.ow;
        G0 = nf * g.Hp + ohf;
        npix = (nl * g.Hp + ohl - G0 + a.kh) * g.Wp;
    }
    const int auw = ((npix * 2 + 63) / 64 + WM - 1) / WM;          // slab DMA instructions per wave (<= SLAB_AUW_MAX: host)
    const int na_steps = g.taps - (BST - 1);                       // k-steps of a slab that carry the next slab's loads
    const int ap = (auw + na_steps - 1) / na_steps;                // ... this many each (<= SLAB_AP_MAX: host)

    // ---- slab loader: LDS unit U (16 bytes) = pixel U>>1, half (U&1); its source is fixed for the life of the tile
    const char* xb = reinterpret_cast<const char*>(a.x);
    const char* zero_page = reinterpret_cast<const char*>(a.zeros);
    const unsigned pix_bytes = (unsigned)a.x_ld * 2u;
#pragma unroll
    for (int k = 0; k < SLAB_AUW_MAX; ++k) {
        const unsigned U = (unsigned)((k * WM + wave) * 64 + lane);
        const unsigned pix = U >> 1;
        const unsigned lh = (U & 1) ^ ((pix >> 3) & 1);
        const unsigned srow = __umulhi(pix, g.magic_wp);
        const int xin = (int)(pix - srow * (unsigned)g.Wp) - a.pad_l;
        const unsigned Gr = (unsigned)G0 + srow;
        const unsigned n = __umulhi(Gr, g.magic_hp);
        const int y = (int)(Gr - n * (unsigned)g.Hp) - a.pad_t;
        const bool ok = (int)pix < npix && (unsigned)y < (unsigned)a.ih && (unsigned)xin < (unsigned)a.iw && (int)n < a.nb;
        a_tab[k * NT + tid] = ok ? ((n * (unsigned)a.ih + (unsigned)y) * (unsigned)a.iw + (unsigned)xin) * pix_bytes + lh * 16u : 0xffffffffu;
    }
    // (each thread reads back only what it wrote: no barrier needed for the table)
    auto dma_a = [&](int k, int slab) {                            // slab unit k of this wave, into the stage of `slab`
        const unsigned off = a_tab[k * NT + tid];
        const char* src = off != 0xffffffffu ? xb + off + slab * 32 : zero_page;
        slab_dma16(src, sA + (slab & 1) * g.a_stage + (k * WM + wave) * 1024);
    };

    // ---- filter loader: row n of the packed [cout][Kpad] filter, 16-value group kc = tap * (cin/16) + slab
    const int lrow = lane >> 1;
    const int b_lh = (lane & 1) ^ ((lrow >> 3) & 1);
    const char* b_ptr[UBW];
    int b_rb[UBW];
#pragma unroll
    for (int i = 0; i < UBW; ++i) {
        int rb = wave + i * WM;
        rb = rb < UB ? rb : UB - 1;                                // surplus slots re-load the last block (same bytes)
        b_rb[i] = rb;
        int n = n0 + rb * 32 + lrow;
        n = n < a.cout ? n : a.cout - 1;
        b_ptr[i] = reinterpret_cast<const char*>(a.w) + (size_t)n * a.Kpad * 2 + b_lh * 16;
    }
    auto dma_b = [&](int i, int kc, int stage) { slab_dma16(b_ptr[i] + kc * 32, sB + stage * B_STAGE + b_rb[i] * 1024); };

    // ---- fragments
    const int fr_row = lane & 31, fr_h = lane >> 5;
    int bp[TM];                                                    // pixel index of this lane's output pixel in the slab
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = min(m0 + (wave * TM + i) * 32 + fr_row, m_last);
        const int n = m / ohow, rem = m - n * ohow;
        const int oy = rem / a.ow, ox = rem - oy * a.ow;
        bp[i] = (n * g.Hp + oy - G0) * g.Wp + ox;
    }
    const int b_frag = fr_row * 32 + ((fr_h ^ ((fr_row >> 3) & 1)) << 4);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 fa[2][TM], fb[2][TN];
    // fragments of the k-step (slab `sl`, tap offset `toff` pixels) whose filter slice sits in ring stage `bst`
    auto read_frags = [&](auto setc, int sl, int toff, int bst) {
        constexpr int S = decltype(setc)::value;
        const char* ab = sA + (sl & 1) * g.a_stage;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pix = bp[i] + toff;
            fa[S][i] = *reinterpret_cast<const u32x4*>(ab + pix * 32 + ((fr_h ^ ((pix >> 3) & 1)) << 4));
        }
        const char* bb = sB + bst * B_STAGE + b_frag;
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[S][j] = *reinterpret_cast<const u32x4*>(bb + j * 1024);
    };

    const int ksteps = g.nslab * g.taps;
    // issue stream state: the k-step whose filter slice is issued next (BST ahead of the MFMAs), and the slab share
    int is_tap = 0, is_slab = 0;                                   // (tap, slab) of k-step j + BST
    auto issue_b = [&](int stage) {
        const int kc = is_tap * g.cg + is_slab;
#pragma unroll
        for (int i = 0; i < UBW; ++i) dma_b(i, kc, stage);
        if (++is_tap == g.taps) { is_tap = 0; ++is_slab; }
    };

    // ---- prologue: slab 0, filter slices 0 .. BST-1
    for (int k = 0; k < auw; ++k) dma_a(k, 0);
#pragma unroll
    for (int t = 0; t < BST; ++t)
        if (t < ksteps) issue_b(t);
    slab_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(std::integral_constant<int, 0>{}, 0, 0, 0);

    int tap = 0, slab = 0, fr = 0, fs = 0;                         // of k-step j; (fr, fs) = filter row / column of `tap`
    int prev_issued = 0;                                           // DMA instructions this wave issued during k-step j-1
    auto body = [&](auto setc, int j) {
        constexpr int S = decltype(setc)::value;
        const bool more = j + 1 < ksteps;
        // (tap, slab) of k-step j+1
        int ntap = tap + 1, nslab_ = slab, nfr = fr, nfs = fs + 1;
        if (nfs == a.kw) { nfs = 0; ++nfr; }
        if (ntap == g.taps) { ntap = 0; ++nslab_; nfr = 0; nfs = 0; }
        int stage_j = j % BST;
        if (more) {
            slab_wait_vm_rt(prev_issued);                          // k-step j+1's slice (and its slab) landed; j-1's issue may fly
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's fragment reads of k-step j are done
            __builtin_amdgcn_s_barrier();
        }
        // issue: the next slab's share of this k-step (older than the filter slice: see the wait above), then the slice
        int issued = 0;
        const bool has_next_slab = slab + 1 < g.nslab;
        const int k_lo = tap * ap;
        const bool do_b = j + BST < ksteps;
        if (more) read_frags(std::integral_constant<int, S ^ 1>{}, nslab_, nfr * g.Wp + nfs, (j + 1) % BST);
#pragma unroll
        for (int mi = 0; mi < TM * TN; ++mi) {
            const int i = mi / TN, jn = mi % TN;
            acc[i][jn] = mfma16<T>(fa[S][i], fb[S][jn], acc[i][jn]);
            if (mi < SLAB_AP_MAX) {
                if (has_next_slab && tap < na_steps && mi < ap && k_lo + mi < auw) { dma_a(k_lo + mi, slab + 1); ++issued; }
            }
            if (mi == (TM * TN > SLAB_AP_MAX ? SLAB_AP_MAX : TM * TN - 1)) {
                if (do_b) { issue_b(stage_j); issued += UBW; }
            }
        }
        if (TM * TN < SLAB_AP_MAX) {                               // (fewer MFMAs than slab slots: the rest behind them)
#pragma unroll
            for (int e = TM * TN; e < SLAB_AP_MAX; ++e)
                if (has_next_slab && tap < na_steps && e < ap && k_lo + e < auw) { dma_a(k_lo + e, slab + 1); ++issued; }
        }
        prev_issued = issued;
        tap = ntap; slab = nslab_; fr = nfr; fs = nfs;
    };
    {
        int j = 0;
        for (; j + 1 < ksteps; j += 2) {
            body(std::integral_constant<int, 0>{}, j);
            body(std::integral_constant<int, 1>{}, j + 1);
        }
        if (j < ksteps) body(std::integral_constant<int, 0>{}, j);
    }

    __syncthreads();                                               // every wave is done with the rings: reuse them for staging
    ConvArgs b = a;
    b.M = m_end;                                                   // rows past the image (per-image tiling) are not stored
    lp_epilogue_staged<T, TM, TN>(b, acc, m0, n0, wave, 0, lane, reinterpret_cast<float*>(smem + wave * EpiGeom<TN>::BYTES));
}

unsigned slab_magic(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

// geometry of the layer for BM-pixel tiles; false when the slab does not fit the kernel's limits
bool slab_geometry(const ConvArgs& a, int BM, int WM, int BN, int TN, SlabGeo* g, size_t* lds) {
    const int ohow = a.oh * a.ow;
    g->Wp = a.ow + a.kw - 1;
    g->Hp = a.oh + a.kh - 1;
    g->taps = a.kh * a.kw;
    g->nslab = g->cg = a.cin / 16;
    g->tpi = ohow >= 4 * BM ? gv_ceil_div(ohow, BM) : 0;
    const int rows_out = (BM - 1 + a.ow - 1) / a.ow + 1;
    int s_max;
    if (g->tpi > 0) {
        s_max = (rows_out < a.oh ? rows_out : a.oh) + a.kh - 1;
    } else {
        const int crossings = (BM - 1) / ohow + 1;
        s_max = rows_out - 1 + crossings * (a.kh - 1) + a.kh;
    }
    const int64_t units = (int64_t)s_max * g->Wp * 2;
    const int instrs = (int)((units + 63) / 64);
    const int auw = (instrs + WM - 1) / WM;
    if (auw > SLAB_AUW_MAX) return false;
    const int na_steps = g->taps - 2;
    if (na_steps < 1 || (auw + na_steps - 1) / na_steps > SLAB_AP_MAX) return false;
    g->a_stage = auw * WM * 1024;
    g->magic_wp = slab_magic(g->Wp);
    g->magic_hp = slab_magic(g->Hp);
    const size_t ring = (size_t)2 * g->a_stage + (size_t)3 * BN * 32 + (size_t)SLAB_AUW_MAX * WM * 64 * 4;
    const size_t epi = (size_t)WM * (TN % 2 == 0 ? 32 * 64 * 4 : 32 * 32 * 4);       // EpiGeom<TN>::BYTES per wave
    *lds = ring > epi ? ring : epi;
    return *lds <= 160 * 1024;
}

template <typename T, int WM, int TM, int TN>
int launch_slab(const ConvArgs& a0, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = TN * 32;
    ConvArgs a = a0;
    SlabGeo g;
    size_t lds = 0;
    if (!slab_geometry(a, BM, WM, BN, TN, &g, &lds)) return GV_E_UNSUPPORTED;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    const int64_t tiles_m = g.tpi > 0 ? (int64_t)a.nb * g.tpi : gv_ceil_div(a.M, BM);
    const int64_t nwg = tiles_m * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    a.zeros = gvconv::dma_zero_page();
    if (!a.zeros) return GV_E_UNSUPPORTED;
    auto kern = &conv_slab<T, WM, TM, TN>;
    if (lds > 64 * 1024) {
        static bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024) == hipSuccess;
        if (!ok) return GV_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WM * 64), lds, st, a, g);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T>
int launch_slab_cfg(int cfg, const ConvArgs& a, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_slab<T, 4, 2, 2>(a, st);      // 256 x 64, 4 waves
        case 1: return launch_slab<T, 4, 2, 3>(a, st);      // 256 x 96
        case 2: return launch_slab<T, 4, 2, 4>(a, st);      // 256 x 128
        case 3: return launch_slab<T, 4, 2, 6>(a, st);      // 256 x 192 (one wave per SIMD)
        case 4: return launch_slab<T, 8, 2, 2>(a, st);      // 512 x 64, 8 waves
        case 5: return launch_slab<T, 8, 2, 3>(a, st);      // 512 x 96
        case 6: return launch_slab<T, 8, 2, 4>(a, st);      // 512 x 128
        case 7: return launch_slab<T, 8, 1, 6>(a, st);      // 256 x 192, 8 waves
        case 8: return launch_slab<T, 4, 1, 6>(a, st);      // 128 x 192, 4 waves
        case 9: return launch_slab<T, 4, 1, 4>(a, st);      // 128 x 128, 4 waves
    }
    return GV_E_UNSUPPORTED;
}

}  // namespace

namespace gvconv {

int slab_lp_num_cfgs() { return 10; }

// the slab kernel's layer class: stride 1, k > 1, whole 16-channel groups, 16-byte aligned pixels, 32-bit byte offsets
bool slab_lp_ok(const ConvArgs& a, bool generic, bool xf32) {
    return !generic && !xf32 && a.stride == 1 && a.dil_shift == 0 && a.kh * a.kw >= 3 && a.cin % 16 == 0 && a.x_ld % 8 == 0 &&
           a.split == 0 && a.y2 == nullptr && (int64_t)a.nb * a.ih * a.iw * a.x_ld * 2 < 0xffffffffll && a.Kpad % 16 == 0 &&
           a.ow + a.kw - 1 < 4096 && a.oh + a.kh - 1 < 4096;
}

int slab_lp_launch(int dtype, int cfg, const ConvArgs& a, hipStream_t st) {
    if (dtype == GV_BF16) return launch_slab_cfg<__bf16>(cfg, a, st);
    if (dtype == GV_F16) return launch_slab_cfg<_Float16>(cfg, a, st);
    return GV_E_UNSUPPORTED;
}

}  // namespace gvconv
