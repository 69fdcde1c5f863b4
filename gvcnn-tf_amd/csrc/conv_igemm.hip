// conv_igemm.hip — NHWC convolution as an implicit GEMM on the CDNA4 matrix cores (fp32 path).
//
//   M = nb*oh*ow (output pixels of the whole folded view batch), N = cout, K = kh*kw*cin
//   A[m][k] = x[n, oy*stride + r - pad_t, ox*stride + s - pad_l, c]   (gathered on the fly, zero outside)
//   B[n][k] = packed filter [cout][Kpad]                               (k = (r*kw+s)*cin + c)
//
// Replaces slim.conv2d (+BN+ReLU / +bias / +residual) of the reference: nets/inception_v3.py:97-405,
// nets/resnet_v2.py:79-91, nets/resnet_utils.py:94-105.
//
// Kernel structure (one workgroup = 4 waves = 256 threads, wave64):
//   * block tile BM x BN, k-tile = NCH chunks of 16 (BK = 16 or 32); wave tile (TM x TN) MFMA tiles
//     of 32x32 on v_mfma_f32_32x32x2_f32 (exact fp32: bit-for-bit a k-ordered fmaf chain, so every
//     tile configuration returns bitwise the same result);
//   * a 16-channel chunk never straddles a filter tap (every cin on the path except the 3-channel
//     stems is a multiple of 16), so the gather is one 16-byte load per lane per chunk quarter with a
//     wave-uniform (r, s, c) position per chunk;
//   * A and B k-tiles are staged global -> registers -> LDS (issue-early / write-late: the next
//     tile's global loads are in flight while the current tile is multiplied), LDS double buffered,
//     one barrier per k-tile;
//   * LDS image is [row][BK + 4 pad] fp32 (80- or 144-byte rows: an odd number of 16-byte slots):
//     16-byte ds_write_b128 from the loader, conflict-free ds_read_b128 by the MFMA lanes.
//     Lane (i = lane&31, h = lane>>5) reads 16-byte slots {h, h+2, ...} of its row, so one read
//     feeds four MFMAs; A and B use the SAME k permutation, which leaves each MFMA's k pair intact;
//   * epilogue straight from the accumulators: y = act(acc*scale[c] + shift[c] (+ residual)),
//     written at a channel offset of a wider (concat) buffer through y_ld — this is what removes
//     the tf.concat copies of nets/inception_v3.py:155,...; the second destination y2 is either
//     a second activation of the same value (next ResNet unit's pre-activation,
//     nets/resnet_v2.py:75) or, with GV_CONV_SPLIT, the home of output columns >= split_col, so
//     sibling 1x1 convs that read the same input run as ONE GEMM (input read once);
//   * 1-D grid, n-tiles fastest, remapped so that each XCD (private L2) gets a contiguous chunk
//     of tile ids: the n-tiles that re-read one A panel run on one L2.
#include "conv_common.h"

namespace {

using gvconv::ConvArgs;

constexpr int CH = 16;        // channels per chunk (fp32)
constexpr int KPAD_ALIGN = 32;

template <int WM, int WN, int TM, int TN, int NCH, bool GENERIC>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int BKT = CH * NCH;              // k-tile depth
    constexpr int LDS_LD = BKT + 4;            // padded row (floats): (BKT+4)/4 is odd
    constexpr int QPR = BKT / 4;               // 16-byte quarters per row
    constexpr int A_LOADS = (BM * QPR + 255) / 256;
    constexpr int B_LOADS = (BN * QPR + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                        // [2][BM][LDS_LD]
    float* sB = smem + 2 * BM * LDS_LD;      // [2][BN][LDS_LD]

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

    // ---- loader state -----------------------------------------------------------------------
    // Load slot idx -> (row, 16-byte quarter).  With 16-deep tiles (80-byte LDS rows) a ds_write_b128
    // is serviced 8 lanes at a time: pairing row R (lanes 0-3) with row R+4 (lanes 4-7) makes the two
    // rows' 4-bank slots disjoint (20*4 = 80 = 16 mod 32).  With 32-deep tiles 8 lanes are one row.
    auto slot_row = [](int idx) -> int {
        if constexpr (NCH == 1) {
            const int g = idx >> 3;
            return (g >> 2) * 8 + (g & 3) + 4 * ((idx >> 2) & 1);
        } else {
            return idx / QPR;
        }
    };
    auto slot_q = [](int idx) -> int { return idx % QPR; };
    int a_img[A_LOADS], a_iy0[A_LOADS], a_ix0[A_LOADS];
    const int ohow = a.oh * a.ow;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int idx = tid + i * 256;
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
            a_iy0[i] = -(1 << 28);           // fails every bounds check below -> zeros
            a_ix0[i] = 0;
        }
    }
    const float* b_ptr[B_LOADS];
    bool b_ok[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int idx = tid + i * 256;
        const int row = slot_row(idx);
        const int n = n0 + row;
        b_ok[i] = (row < BN) && (n < a.cout);
        b_ptr[i] = (const float*)a.w + (size_t)(b_ok[i] ? n : 0) * a.Kpad + 4 * slot_q(idx);
    }

    f32x4 ra[A_LOADS], rb[B_LOADS];
    // wave-uniform position of each 16-channel chunk of the NEXT tile to load: filter tap (r, s),
    // channel base c, and whether the chunk lies below K (the zero padding of Kpad is skipped)
    int fr[NCH], fs[NCH], fc[NCH];
    auto step_chunk = [&](int& r, int& s, int& c) {
        c += CH;
        if (c >= a.cin) {
            c = 0;
            if (++s == a.kw) { s = 0; ++r; }
        }
    };
    if constexpr (!GENERIC) {
        fr[0] = fs[0] = fc[0] = 0;
#pragma unroll
        for (int j = 1; j < NCH; ++j) {
            fr[j] = fr[j - 1]; fs[j] = fs[j - 1]; fc[j] = fc[j - 1];
            step_chunk(fr[j], fs[j], fc[j]);
        }
    }

    auto load_tile = [&](int kt) {
        const int k0 = kt * BKT;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int kq = slot_q(tid + i * 256);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!GENERIC) {
                int r = fr[0], s = fs[0], c = fc[0];
                if constexpr (NCH == 2) {
                    if (kq >= 4) { r = fr[1]; s = fs[1]; c = fc[1]; }
                }
                const int iy = a_iy0[i] + r;
                const int ix = a_ix0[i] + s;
                if ((unsigned)iy < (unsigned)a.ih && (unsigned)ix < (unsigned)a.iw && r < a.kh) {
                    const float* p = a.x + ((size_t)(a_img[i] + iy) * a.iw + ix) * a.x_ld + c + 4 * (kq & 3);
                    v = *reinterpret_cast<const f32x4*>(p);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + 4 * kq + j;
                    if (k < a.K) {
                        const int rs = k / a.cin;
                        const int c = k - rs * a.cin;
                        const int r = rs / a.kw;
                        const int s = rs - r * a.kw;
                        const int iy = a_iy0[i] + r;
                        const int ix = a_ix0[i] + s;
                        if ((unsigned)iy < (unsigned)a.ih && (unsigned)ix < (unsigned)a.iw)
                            v[j] = a.x[((size_t)(a_img[i] + iy) * a.iw + ix) * a.x_ld + c];
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b_ok[i]) v = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0);
            rb[i] = v;
        }
    };
    auto advance_taps = [&]() {
        if constexpr (!GENERIC) {
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int t = 0; t < NCH; ++t) step_chunk(fr[j], fs[j], fc[j]);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            const int idx = tid + i * 256;
            if (A_LOADS * 256 == BM * QPR || idx < BM * QPR)
                *reinterpret_cast<f32x4*>(sA + buf * BM * LDS_LD + slot_row(idx) * LDS_LD + 4 * slot_q(idx)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            const int idx = tid + i * 256;
            if (B_LOADS * 256 == BN * QPR || idx < BN * QPR)
                *reinterpret_cast<f32x4*>(sB + buf * BN * LDS_LD + slot_row(idx) * LDS_LD + 4 * slot_q(idx)) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * LDS_LD + 4 * (lane >> 5);
    const float* a_frag = sA + (wm * TM * 32) * LDS_LD + frag_off;
    const float* b_frag = sB + (wn * TN * 32) * LDS_LD + frag_off;

    // ---- main loop ----------------------------------------------------------------------------
    // Software pipeline (per wave):  fragments are double buffered in registers and read one
    // sub-step (8 k) ahead of the MFMAs that consume them; the k-tile barrier sits between the last
    // two MFMA groups of a tile, so the LDS latency of the next tile's first fragments is covered by
    // this wave's own MFMAs instead of by luck with its neighbours; global loads run two k-tiles
    // ahead of their ds_write (registers are re-filled right after they are written to LDS).
    constexpr int NQ = 2 * NCH;                 // sub-steps per k-tile
    f32x4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int set, int buf, int q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fa[set][i] = *reinterpret_cast<const f32x4*>(a_frag + buf * BM * LDS_LD + i * 32 * LDS_LD + 8 * q);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[set][j] = *reinterpret_cast<const f32x4*>(b_frag + buf * BN * LDS_LD + j * 32 * LDS_LD + 8 * q);
    };
    auto mfma_group = [&](int set) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][e], fb[set][j][e], acc[i][j], 0, 0, 0);
    };

    load_tile(0);
    store_tile(0);
    if (a.ktiles > 1) {
        advance_taps();
        load_tile(1);
    }
    __syncthreads();
    read_frags(0, 0, 0);
    for (int kt = 0; kt + 1 < a.ktiles; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int q = 0; q + 1 < NQ; ++q) {
            read_frags((q + 1) & 1, buf, q + 1);
            mfma_group(q & 1);
        }
        store_tile(buf ^ 1);                                       // registers hold tile kt+1
        if (kt + 2 < a.ktiles) {
            advance_taps();
            load_tile(kt + 2);                                     // refill them: two tiles ahead
        }
        __syncthreads();
        read_frags(0, buf ^ 1, 0);
        mfma_group((NQ - 1) & 1);
    }
    {                                                              // last k-tile (peeled: no barrier)
        const int buf = (a.ktiles - 1) & 1;
#pragma unroll
        for (int q = 0; q + 1 < NQ; ++q) {
            read_frags((q + 1) & 1, buf, q + 1);
            mfma_group(q & 1);
        }
        mfma_group((NQ - 1) & 1);
    }

    // ---- epilogue -----------------------------------------------------------------------------
    gvconv::conv_epilogue<TM, TN>(a, acc, m0, n0, wm, wn, lane);
}

// [kh][kw][cin][cout] fp32 -> [cout][Kpad] fp32, zero padded
__global__ void pack_filter_hwio_f32(const float* __restrict__ w, int K, int Kpad, int cout,
                                     float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)cout * Kpad) return;
    const int n = (int)(i / Kpad);
    const int k = (int)(i - (int64_t)n * Kpad);
    out[i] = (k < K) ? w[(size_t)k * cout + n] : 0.f;
}

int g_tile_override = -1;   // tuning hook: force a tile configuration (see gv_conv2d_set_tile_override)
int g_debug = 0;            // ablation bits, see ConvArgs::dbg

struct TileCfg { int bm, bn, nch; };
constexpr TileCfg kTiles[] = {{128, 128, 1}, {128, 64, 1}, {128, 96, 1}, {128, 32, 1}, {64, 64, 1}, {64, 128, 1},
                              {128, 128, 2}, {128, 64, 2}, {128, 96, 2}, {128, 32, 2}, {64, 64, 2}, {64, 128, 2}};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

template <int WM, int WN, int TM, int TN, int NCH>
int launch_cfg(const ConvArgs& a0, bool generic, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    ConvArgs a = a0;
    a.tiles_n = gv_ceil_div(a.cout, BN);
    a.ktiles = a.Kpad / (CH * NCH);
    const int tiles_m = gv_ceil_div(a.M, BM);
    const int64_t nwg = (int64_t)tiles_m * a.tiles_n;
    if (nwg > 0x7fffffff) return GV_E_UNSUPPORTED;
    const size_t lds = (size_t)(2 * BM + 2 * BN) * (CH * NCH + 4) * sizeof(float);
    if constexpr (NCH == 1) {
        if (generic) {
            hipLaunchKernelGGL((conv_igemm_f32<WM, WN, TM, TN, 1, true>), dim3((unsigned)nwg), dim3(256), lds, st, a);
            GV_LAUNCH_CHECK();
            return GV_OK;
        }
    }
    if (lds > 64 * 1024) {
        if (!GV_BIG_LDS_OK((&conv_igemm_f32<WM, WN, TM, TN, NCH, false>), 160 * 1024)) return GV_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL((conv_igemm_f32<WM, WN, TM, TN, NCH, false>), dim3((unsigned)nwg), dim3(256), lds, st, a);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

int launch_tile(int cfg, const ConvArgs& a, bool generic, hipStream_t st) {
    if (generic && cfg >= 6) cfg -= 6;     // the scalar-gather loader exists for 16-deep k-tiles only
    switch (cfg) {
        case 0: return launch_cfg<2, 2, 2, 2, 1>(a, generic, st);
        case 1: return launch_cfg<2, 2, 2, 1, 1>(a, generic, st);
        case 2: return launch_cfg<4, 1, 1, 3, 1>(a, generic, st);
        case 3: return launch_cfg<4, 1, 1, 1, 1>(a, generic, st);
        case 4: return launch_cfg<2, 2, 1, 1, 1>(a, generic, st);
        case 5: return launch_cfg<2, 2, 1, 2, 1>(a, generic, st);
        case 6: return launch_cfg<2, 2, 2, 2, 2>(a, generic, st);
        case 7: return launch_cfg<2, 2, 2, 1, 2>(a, generic, st);
        case 8: return launch_cfg<4, 1, 1, 3, 2>(a, generic, st);
        case 9: return launch_cfg<4, 1, 1, 1, 2>(a, generic, st);
        case 10: return launch_cfg<2, 2, 1, 1, 2>(a, generic, st);
        case 11: return launch_cfg<2, 2, 1, 2, 2>(a, generic, st);
    }
    return GV_E_UNSUPPORTED;
}

// Default choice when the descriptor does not name a tile (plans normally carry a measured one):
// least padded N first, then the larger tile (fewer re-reads of the A panel).
int pick_tile(int M, int N, int K) {
    int best = 1;
    double best_cost = 1e30;
    const int order[] = {0, 2, 1, 3};                 // 128 x {128, 96, 64, 32}
    const double pen[] = {1.00, 1.00, 1.03, 1.10};
    for (int t = 0; t < 4; ++t) {
        const int bn = kTiles[order[t]].bn;
        const double cost = (double)gv_ceil_div(N, bn) * bn * pen[t];
        if (cost < best_cost) { best_cost = cost; best = order[t]; }
    }
    const int64_t blocks = (int64_t)gv_ceil_div(M, 128) * gv_ceil_div(N, kTiles[best].bn);
    if (blocks < 1024) {                              // small problems: halve BM so the grid covers the chip
        if (kTiles[best].bn == 128) best = 5;
        else if (kTiles[best].bn == 64) best = 4;
    }
    if (K >= 64) best += 6;                           // 32-deep k-tiles: half the barriers
    return best;
}

}  // namespace

extern "C" void gv_conv2d_set_tile_override(int cfg) { g_tile_override = cfg; }
extern "C" void gv_conv2d_set_debug(int bits) { g_debug = bits; }
#ifdef GV_PHASE_TIMES
unsigned long long* g_phase_buf = nullptr;
extern "C" void gv_conv2d_set_phase_buffer(void* p) { g_phase_buf = (unsigned long long*)p; }
#endif
static int planes_of(int math_mode) {
    switch (math_mode) {
        case GV_MATH_F32: return 0;
        case GV_MATH_BF16X3: return 3;
        case GV_MATH_BF16X2: return 2;
        case GV_MATH_BF16X1: return 1;
    }
    return -1;
}

// index of the strip / halo kernels of the stem layers among the tile configurations (16-bit storage: math_mode -1)
extern "C" int gv_conv2d_special_tile_cfg(int32_t math_mode) {
    if (math_mode == -1) return gvconv::lp_special_cfg();
    const int np = planes_of(math_mode);
    return np <= 0 ? GV_E_BADARG : gvconv::bf16s_special_cfg();
}

extern "C" int gv_conv2d_num_tile_cfgs(int32_t math_mode) {
    if (math_mode == -1) return gvconv::lp_num_cfgs();     // the 16-bit storage kernels
    if (math_mode == -3) return gvconv::dma_x3_num_cfgs();  // three-plane input (GV_CONV_X_P3)
    const int np = planes_of(math_mode);
    return np < 0 ? GV_E_BADARG : (np == 0 ? kNumTiles : gvconv::bf16s_num_cfgs());
}

extern "C" int64_t gv_packed_filter_bytes(int32_t kh, int32_t kw, int32_t cin, int32_t cout,
                                          int32_t dtype, int32_t math_mode) {
    if (kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return GV_E_BADARG;
    if (dtype == GV_BF16 || dtype == GV_F16) return gvconv::lp_packed_bytes(kh, kw, cin, cout);
    if (dtype != GV_F32) return GV_E_UNSUPPORTED;
    const int np = planes_of(math_mode);
    if (np < 0) return GV_E_BADARG;
    if (np > 0) return gvconv::bf16s_packed_bytes(kh, kw, cin, cout, np);
    const int64_t K = (int64_t)kh * kw * cin;
    return 4 * (int64_t)cout * ((K + KPAD_ALIGN - 1) / KPAD_ALIGN * KPAD_ALIGN);
}

extern "C" int gv_pack_filter_hwio(const float* w_hwio, int32_t kh, int32_t kw, int32_t cin,
                                   int32_t cout, void* w_packed, int32_t dtype, int32_t math_mode,
                                   void* stream) {
    if (!w_hwio || !w_packed || kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return GV_E_BADARG;
    if (dtype == GV_BF16 || dtype == GV_F16)
        return gvconv::lp_pack_filter(w_hwio, kh, kw, cin, cout, dtype, w_packed, (hipStream_t)stream);
    if (dtype != GV_F32) return GV_E_UNSUPPORTED;
    const int np = planes_of(math_mode);
    if (np < 0) return GV_E_BADARG;
    if (np > 0) return gvconv::bf16s_pack_filter(w_hwio, kh, kw, cin, cout, np, w_packed, (hipStream_t)stream);
    const int K = kh * kw * cin;
    const int Kpad = (K + KPAD_ALIGN - 1) / KPAD_ALIGN * KPAD_ALIGN;
    const int64_t total = (int64_t)cout * Kpad;
    hipLaunchKernelGGL(pack_filter_hwio_f32, dim3((unsigned)gv_ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, w_hwio, K, Kpad, cout, (float*)w_packed);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_pack_filters_batched(const gv_pack_job* jobs_dev, int32_t num_jobs, const int32_t* block_job_dev,
                                       int32_t num_blocks, int32_t dtype, void* stream) {
    if (!jobs_dev || !block_job_dev || num_jobs <= 0 || num_blocks <= 0) return GV_E_BADARG;
    if (dtype != GV_BF16 && dtype != GV_F16) return GV_E_UNSUPPORTED;
    return gvconv::lp_pack_filters_batched(jobs_dev, block_job_dev, num_blocks, dtype, (hipStream_t)stream);
}

static int conv2d_fwd_impl(const gv_conv_desc* d, const void* x, const float* xscale, const float* xshift,
                           const void* w_packed, const float* scale, const float* shift, const void* residual,
                           void* y, void* y2, const float* scale2, const float* shift2, void* stream,
                           const gv_bn_stats* stats = nullptr);

extern "C" int gv_conv2d_fwd(const gv_conv_desc* d, const void* x, const void* w_packed,
                             const float* scale, const float* shift, const void* residual,
                             void* y, void* y2, const float* scale2, const float* shift2,
                             void* stream) {
    return conv2d_fwd_impl(d, x, nullptr, nullptr, w_packed, scale, shift, residual, y, y2, scale2, shift2, stream);
}

extern "C" int gv_conv2d_fwd_xpre(const gv_conv_desc* d, const void* x, const float* xscale, const float* xshift,
                                  const void* w_packed, const float* scale, const float* shift,
                                  const void* residual, void* y, void* y2, const float* scale2,
                                  const float* shift2, void* stream) {
    if (!xscale || !xshift) return GV_E_BADARG;
    return conv2d_fwd_impl(d, x, xscale, xshift, w_packed, scale, shift, residual, y, y2, scale2, shift2, stream);
}

extern "C" int gv_conv2d_fwd_bnstats(const gv_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                                     const float* shift, const void* residual, void* y, const gv_bn_stats* stats,
                                     void* stream) {
    if (!stats) return GV_E_BADARG;
    return conv2d_fwd_impl(d, x, nullptr, nullptr, w_packed, scale, shift, residual, y, nullptr, nullptr, nullptr, stream, stats);
}

static int conv2d_fwd_impl(const gv_conv_desc* d, const void* x, const float* xscale, const float* xshift,
                           const void* w_packed, const float* scale, const float* shift, const void* residual,
                           void* y, void* y2, const float* scale2, const float* shift2, void* stream,
                           const gv_bn_stats* stats) {
    if (!d || !x || !w_packed || !scale || !shift || !y) return GV_E_BADARG;
    if (d->nb <= 0 || d->ih <= 0 || d->iw <= 0 || d->cin <= 0 || d->cout <= 0 || d->kh <= 0 ||
        d->kw <= 0 || d->stride <= 0 || d->oh <= 0 || d->ow <= 0 || d->pad_t < 0 || d->pad_l < 0)
        return GV_E_BADARG;
    const bool split = (d->flags & GV_CONV_SPLIT) != 0;
    if (split) {
        if (!y2 || d->split_col <= 0 || d->split_col >= d->cout) return GV_E_BADARG;
        if (d->y_ld < d->split_col || d->y2_ld < d->cout - d->split_col) return GV_E_BADARG;
    } else {
        if (d->y_ld < d->cout) return GV_E_BADARG;
        if (y2 && (!scale2 || !shift2 || d->y2_ld < d->cout)) return GV_E_BADARG;
    }
    if (d->x_ld < d->cin) return GV_E_BADARG;
    if (residual && d->res_ld < d->cout) return GV_E_BADARG;
    // the window of the last output must start inside the padded input
    {
        // (a data-gradient launch may legitimately have output rows no window reaches: they get zeros)
        if (d->in_dilation != 2 && d->y_step != 2 && ((d->oh - 1) * d->stride - d->pad_t >= d->ih ||
                                    (d->ow - 1) * d->stride - d->pad_l >= d->iw))
            return GV_E_BADARG;
    }
    const bool lp = d->dtype == GV_BF16 || d->dtype == GV_F16;
    if (d->dtype != GV_F32 && !lp) return GV_E_UNSUPPORTED;
    if ((d->flags & GV_CONV_X_F32) && !lp) return GV_E_BADARG;
    const int np = lp ? 1 : planes_of(d->math_mode);
    if (np < 0) return GV_E_BADARG;
    const bool xp3 = (d->flags & GV_CONV_X_P3) != 0;
    const bool yp3 = (d->flags & GV_CONV_Y_P3) != 0, y2p3 = (d->flags & GV_CONV_Y2_P3) != 0;
    if ((xp3 || yp3 || y2p3) && (lp || np != 3)) return GV_E_BADARG;
    if (y2p3 && !split) return GV_E_BADARG;
    if (yp3 || y2p3) {
        // whole 16-channel groups per destination, 8-column chunks in the staged epilogue, no second activation
        if (d->cout % 8 != 0 || (y2 && !split) || (residual && d->res_ld % 4 != 0)) return GV_E_UNSUPPORTED;
        if (yp3 && (d->y_ld % 16 != 0 || (split ? d->split_col : d->cout) % 16 != 0)) return GV_E_UNSUPPORTED;
        if (y2p3 && (d->y2_ld % 16 != 0 || (d->cout - d->split_col) % 16 != 0 || d->split_col % 8 != 0)) return GV_E_UNSUPPORTED;
        if (!yp3 && (d->y_ld % 4 != 0 || !gv_aligned16(y))) return GV_E_ALIGN;
        if (split && !y2p3 && (d->y2_ld % 4 != 0 || !gv_aligned16(y2))) return GV_E_ALIGN;
        if (split && d->split_col % 8 != 0) return GV_E_UNSUPPORTED;
    }
    const int ncfg = lp ? gvconv::lp_num_cfgs()
                        : (xp3 ? gvconv::dma_x3_num_cfgs() : (np == 0 ? kNumTiles : gvconv::bf16s_num_cfgs()));
    if (d->tile_cfg < 0 || d->tile_cfg > ncfg) return GV_E_BADARG;
    const int64_t M64 = (int64_t)d->nb * d->oh * d->ow;
    if (M64 > 0x7fffffff || (int64_t)d->nb * d->ih * d->iw > 0x7fffffff) return GV_E_UNSUPPORTED;
    if (!gv_aligned16(w_packed)) return GV_E_ALIGN;

    ConvArgs a;
    a.x = (const float*)x; a.w = w_packed; a.scale = scale; a.shift = shift;
    a.res = (const float*)residual; a.y = (float*)y; a.y2 = (float*)y2;
    a.scale2 = scale2; a.shift2 = shift2;
    a.xscale = xscale; a.xshift = xshift;
    a.nb = d->nb; a.ih = d->ih; a.iw = d->iw; a.cin = d->cin; a.x_ld = d->x_ld;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
    a.oh = d->oh; a.ow = d->ow; a.cout = d->cout; a.y_ld = d->y_ld; a.res_ld = d->res_ld;
    a.y2_ld = d->y2_ld;
    a.M = (int)M64; a.K = d->kh * d->kw * d->cin;
    a.Kpad = (a.K + KPAD_ALIGN - 1) / KPAD_ALIGN * KPAD_ALIGN;
    a.ktiles = 0;
    a.relu = (d->flags & GV_CONV_RELU) ? 1 : 0;
    a.relu2 = (d->flags & GV_CONV_RELU2) ? 1 : 0;
    a.split = split ? d->split_col : 0;
    if (d->relu_cols < 0) return GV_E_BADARG;
    a.relu_limit = d->relu_cols > 0 ? d->relu_cols : 0x7fffffff;
    a.tiles_n = 0;
    a.dbg = g_debug;
#ifdef GV_PHASE_TIMES
    a.phase_buf = g_phase_buf;
#endif
    a.zeros = nullptr;
    a.y_p3 = (d->flags & GV_CONV_Y_P3) ? 1 : 0;
    a.y2_p3 = (d->flags & GV_CONV_Y2_P3) ? 1 : 0;
    if (d->in_dilation != 0 && d->in_dilation != 1 && d->in_dilation != 2) return GV_E_BADARG;
    a.dil_shift = d->in_dilation == 2 ? 1 : 0;
    if (a.dil_shift && (np == 0 || (d->cin % CH != 0) || d->stride != 1)) return GV_E_UNSUPPORTED;
    if (d->y_step != 0) {
        // one parity class of a stride-2 data gradient: stride-1 launch whose output rows land on every second pixel
        if (d->y_step != 2 || d->y_py < 0 || d->y_py > 1 || d->y_px < 0 || d->y_px > 1) return GV_E_BADARG;
        if (2 * (d->oh - 1) + d->y_py >= d->y_ih || 2 * (d->ow - 1) + d->y_px >= d->y_iw) return GV_E_BADARG;
        if (!lp || split || y2 || d->stride != 1 || a.dil_shift) return GV_E_UNSUPPORTED;
        if ((int64_t)d->nb * d->y_ih * d->y_iw > 0x7fffffff) return GV_E_UNSUPPORTED;
        a.y_step = 2; a.y_py = d->y_py; a.y_px = d->y_px; a.y_ih = d->y_ih; a.y_iw = d->y_iw;
    }
    // exact m / (oh*ow) and rem / ow by multiplication: the loaders' row -> (image, y, x) walk (four 32-bit divisions per
    // lane of every workgroup's prologue otherwise, ~35 instructions each) and the parity-class epilogue
    a.y_div_img = gv_fast_div(d->oh * d->ow);
    a.y_div_row = gv_fast_div(d->ow);
    if (stats) {
        // BatchNorm sums in the epilogue: 16-bit storage, one plain destination (the sums are those of the stored values)
        if (stats->mode != GV_BN_STATS_FWD && stats->mode != GV_BN_STATS_BWD) return GV_E_BADARG;
        if (stats->groups <= 0 || stats->nseg <= 0 || stats->nseg > GV_BN_STATS_MAX_SEG) return GV_E_BADARG;
        if (!lp || split || y2 || (d->flags & (GV_CONV_RELU | GV_CONV_RELU2))) return GV_E_UNSUPPORTED;
        // whole 16-byte chunks of 8 channels everywhere (the lean epilogue of the instantiations that fold the sums)
        if (d->cout % 8 != 0 || d->y_ld % 8 != 0 || !gv_aligned16(y) || (residual && (d->res_ld % 8 != 0 || !gv_aligned16(residual))))
            return GV_E_UNSUPPORTED;
        a.st.mode = stats->mode == GV_BN_STATS_FWD ? gvconv::STAT_FWD : gvconv::STAT_BWD;
        a.st.hw = d->oh * d->ow;
        a.st.G = stats->groups;
        a.st.nseg = stats->nseg;
        a.st.hw_magic = a.st.hw > 1 ? (unsigned)((0x100000000ull + (unsigned)a.st.hw - 1) / (unsigned)a.st.hw) : 0u;
        a.st.lds_off = a.st.slots = a.st.fold = a.st.pad_ = 0;
        a.st.dbg = g_debug;
        for (int i = 0; i < stats->nseg; ++i) {
            const gv_bn_stats_seg& g = stats->seg[i];
            if (g.c0 < 0 || g.c1 <= g.c0 || g.c1 > d->cout) return GV_E_BADARG;
            if (stats->mode == GV_BN_STATS_BWD && g.acc && (!g.z || g.z_ld < g.c1 - g.c0 || (g.scale == nullptr) != (g.shift == nullptr)))
                return GV_E_BADARG;
            if (stats->mode == GV_BN_STATS_BWD && g.acc && (g.z_ld % 8 != 0 || !gv_aligned16(g.z))) return GV_E_ALIGN;
            a.st.seg[i].c0 = g.c0; a.st.seg[i].c1 = g.c1; a.st.seg[i].z_ld = g.z_ld; a.st.seg[i].pad_ = 0;
            a.st.seg[i].z = (const unsigned short*)g.z; a.st.seg[i].scale = g.scale; a.st.seg[i].shift = g.shift;
            a.st.seg[i].acc = g.acc;
        }
    }

    if (d->flags & (GV_CONV_MAXPOOL3S2 | GV_CONV_MAXPOOL3S2_SAME)) {
        // conv -> max_pool2d 3x3 / 2 in one launch: y is the pooled tensor (the halo / stem strip kernels' classes only)
        const bool same = (d->flags & GV_CONV_MAXPOOL3S2_SAME) != 0;
        if (same && (d->flags & GV_CONV_MAXPOOL3S2)) return GV_E_BADARG;
        // (fp32 storage: three-plane math, the VALID pool behind the halo kernel's class — checked below)
        if ((!lp && (np != 3 || same || xp3 || yp3)) || split || y2 || residual || stats || xscale || d->y_step != 0 || a.dil_shift ||
            d->oh < 3 || d->ow < 3)
            return GV_E_UNSUPPORTED;
        if (same && ((d->oh | d->ow) & 1)) return GV_E_UNSUPPORTED;          // TF's SAME pads (0, 1) on an even map only
        a.pool = same ? 2 : 1;
        a.ph = same ? d->oh / 2 : (d->oh - 3) / 2 + 1;
        a.pw = same ? d->ow / 2 : (d->ow - 3) / 2 + 1;
        // GV_CONV_POOL_ACT2: the pooled tensor leaves as act2(pool * scale2 + shift2) (16-bit storage, the stem strip kernel)
        if (d->flags & GV_CONV_POOL_ACT2) {
            if (!scale2 || !shift2) return GV_E_BADARG;
            if (!lp) return GV_E_UNSUPPORTED;
        } else {
            a.scale2 = a.shift2 = nullptr;
        }
    } else if (d->flags & GV_CONV_POOL_ACT2) {
        return GV_E_BADARG;
    }
    if (lp) {
        // vector loader: 8-channel (16-byte) chunks inside one filter tap, 16-byte aligned pixels
        const bool xf32 = (d->flags & GV_CONV_X_F32) != 0;
        const bool generic = xf32 || (d->cin % 8 != 0) || (d->x_ld % 8 != 0) || !gv_aligned16(x);
        if (a.pool && !((a.pool == 1 && gvconv::lp_halo_pool_ok(a, generic)) || gvconv::lp_stem_pool_ok(a, xf32)))
            return GV_E_UNSUPPORTED;
        if (a.dil_shift && generic) return GV_E_UNSUPPORTED;
        // the vector loader keeps 32-bit element offsets
        if (!generic && (int64_t)d->nb * d->ih * d->iw * d->x_ld > 0xffffffffll) return GV_E_UNSUPPORTED;
        if (xscale) {
            // pre-activation on load: the register-staged loader of the 1x1 / unpadded class (a padding tap would have
            // to read relu(xshift), not 0), (scale, shift) table of the input channels in LDS
            const int want = g_tile_override >= 0 && g_tile_override < ncfg ? g_tile_override : d->tile_cfg - 1;
            // the special tile index: the streaming TAIL form of the bottleneck launch (csrc/conv_chain.hip), where it serves
            if (want == gvconv::lp_special_cfg() && !generic)
                return gvconv::chain_tail_launch(d->dtype, a, (hipStream_t)stream);
            if (generic || d->kh != 1 || d->kw != 1 || d->pad_t != 0 || d->pad_l != 0 || a.dil_shift ||
                d->cin > 2048 || (want >= 0 && !gvconv::lp_xpre_cfg_ok(want)))
                return GV_E_UNSUPPORTED;
            const int cfg = want >= 0 ? want : gvconv::lp_xpre_pick(a.M, a.cout);
            return gvconv::lp_launch(d->dtype, cfg, a, false, false, (hipStream_t)stream);
        }
        const int cfg = g_tile_override >= 0 && g_tile_override < ncfg ? g_tile_override
                        : (d->tile_cfg > 0 ? d->tile_cfg - 1
                           : ((gvconv::lp_halo_ok(a, generic) || gvconv::lp_stem_ok(a, xf32)) && (a.M >= 100000 || a.pool)
                                  ? gvconv::lp_special_cfg()
                                                                              : gvconv::lp_pick_tile(a.M, a.cout, a.K)));
        return gvconv::lp_launch(d->dtype, cfg, a, generic, xf32, (hipStream_t)stream);
    }
    if (xscale) return GV_E_UNSUPPORTED;                 // pre-activation on load: 16-bit storage only
    if (xp3) {                                           // three-plane input: the LDS-DMA kernel
        if (!gvconv::dma_x3_ok(a) || !gv_aligned16(x) || (y2 && !split)) return GV_E_UNSUPPORTED;
        if (split && (d->split_col % 8 != 0 || d->cout % 8 != 0 || (!yp3 && (d->y_ld % 4 != 0 || !gv_aligned16(y))) ||
                      (!y2p3 && (d->y2_ld % 4 != 0 || !gv_aligned16(y2)))))
            return GV_E_UNSUPPORTED;
        const int cfg = g_tile_override >= 0 && g_tile_override < ncfg ? g_tile_override
                        : (d->tile_cfg > 0 ? d->tile_cfg - 1 : 0);
        return gvconv::dma_x3_launch(cfg, a, (hipStream_t)stream);
    }
    // vector loader needs 16-channel chunks inside one filter tap and 16-byte aligned pixels
    const bool generic = (d->cin % CH != 0) || (d->x_ld % 4 != 0) || !gv_aligned16(x);
    if (a.pool) {                                        // conv -> max pool on fp32 storage: one kernel serves it
        if (!gvconv::bf16s_halo_pool_ok(np, a, generic)) return GV_E_UNSUPPORTED;
        return gvconv::bf16s_launch(np, gvconv::bf16s_special_cfg(), a, generic, (hipStream_t)stream);
    }
    if (np > 0) {
        // the vector loader keeps 32-bit element offsets
        if (!generic && (int64_t)d->nb * d->ih * d->iw * d->x_ld > 0xffffffffll) return GV_E_UNSUPPORTED;
        const int cfg = g_tile_override >= 0 && g_tile_override < ncfg ? g_tile_override
                        : (d->tile_cfg > 0 ? d->tile_cfg - 1
                           : ((gvconv::bf16s_halo_ok(np, a, generic) || gvconv::bf16s_stem_ok(np, a)) &&
                                      a.M >= 100000
                                  ? gvconv::bf16s_special_cfg()
                                                                                     : gvconv::bf16s_pick_tile(np, a.M, a.cout, a.K)));
        return gvconv::bf16s_launch(np, cfg, a, generic, (hipStream_t)stream);
    }
    const int cfg = g_tile_override >= 0 && g_tile_override < kNumTiles ? g_tile_override
                    : (d->tile_cfg > 0 ? d->tile_cfg - 1 : pick_tile(a.M, a.cout, a.K));
    return launch_tile(cfg, a, generic, (hipStream_t)stream);
}

extern "C" int gv_conv2d_time(const gv_conv_desc* d, const void* x, const void* w_packed,
                              const float* scale, const float* shift, void* y, int32_t iters,
                              float* ms_avg_host, void* stream) {
    if (!ms_avg_host || iters <= 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    GV_HIP_CHECK(hipEventCreate(&e0));
    GV_HIP_CHECK(hipEventCreate(&e1));
    int rc = gv_conv2d_fwd(d, x, w_packed, scale, shift, nullptr, y, nullptr, nullptr, nullptr, stream);  // warm
    if (rc == GV_OK) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters && rc == GV_OK; ++i)
            rc = gv_conv2d_fwd(d, x, w_packed, scale, shift, nullptr, y, nullptr, nullptr, nullptr, stream);
        (void)hipEventRecord(e1, st);
        hipError_t e = hipEventSynchronize(e1);
        if (rc == GV_OK && e != hipSuccess) rc = (int)e;
        float ms = 0.f;
        if (rc == GV_OK) { (void)hipEventElapsedTime(&ms, e0, e1); *ms_avg_host = ms / (float)iters; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
