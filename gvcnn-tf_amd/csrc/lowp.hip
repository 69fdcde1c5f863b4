// lowp.hip — the HBM-bound ops of the path on 16-bit storage (GV_BF16 / GV_F16, configs c3-c5):
// pools, stand-alone scale/shift/ReLU, global average pool, the scorer's GAP.Dense(1) and the fused
// view pooling + group fusion.  Same algorithms as pool.hip / grouping.hip; a thread owns 8 consecutive
// channels (one 16-byte load), arithmetic is fp32 and every output is rounded to the storage type
// exactly once (max pooling is rounding-free).
#include <math.h>

#include <type_traits>

#include "lowp.h"
#include "lp_elem.h"

namespace {

using namespace gvlp_elem;

template <typename T, int VEC>
__global__ __launch_bounds__(256) void pool2d_lp(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                 int nb, int ih, int iw, int c, int x_ld, int kh, int kw,
                                                 int stride, int pad_t, int pad_l, int oh, int ow, int y_ld,
                                                 int mode) {
    const int cg = c / VEC;
    const int64_t total = (int64_t)nb * oh * ow * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        const int64_t pix = idx / cg;
        const int ox = (int)(pix % ow);
        const int64_t t = pix / ow;
        const int oy = (int)(t % oh);
        const int n = (int)(t / oh);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = mode == GV_POOL_MAX ? -INFINITY : 0.f;
        int cnt = 0;
        for (int r = 0; r < kh; ++r) {
            const int iy = oy * stride + r - pad_t;
            if ((unsigned)iy >= (unsigned)ih) continue;
            for (int s = 0; s < kw; ++s) {
                const int ix = ox * stride + s - pad_l;
                if ((unsigned)ix >= (unsigned)iw) continue;
                float v[8];
                load_v<T, VEC>(x + ((size_t)(n * ih + iy) * iw + ix) * x_ld + g * VEC, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = mode == GV_POOL_MAX ? fmaxf(acc[e], v[e]) : acc[e] + v[e];
                ++cnt;
            }
        }
        if (mode != GV_POOL_MAX) {
            const float inv = (float)cnt;             // divisor = number of valid taps (TF SAME semantics)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                acc[e] = acc[e] / inv;
                if (mode == GV_POOL_AVG_RELU) acc[e] = fmaxf(acc[e], 0.f);
            }
        }
        store_v<T, VEC>(y + (size_t)pix * y_ld + g * VEC, acc);
    }
}

// 3x3 / stride 1 / SAME average pool: 4 horizontally adjacent outputs of an 8-channel group per thread
// (shared column sums), see avgpool3x3s1_row4_f32 in pool.hip.
// The same pool with TWO output rows per thread (rows oy and oy + 1 of the pair are loaded once) and workgroups mapped so
// that an XCD owns a contiguous range of them: the rows two outputs share are then in ITS L2 (under round-robin dispatch
// vertical neighbours sat on different XCDs and c5's nine pooled branches fetched 3.9 x their input from beyond L2).  Same
// additions in the same order: bitwise the same values.  Launched with exactly ceil(total / 256) workgroups.
template <typename T>
__global__ __launch_bounds__(256) void avgpool3x3s1_row4x2_lp(const unsigned short* __restrict__ x,
                                                              unsigned short* __restrict__ y, int nb, int ih, int iw,
                                                              int c, int x_ld, int y_ld, int relu) {
    const int cg = c >> 3, wg = (iw + 3) >> 2, hg = (ih + 1) >> 1;
    const int64_t total = (int64_t)nb * hg * wg * cg;
    const int64_t idx = (int64_t)gv_xcd_remap((int)blockIdx.x, (int)gridDim.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % cg);
    int64_t t = idx / cg;
    const int xg = (int)(t % wg);
    t /= wg;
    const int yg = (int)(t % hg);
    const int n = (int)(t / hg);
    const int oy0 = yg * 2, ox0 = xg * 4;
    float col[2][6][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) col[q][j][e] = 0.f;
#pragma unroll
    for (int r = -1; r <= 2; ++r) {
        const int iy = oy0 + r;
        if ((unsigned)iy >= (unsigned)ih) continue;
        const unsigned short* rowp = x + ((size_t)(n * ih + iy) * iw) * x_ld + g * 8;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ix = ox0 - 1 + j;
            if ((unsigned)ix >= (unsigned)iw) continue;
            float v[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(rowp + (size_t)ix * x_ld), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (r <= 1) col[0][j][e] += v[e];
                if (r >= 0) col[1][j][e] += v[e];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int oy = oy0 + q;
        if (oy >= ih) break;
        const int rows = 1 + (oy > 0 ? 1 : 0) + (oy + 1 < ih ? 1 : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = ox0 + j;
            if (ox >= iw) break;
            const int cols = 1 + (ox > 0 ? 1 : 0) + (ox + 1 < iw ? 1 : 0);
            const float inv = (float)(rows * cols);           // valid taps only (TF SAME semantics)
            // s / inv for eight values: the correctly rounded reciprocal once, then per value a product and one exact-remainder
            // correction (Markstein) — the correctly rounded quotient, i.e. the bits of the division, at 3 instead of ~10
            // instructions per value (the divisions were most of this kernel's vector instructions)
            // — for a FINITE sum; a non-finite one keeps sum * rcp (inf stays inf, NaN stays NaN, as the division gives)
            const float rcp = 1.0f / inv;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sum = col[q][j][e] + col[q][j + 1][e] + col[q][j + 2][e];
                const float q0 = sum * rcp;
                const float qc = __builtin_fmaf(__builtin_fmaf(-inv, q0, sum), rcp, q0);
                v[e] = __builtin_fabsf(q0) < __builtin_inff() ? qc : q0;   // (+-inf / NaN sums: the correction would turn inf into NaN)
                if (relu) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<u32x4*>(y + ((size_t)(n * ih + oy) * iw + ox) * y_ld + g * 8) = pack8<T>(v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool3x3s1_row4_lp(const unsigned short* __restrict__ x,
                                                            unsigned short* __restrict__ y, int nb, int ih, int iw,
                                                            int c, int x_ld, int y_ld, int relu) {
    const int cg = c >> 3;
    const int wg = (iw + 3) >> 2;
    const int64_t total = (int64_t)nb * ih * wg * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        int64_t t = idx / cg;
        const int xg = (int)(t % wg);
        t /= wg;
        const int oy = (int)(t % ih);
        const int n = (int)(t / ih);
        const int ox0 = xg * 4;
        float col[6][8];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) col[j][e] = 0.f;
        int rows = 0;
#pragma unroll
        for (int r = -1; r <= 1; ++r) {
            const int iy = oy + r;
            if ((unsigned)iy >= (unsigned)ih) continue;
            ++rows;
            const unsigned short* rowp = x + ((size_t)(n * ih + iy) * iw) * x_ld + g * 8;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ix = ox0 - 1 + j;
                if ((unsigned)ix < (unsigned)iw) {
                    float v[8];
                    unpack8<T>(*reinterpret_cast<const u32x4*>(rowp + (size_t)ix * x_ld), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) col[j][e] += v[e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = ox0 + j;
            if (ox >= iw) break;
            const int cols = 1 + (ox > 0 ? 1 : 0) + (ox + 1 < iw ? 1 : 0);
            const float inv = (float)(rows * cols);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = (col[j][e] + col[j + 1][e] + col[j + 2][e]) / inv;
                if (relu) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<u32x4*>(y + ((size_t)(n * ih + oy) * iw + ox) * y_ld + g * 8) = pack8<T>(v);
        }
    }
}

// 3x3 / stride 2 / VALID max pool, RH vertically adjacent outputs of an 8-channel group per thread: the window rows
// 2*oy .. 2*oy + 2 of consecutive outputs share a row, so a thread that walks down RH outputs reads 2*RH + 1 input rows
// instead of 3*RH — and, what matters more, the shared rows are not re-fetched by ANOTHER workgroup on another XCD
// (pool2d_lp: one output per thread, L2 hit 0.32, 1.5x the algorithmic bytes from the fabric; r3_pmc / r4_pmc).
// Giving every workgroup one contiguous span of pool2d_lp's index instead was measured and dropped (round 4: the pools
// of a step 1.10 -> 1.25 ms — 4096 spans in flight at once walk 4096 far-apart regions).
template <typename T, int RH>
__global__ __launch_bounds__(256) void maxpool3x3s2_rows_lp(const unsigned short* __restrict__ x,
                                                            unsigned short* __restrict__ y, int nb, int ih, int iw,
                                                            int c, int x_ld, int oh, int ow, int y_ld) {
    const int cg = c >> 3;
    const int nob = (oh + RH - 1) / RH;
    const int64_t total = (int64_t)nb * nob * ow * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        int64_t t = idx / cg;
        const int ox = (int)(t % ow);
        t /= ow;
        const int ob = (int)(t % nob);
        const int n = (int)(t / nob);
        const int oy0 = ob * RH;
        const int nr = min(RH, oh - oy0);                          // output rows of this thread
        const unsigned short* xp = x + ((size_t)(n * ih + 2 * oy0) * iw + 2 * ox) * x_ld + g * 8;
        unsigned short* yp = y + ((size_t)(n * oh + oy0) * ow + ox) * y_ld + g * 8;
        float acc[8];
#pragma unroll
        for (int r = 0; r <= 2 * RH; ++r) {
            if (r > 2 * nr) break;                                 // (VALID: input row 2*oy + 2 exists for every oy < oh)
            const unsigned short* rp = xp + (size_t)r * iw * x_ld;
            float a[8], b[8], d_[8], hm[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(rp), a);
            unpack8<T>(*reinterpret_cast<const u32x4*>(rp + x_ld), b);
            unpack8<T>(*reinterpret_cast<const u32x4*>(rp + 2 * x_ld), d_);
#pragma unroll
            for (int e = 0; e < 8; ++e) hm[e] = fmaxf(fmaxf(a[e], b[e]), d_[e]);
            if (r == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = hm[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], hm[e]);
                if ((r & 1) == 0) {                                // the third row of output r/2 - 1 = the first of output r/2
                    *reinterpret_cast<u32x4*>(yp + (size_t)(r / 2 - 1) * ow * y_ld) = pack8<T>(acc);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = hm[e];
                }
            }
        }
    }
}

// (A 4 x 4 block of outputs per thread for the 3x3 / stride-1 AVERAGE pool — 36 loads for 16 outputs instead of the row
// form's 18 for 4 — was measured and dropped in round 4: 0.027 -> 0.038 ms on a Mixed_5 branch, 0.018 -> 0.024 ms on a
// Mixed_6 one: a quarter of the threads and 96 live partial sums each.)

template <typename T, int VEC>
__global__ __launch_bounds__(256) void scale_shift_act_lp(const unsigned short* __restrict__ x, int64_t npix, int c,
                                                          int x_ld, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu,
                                                          unsigned short* __restrict__ y, int y_ld) {
    const int cg = c / VEC;
    const int64_t total = npix * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        const int64_t pix = idx / cg;
        float v[8];
        load_v<T, VEC>(x + (size_t)pix * x_ld + g * VEC, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[e] = v[e] * scale[g * VEC + e] + shift[g * VEC + e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        store_v<T, VEC>(y + (size_t)pix * y_ld + g * VEC, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void global_avg_pool_lp(const unsigned short* __restrict__ x, int nb, int hw, int c,
                                                          int x_ld, float* __restrict__ y) {
    const int64_t total = (int64_t)nb * c;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % c);
    const int b = (int)(idx / c);
    const unsigned short* p = x + (size_t)b * hw * x_ld + ch;
    float s = 0.f;
    for (int i = 0; i < hw; ++i) s += up<T>(p[(size_t)i * x_ld]);
    y[idx] = s / (float)hw;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// nets/model.py:144-145 on a 16-bit raw map; fixed reduction tree => bitwise reproducible
template <typename T>
__global__ __launch_bounds__(256) void view_score_partial_lp(const unsigned short* __restrict__ raw, int hw, int cr,
                                                             int raw_ld, const float* __restrict__ kernel,
                                                             const float* __restrict__ bias, int num_views,
                                                             int num_shapes, int order, float* __restrict__ r_img) {
    const int b = blockIdx.x;
    const int v = order == GV_ORDER_SHAPE_MAJOR ? b % num_views : b / num_shapes;
    const float* kv = kernel + (size_t)v * cr;
    const unsigned short* xb = raw + (size_t)b * hw * raw_ld;
    float s = 0.f;
    if ((cr & 7) == 0 && (raw_ld & 7) == 0 && ((((uintptr_t)raw) | ((uintptr_t)kernel)) & 15) == 0) {
        const int cg = cr >> 3;
        const int total = hw * cg;
        // four chunks in flight per thread, each into its own partial sum (one load at a time left this 85 MB read
        // latency-bound: 69 us for the raw tap of 384 views, 1.2 TB/s); the order of the additions is fixed, so every rank
        // of a sharded job still computes the same bits
        float sp[4] = {0.f, 0.f, 0.f, 0.f};
        int i = threadIdx.x;
        for (; i + 3 * 256 < total; i += 4 * 256) {
            u32x4 q[4];
            int gq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = i + u * 256;
                const int p = ii / cg;
                gq[u] = ii - p * cg;
                q[u] = *reinterpret_cast<const u32x4*>(xb + (size_t)p * raw_ld + 8 * gq[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float x[8];
                unpack8<T>(q[u], x);
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(kv + 8 * gq[u]), k1 = *reinterpret_cast<const f32x4*>(kv + 8 * gq[u] + 4);
                sp[u] += x[0] * k0[0] + x[1] * k0[1] + x[2] * k0[2] + x[3] * k0[3] + x[4] * k1[0] + x[5] * k1[1] + x[6] * k1[2] + x[7] * k1[3];
            }
        }
        for (; i < total; i += 256) {
            const int p = i / cg, g = i - p * cg;
            float x[8];
            unpack8<T>(*reinterpret_cast<const u32x4*>(xb + (size_t)p * raw_ld + 8 * g), x);
#pragma unroll
            for (int e = 0; e < 8; ++e) sp[0] += x[e] * kv[8 * g + e];
        }
        s = (sp[0] + sp[1]) + (sp[2] + sp[3]);
    } else {
        const int total = hw * cr;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int p = i / cr, c = i - p * cr;
            s += up<T>(xb[(size_t)p * raw_ld + c]) * kv[c];
        }
    }
    __shared__ float part[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) r_img[b] = (part[0] + part[1] + part[2] + part[3]) / (float)hw + bias[v];
}

// nets/model.py:44-102 fused, one read of every descriptor element (view_pool_fuse_f32 in grouping.hip)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void view_pool_fuse_lp(const unsigned short* __restrict__ F, int V, int N, int64_t E,
                                                         int64_t view_stride, int64_t shape_stride,
                                                         const int* __restrict__ scheme, int G,
                                                         const float* __restrict__ weight, int mode, float fill,
                                                         unsigned short* __restrict__ D, unsigned short* __restrict__ S,
                                                         int64_t scheme_stride, int64_t weight_stride) {
    __shared__ unsigned long long s_mask[64];
    __shared__ float s_w[64];
    __shared__ float s_wsum;
    const int n = blockIdx.y;                       // one shape per grid row: its own scheme when strides != 0
    scheme += (size_t)n * scheme_stride;
    weight += (size_t)n * weight_stride;
    for (int g = threadIdx.x; g < G; g += 256) {
        unsigned long long m = 0;
        for (int v = 0; v < V; ++v)
            if (scheme[g * V + v] != 0) m |= 1ull << v;
        s_mask[g] = m;
        s_w[g] = weight[g];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ws = 0.f;
        for (int g = 0; g < G; ++g) ws = __fadd_rn(ws, s_w[g]);
        s_wsum = ws;
    }
    __syncthreads();
    const float wsum = s_wsum;
    const int64_t eg = E / VEC;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < eg;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = idx * VEC;
        const unsigned short* base = F + (size_t)n * shape_stride + e;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int g = 0; g < G; ++g) {
            unsigned long long m = s_mask[g];
            float d[8];
            if (m == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = fill;
            } else {
                const int cnt = __popcll(m);
                int v = __ffsll((long long)m) - 1;
                m &= m - 1;
                load_v<T, VEC>(base + (size_t)v * view_stride, d);
                while (m) {
                    v = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    float x[8];
                    load_v<T, VEC>(base + (size_t)v * view_stride, x);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) d[k] = mode == GV_VIEWPOOL_MAX ? fmaxf(d[k], x[k]) : d[k] + x[k];
                }
                if (mode == GV_VIEWPOOL_MEAN) {
                    const float c = (float)cnt;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) d[k] = d[k] / c;
                }
            }
            if (D) store_v<T, VEC>(D + ((size_t)g * N + n) * E + e, d);
            const float w = s_w[g];
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = __fadd_rn(acc[k], __fmul_rn(w, d[k]));
        }
        if (S) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = wsum != 0.f ? __fdiv_rn(acc[k], wsum) : 0.f;
            store_v<T, VEC>(S + (size_t)n * E + e, acc);
        }
    }
}

int g_pool_rows = 1;      // multi-row form of the 3x3 / stride-2 max pool (gv_pool2d_set_rows: 0 = one output per thread; A/B)

inline unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int64_t cap = 256 * 16;
    return (unsigned)(b < cap ? (b > 0 ? b : 1) : cap);
}

template <typename T>
int pool2d_t(const gv_pool_desc* d, const unsigned short* x, unsigned short* y, hipStream_t st) {
    const bool vec = (d->c % 8 == 0) && (d->x_ld % 8 == 0) && (d->y_ld % 8 == 0) && gv_aligned16(x) && gv_aligned16(y);
    if (vec && d->mode != GV_POOL_MAX && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 &&
        d->pad_l == 1 && d->oh == d->ih && d->ow == d->iw) {
        const int64_t tot4 = (int64_t)d->nb * d->ih * ((d->iw + 3) / 4) * (d->c / 8);
        const int64_t blk2 = ((int64_t)d->nb * ((d->ih + 1) / 2) * ((d->iw + 3) / 4) * (d->c / 8) + 255) / 256;
        if (blk2 < 0x7fffffff)                                     // two rows per thread, XCD-contiguous workgroups
            hipLaunchKernelGGL(avgpool3x3s1_row4x2_lp<T>, dim3((unsigned)blk2), dim3(256), 0, st, x, y, d->nb, d->ih, d->iw, d->c,
                               d->x_ld, d->y_ld, d->mode == GV_POOL_AVG_RELU ? 1 : 0);
        else
            hipLaunchKernelGGL(avgpool3x3s1_row4_lp<T>, dim3(grid_for(tot4)), dim3(256), 0, st, x, y, d->nb, d->ih,
                               d->iw, d->c, d->x_ld, d->y_ld, d->mode == GV_POOL_AVG_RELU ? 1 : 0);
    } else if (vec && g_pool_rows && d->mode == GV_POOL_MAX && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_t == 0 &&
               d->pad_l == 0 && d->oh == (d->ih - 3) / 2 + 1 && d->ow == (d->iw - 3) / 2 + 1) {
        constexpr int RH = 4;
        const int64_t tot = (int64_t)d->nb * ((d->oh + RH - 1) / RH) * d->ow * (d->c / 8);
        hipLaunchKernelGGL((maxpool3x3s2_rows_lp<T, RH>), dim3(grid_for(tot)), dim3(256), 0, st, x, y, d->nb, d->ih, d->iw,
                           d->c, d->x_ld, d->oh, d->ow, d->y_ld);
    } else if (vec) {
        const int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / 8);
        hipLaunchKernelGGL((pool2d_lp<T, 8>), dim3(grid_for(total)), dim3(256), 0, st, x, y, d->nb, d->ih, d->iw,
                           d->c, d->x_ld, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, d->y_ld, d->mode);
    } else {
        const int64_t total = (int64_t)d->nb * d->oh * d->ow * d->c;
        hipLaunchKernelGGL((pool2d_lp<T, 1>), dim3(grid_for(total)), dim3(256), 0, st, x, y, d->nb, d->ih, d->iw,
                           d->c, d->x_ld, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, d->y_ld, d->mode);
    }
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T>
int ssa_t(const unsigned short* x, int64_t npix, int c, int x_ld, const float* scale, const float* shift, int relu,
          unsigned short* y, int y_ld, hipStream_t st) {
    const bool vec = (c % 8 == 0) && (x_ld % 8 == 0) && (y_ld % 8 == 0) && gv_aligned16(x) && gv_aligned16(y);
    if (vec)
        hipLaunchKernelGGL((scale_shift_act_lp<T, 8>), dim3(grid_for(npix * (c / 8))), dim3(256), 0, st, x, npix, c,
                           x_ld, scale, shift, relu, y, y_ld);
    else
        hipLaunchKernelGGL((scale_shift_act_lp<T, 1>), dim3(grid_for(npix * c)), dim3(256), 0, st, x, npix, c, x_ld,
                           scale, shift, relu, y, y_ld);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

template <typename T>
int fuse_t(const unsigned short* F, int V, int N, int64_t E, int64_t vs, int64_t ss, const int* scheme, int G,
           const float* weight, int mode, float fill, unsigned short* D, unsigned short* S, hipStream_t st,
           int64_t scheme_stride, int64_t weight_stride) {
    const bool vec = (E % 8 == 0) && (vs % 8 == 0) && (ss % 8 == 0) && gv_aligned16(F) && (!D || gv_aligned16(D)) &&
                     (!S || gv_aligned16(S));
    const int64_t per_shape = vec ? E / 8 : E;
    int64_t bx = (per_shape + 255) / 256;
    if (bx > 1024) bx = 1024;
    const dim3 grid((unsigned)bx, (unsigned)N);
    if (vec)
        hipLaunchKernelGGL((view_pool_fuse_lp<T, 8>), grid, dim3(256), 0, st, F, V, N, E, vs, ss,
                           scheme, G, weight, mode, fill, D, S, scheme_stride, weight_stride);
    else
        hipLaunchKernelGGL((view_pool_fuse_lp<T, 1>), grid, dim3(256), 0, st, F, V, N, E, vs, ss,
                           scheme, G, weight, mode, fill, D, S, scheme_stride, weight_stride);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

}  // namespace

namespace gvlp {

int pool2d(const gv_pool_desc* d, const void* x, void* y, hipStream_t st) {
    if (d->dtype == GV_BF16) return pool2d_t<__bf16>(d, (const unsigned short*)x, (unsigned short*)y, st);
    if (d->dtype == GV_F16) return pool2d_t<_Float16>(d, (const unsigned short*)x, (unsigned short*)y, st);
    return GV_E_UNSUPPORTED;
}

int scale_shift_act(int dtype, const void* x, int64_t npix, int c, int x_ld, const float* scale, const float* shift,
                    int relu, void* y, int y_ld, hipStream_t st) {
    if (dtype == GV_BF16) return ssa_t<__bf16>((const unsigned short*)x, npix, c, x_ld, scale, shift, relu, (unsigned short*)y, y_ld, st);
    if (dtype == GV_F16) return ssa_t<_Float16>((const unsigned short*)x, npix, c, x_ld, scale, shift, relu, (unsigned short*)y, y_ld, st);
    return GV_E_UNSUPPORTED;
}

int global_avg_pool(int dtype, const void* x, int nb, int hw, int c, int x_ld, float* y, hipStream_t st) {
    const int64_t total = (int64_t)nb * c;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == GV_BF16)
        hipLaunchKernelGGL(global_avg_pool_lp<__bf16>, grid, dim3(256), 0, st, (const unsigned short*)x, nb, hw, c, x_ld, y);
    else if (dtype == GV_F16)
        hipLaunchKernelGGL(global_avg_pool_lp<_Float16>, grid, dim3(256), 0, st, (const unsigned short*)x, nb, hw, c, x_ld, y);
    else
        return GV_E_UNSUPPORTED;
    GV_LAUNCH_CHECK();
    return GV_OK;
}

int view_score_partial(int dtype, const void* raw, int nb, int hw, int cr, int raw_ld, const float* kernel,
                       const float* bias, int num_views, int order, float* r_img, hipStream_t st) {
    if (dtype == GV_BF16)
        hipLaunchKernelGGL(view_score_partial_lp<__bf16>, dim3(nb), dim3(256), 0, st, (const unsigned short*)raw, hw, cr,
                           raw_ld, kernel, bias, num_views, nb / num_views, order, r_img);
    else if (dtype == GV_F16)
        hipLaunchKernelGGL(view_score_partial_lp<_Float16>, dim3(nb), dim3(256), 0, st, (const unsigned short*)raw, hw,
                           cr, raw_ld, kernel, bias, num_views, nb / num_views, order, r_img);
    else
        return GV_E_UNSUPPORTED;
    GV_LAUNCH_CHECK();
    return GV_OK;
}

int view_pool_fuse(int dtype, const void* F, int V, int N, int64_t E, int64_t vs, int64_t ss, const int* scheme,
                   int G, const float* weight, int mode, float fill, void* D, void* S, hipStream_t st,
                   int64_t scheme_stride, int64_t weight_stride) {
    if (dtype == GV_BF16)
        return fuse_t<__bf16>((const unsigned short*)F, V, N, E, vs, ss, scheme, G, weight, mode, fill,
                              (unsigned short*)D, (unsigned short*)S, st, scheme_stride, weight_stride);
    if (dtype == GV_F16)
        return fuse_t<_Float16>((const unsigned short*)F, V, N, E, vs, ss, scheme, G, weight, mode, fill,
                                (unsigned short*)D, (unsigned short*)S, st, scheme_stride, weight_stride);
    return GV_E_UNSUPPORTED;
}

}  // namespace gvlp

namespace gvlp { int pool_rows() { return g_pool_rows; } }
extern "C" void gv_pool2d_set_rows(int on) { g_pool_rows = on; }
