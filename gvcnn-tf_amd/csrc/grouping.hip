// grouping.hip — the grouping module of nets/model.py on device (fp32 path).
//
//   scorer            model.py:144-147   GAP -> Dense(1) per view -> batch mean -> sigmoid(log|.|)
//   group assignment  model.py:16-41     bin = (int)(score * 10f) ; weight = 1 + count
//   view pooling      model.py:44-74     per group: max over its views, ones when empty
//   group fusion      model.py:77-102    sum_g w_g D_g / sum_g w_g
//   classifier        model.py:163-164   GAP -> Dense(C)
//
// The reference computes scheme/weight with host numpy between two partial_run calls
// (train.py:270-288): scores D2H, Python loops, scheme/weight H2D.  Here they stay on device.
// The pooling + fusion kernel is a single pass over the V view descriptors: every descriptor
// element is read exactly once (the algorithmic minimum; the TF graph makes >= 3 passes).
#include <math.h>

#include <type_traits>

#include "gv_common.h"
#include "lowp.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ int view_of_image(int b, int num_views, int num_shapes, int order) {
    return order == GV_ORDER_SHAPE_MAJOR ? b % num_views : b / num_shapes;
}

// r_img[b] = (1/hw) * sum_{p,c} raw[b,p,c] * k[v(b)][c] + bias[v(b)]; one workgroup per image,
// fixed reduction tree => bitwise reproducible.
__global__ __launch_bounds__(256) void view_score_partial_f32(
    const float* __restrict__ raw, int hw, int cr, int raw_ld, const float* __restrict__ kernel,
    const float* __restrict__ bias, int num_views, int num_shapes, int order,
    float* __restrict__ r_img) {
    const int b = blockIdx.x;
    const int v = view_of_image(b, num_views, num_shapes, order);
    const float* kv = kernel + (size_t)v * cr;
    const float* xb = raw + (size_t)b * hw * raw_ld;
    float s = 0.f;
    if ((cr & 3) == 0 && (raw_ld & 3) == 0 && ((((uintptr_t)raw) | ((uintptr_t)kernel)) & 15) == 0) {
        const int cg = cr >> 2;
        const int total = hw * cg;
        // four chunks in flight per thread, each into its own partial sum (lowp.hip: view_score_partial_lp); fixed order
        float sp[4] = {0.f, 0.f, 0.f, 0.f};
        int i = threadIdx.x;
        for (; i + 3 * 256 < total; i += 4 * 256) {
            f32x4 q[4];
            int gq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = i + u * 256;
                const int p = ii / cg;
                gq[u] = ii - p * cg;
                q[u] = *reinterpret_cast<const f32x4*>(xb + (size_t)p * raw_ld + 4 * gq[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 k = *reinterpret_cast<const f32x4*>(kv + 4 * gq[u]);
                sp[u] += q[u][0] * k[0] + q[u][1] * k[1] + q[u][2] * k[2] + q[u][3] * k[3];
            }
        }
        for (; i < total; i += 256) {
            const int p = i / cg, g = i - p * cg;
            const f32x4 x = *reinterpret_cast<const f32x4*>(xb + (size_t)p * raw_ld + 4 * g);
            const f32x4 k = *reinterpret_cast<const f32x4*>(kv + 4 * g);
            sp[0] += x[0] * k[0] + x[1] * k[1] + x[2] * k[2] + x[3] * k[3];
        }
        s = (sp[0] + sp[1]) + (sp[2] + sp[3]);
    } else {
        const int total = hw * cr;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int p = i / cr, c = i - p * cr;
            s += xb[(size_t)p * raw_ld + c] * kv[c];
        }
    }
    __shared__ float part[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) r_img[b] = (part[0] + part[1] + part[2] + part[3]) / (float)hw + bias[v];
}

// score[v] = sigmoid(log(|mean_n r_img[b(n,v)]|)); one thread per view, n ascending.
__global__ void view_score_finalize_f32(const float* __restrict__ r_img, int num_shapes, int num_views,
                                        int order, float* __restrict__ scores) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= num_views) return;
    float s = 0.f;
    for (int n = 0; n < num_shapes; ++n)
        s += r_img[order == GV_ORDER_SHAPE_MAJOR ? n * num_views + v : v * num_shapes + n];
    const float r = fabsf(s / (float)num_shapes);
    const float lg = logf(r);                        // r == 0 -> -inf -> score 0
    scores[v] = 1.0f / (1.0f + expf(-lg));
}

// One workgroup.  Integer path: must be bit-exact with numpy's int(np.float32(score) * 10).
__global__ __launch_bounds__(64) void group_assign_kernel(const float* __restrict__ scores, int V, int G,
                                                          int num_bins, int* __restrict__ gidx,
                                                          int* __restrict__ scheme,
                                                          float* __restrict__ weight,
                                                          int* __restrict__ status) {
    __shared__ int s_gidx[64];
    __shared__ int s_status;
    const int t = threadIdx.x;
    if (t == 0) s_status = 0;
    __syncthreads();
    if (t < V) {
        const float sc = scores[t];
        const float prod = __fmul_rn(sc, (float)num_bins);   // one IEEE fp32 multiply, no contraction
        int b = (int)prod;                                   // truncation toward zero
        if (sc != sc) { atomicOr(&s_status, 2); b = -1; }
        else if (b >= G || b < 0 || prod >= 2147483648.0f) { atomicOr(&s_status, 1); if (prod >= 2147483648.0f) b = 0x7fffffff; }
        s_gidx[t] = b;
        gidx[t] = b;
    }
    __syncthreads();
    for (int i = t; i < G * V; i += 64) {
        const int g = i / V, v = i - g * V;
        scheme[i] = (s_gidx[v] == g) ? 1 : 0;
    }
    for (int g = t; g < G; g += 64) {
        int cnt = 1;                                          // model.py:32 `sum = 1`
        for (int v = 0; v < V; ++v) cnt += (s_gidx[v] == g) ? 1 : 0;
        weight[g] = (float)cnt;
    }
    if (t == 0) *status = s_status;
}

__global__ void group_weight_kernel(const int* __restrict__ scheme, int G, int V, float* __restrict__ weight) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    int cnt = 1;                                              // model.py:32 `sum = 1`
    for (int v = 0; v < V; ++v) cnt += (scheme[g * V + v] == 1) ? 1 : 0;   // model.py:34
    weight[g] = (float)cnt;
}

// Fused view pooling + group fusion.  Thread per (shape n, VEC consecutive descriptor elements).
// Group membership is turned into per-group 64-bit view masks held in LDS; each view of each
// group is loaded once, so a one-hot scheme costs exactly V loads per output element.
template <int VEC>
__global__ __launch_bounds__(256) void view_pool_fuse_f32(
    const float* __restrict__ F, int V, int N, int64_t E, int64_t view_stride, int64_t shape_stride,
    const int* __restrict__ scheme, int G, const float* __restrict__ weight, int mode, float fill,
    float* __restrict__ D, float* __restrict__ S, int64_t scheme_stride, int64_t weight_stride) {
    using Vt = typename std::conditional<VEC == 4, f32x4, float>::type;
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
        for (int g = 0; g < G; ++g) ws = __fadd_rn(ws, s_w[g]);   // tf.reduce_sum(group_weight_list)
        s_wsum = ws;
    }
    __syncthreads();
    const float wsum = s_wsum;
    const int64_t eg = E / VEC;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < eg;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = idx * VEC;
        const float* base = F + (size_t)n * shape_stride + e;
        Vt acc;
        if constexpr (VEC == 4) acc = Vt{0.f, 0.f, 0.f, 0.f}; else acc = 0.f;
        for (int g = 0; g < G; ++g) {
            unsigned long long m = s_mask[g];
            Vt d;
            if (m == 0) {
                if constexpr (VEC == 4) d = Vt{fill, fill, fill, fill}; else d = fill;
            } else {
                const int cnt = __popcll(m);
                int v = __ffsll((long long)m) - 1;
                m &= m - 1;
                d = *reinterpret_cast<const Vt*>(base + (size_t)v * view_stride);
                while (m) {
                    v = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const Vt x = *reinterpret_cast<const Vt*>(base + (size_t)v * view_stride);
                    if (mode == GV_VIEWPOOL_MAX) {
                        if constexpr (VEC == 4) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) d[k] = fmaxf(d[k], x[k]);
                        } else {
                            d = fmaxf(d, x);
                        }
                    } else {
                        d += x;
                    }
                }
                if (mode == GV_VIEWPOOL_MEAN) {
                    const float c = (float)cnt;
                    if constexpr (VEC == 4) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) d[k] = d[k] / c;
                    } else {
                        d = d / c;
                    }
                }
            }
            if (D) *reinterpret_cast<Vt*>(D + ((size_t)g * N + n) * E + e) = d;
            const float w = s_w[g];
            if constexpr (VEC == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __fadd_rn(acc[k], __fmul_rn(w, d[k]));   // multiply, add_n
            } else {
                acc = __fadd_rn(acc, __fmul_rn(w, d));
            }
        }
        if (S) {
            if constexpr (VEC == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = wsum != 0.f ? __fdiv_rn(acc[k], wsum) : 0.f;   // tf.div
            } else {
                acc = wsum != 0.f ? __fdiv_rn(acc, wsum) : 0.f;
            }
            *reinterpret_cast<Vt*>(S + (size_t)n * E + e) = acc;
        }
    }
}

// y[n][c] = x[n][:] . kernel[:, c] + bias[c]; one workgroup per (n, c) pair group.
__global__ __launch_bounds__(256) void dense_f32(const float* __restrict__ x, int f,
                                                 const float* __restrict__ kernel,
                                                 const float* __restrict__ bias, int c,
                                                 float* __restrict__ y) {
    const int n = blockIdx.x;
    const int cc = blockIdx.y;
    const float* xr = x + (size_t)n * f;
    float s = 0.f;
    for (int i = threadIdx.x; i < f; i += 256) s += xr[i] * kernel[(size_t)i * c + cc];
    __shared__ float part[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) y[(size_t)n * c + cc] = part[0] + part[1] + part[2] + part[3] + bias[cc];
}

}  // namespace

extern "C" int gv_view_score_partial(const void* raw, int32_t nb, int32_t hw, int32_t cr,
                                     int32_t raw_ld, const float* kernel, const float* bias,
                                     int32_t num_views, int32_t order, float* r_img, int32_t dtype,
                                     void* stream) {
    if (!raw || !kernel || !bias || !r_img || nb <= 0 || hw <= 0 || cr <= 0 || raw_ld < cr ||
        num_views <= 0 || nb % num_views != 0)
        return GV_E_BADARG;
    if (order != GV_ORDER_SHAPE_MAJOR && order != GV_ORDER_VIEW_MAJOR) return GV_E_BADARG;
    if (dtype != GV_F32)
        return gvlp::view_score_partial(dtype, raw, nb, hw, cr, raw_ld, kernel, bias, num_views, order, r_img,
                                        (hipStream_t)stream);
    hipLaunchKernelGGL(view_score_partial_f32, dim3(nb), dim3(256), 0, (hipStream_t)stream,
                       (const float*)raw, hw, cr, raw_ld, kernel, bias, num_views, nb / num_views,
                       order, r_img);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_view_score_finalize(const float* r_img, int32_t num_shapes, int32_t num_views,
                                      int32_t order, float* scores, void* stream) {
    if (!r_img || !scores || num_shapes <= 0 || num_views <= 0) return GV_E_BADARG;
    if (order != GV_ORDER_SHAPE_MAJOR && order != GV_ORDER_VIEW_MAJOR) return GV_E_BADARG;
    hipLaunchKernelGGL(view_score_finalize_f32, dim3((num_views + 63) / 64), dim3(64), 0,
                       (hipStream_t)stream, r_img, num_shapes, num_views, order, scores);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_group_assign(const float* scores, int32_t num_views, int32_t num_groups,
                               int32_t num_bins, int32_t* gidx, int32_t* scheme, float* weight,
                               int32_t* status, void* stream) {
    if (!scores || !gidx || !scheme || !weight || !status) return GV_E_BADARG;
    if (num_views <= 0 || num_groups <= 0 || num_bins <= 0) return GV_E_BADARG;
    if (num_views > 64 || num_groups > 64) return GV_E_UNSUPPORTED;
    hipLaunchKernelGGL(group_assign_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scores,
                       num_views, num_groups, num_bins, gidx, scheme, weight, status);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_group_weight(const int32_t* scheme, int32_t num_groups, int32_t num_views,
                               float* weight, void* stream) {
    if (!scheme || !weight || num_groups <= 0 || num_views <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(group_weight_kernel, dim3((num_groups + 63) / 64), dim3(64), 0,
                       (hipStream_t)stream, scheme, num_groups, num_views, weight);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

static int pool_fuse_launch(const void* F, int32_t num_views, int32_t num_shapes, int64_t E, int64_t view_stride,
                            int64_t shape_stride, const int32_t* scheme, int32_t num_groups, const float* weight,
                            int32_t mode, float empty_fill, void* D, void* S, int32_t dtype, void* stream,
                            int64_t scheme_stride, int64_t weight_stride) {
    if (!F || !scheme || !weight || (!D && !S)) return GV_E_BADARG;
    if (num_views <= 0 || num_shapes <= 0 || E <= 0 || num_groups <= 0 || view_stride < 0 ||
        shape_stride < 0)
        return GV_E_BADARG;
    if (mode != GV_VIEWPOOL_MAX && mode != GV_VIEWPOOL_MEAN) return GV_E_BADARG;
    if (num_views > 64 || num_groups > 64 || num_shapes > 65535) return GV_E_UNSUPPORTED;
    if (dtype != GV_F32)
        return gvlp::view_pool_fuse(dtype, F, num_views, num_shapes, E, view_stride, shape_stride, scheme, num_groups,
                                    weight, mode, empty_fill, D, S, (hipStream_t)stream, scheme_stride, weight_stride);
    const bool vec = (E % 4 == 0) && (view_stride % 4 == 0) && (shape_stride % 4 == 0) &&
                     gv_aligned16(F) && (!D || gv_aligned16(D)) && (!S || gv_aligned16(S));
    const int64_t per_shape = vec ? E / 4 : E;
    int64_t bx = (per_shape + 255) / 256;
    if (bx > 1024) bx = 1024;
    const dim3 grid((unsigned)bx, (unsigned)num_shapes);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(view_pool_fuse_f32<4>, grid, dim3(256), 0, st,
                           (const float*)F, num_views, num_shapes, E, view_stride, shape_stride,
                           scheme, num_groups, weight, mode, empty_fill, (float*)D, (float*)S, scheme_stride,
                           weight_stride);
    else
        hipLaunchKernelGGL(view_pool_fuse_f32<1>, grid, dim3(256), 0, st,
                           (const float*)F, num_views, num_shapes, E, view_stride, shape_stride,
                           scheme, num_groups, weight, mode, empty_fill, (float*)D, (float*)S, scheme_stride,
                           weight_stride);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_view_pool_fuse_fwd(const void* F, int32_t num_views, int32_t num_shapes, int64_t E,
                                     int64_t view_stride, int64_t shape_stride,
                                     const int32_t* scheme, int32_t num_groups, const float* weight,
                                     int32_t mode, float empty_fill, void* D, void* S, int32_t dtype,
                                     void* stream) {
    return pool_fuse_launch(F, num_views, num_shapes, E, view_stride, shape_stride, scheme, num_groups, weight, mode,
                            empty_fill, D, S, dtype, stream, 0, 0);
}

extern "C" int gv_view_pool_fuse_fwd_per_shape(const void* F, int32_t num_views, int32_t num_shapes, int64_t E,
                                               int64_t view_stride, int64_t shape_stride, const int32_t* scheme,
                                               int32_t num_groups, const float* weight, int32_t mode,
                                               float empty_fill, void* D, void* S, int32_t dtype, void* stream) {
    return pool_fuse_launch(F, num_views, num_shapes, E, view_stride, shape_stride, scheme, num_groups, weight, mode,
                            empty_fill, D, S, dtype, stream, (int64_t)num_groups * num_views, num_groups);
}

// sigmoid(log|r|) per image (the paper's per-shape scorer; model.py:147 without the batch mean of :146)
__global__ void view_score_per_shape_f32(const float* __restrict__ r_img, int nb, float* __restrict__ scores) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const float lg = logf(fabsf(r_img[b]));
    scores[b] = 1.0f / (1.0f + expf(-lg));
}

extern "C" int gv_view_score_per_shape(const float* r_img, int32_t nb, float* scores, void* stream) {
    if (!r_img || !scores || nb <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(view_score_per_shape_f32, dim3((nb + 255) / 256), dim3(256), 0, (hipStream_t)stream, r_img, nb,
                       scores);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

// one workgroup per shape: binning exactly as group_assign_kernel
__global__ __launch_bounds__(64) void group_assign_per_shape_kernel(const float* __restrict__ scores, int V, int G,
                                                                    int num_bins, int weight_mode,
                                                                    int* __restrict__ gidx, int* __restrict__ scheme,
                                                                    float* __restrict__ weight,
                                                                    int* __restrict__ status) {
    __shared__ int s_gidx[64];
    __shared__ float s_score[64];
    const int n = blockIdx.x, t = threadIdx.x;
    scores += (size_t)n * V;
    gidx += (size_t)n * V;
    scheme += (size_t)n * G * V;
    weight += (size_t)n * G;
    if (t < V) {
        const float sc = scores[t];
        const float prod = __fmul_rn(sc, (float)num_bins);
        int b = (int)prod;
        if (sc != sc) { atomicOr(status, 2); b = -1; }
        else if (b >= G || b < 0 || prod >= 2147483648.0f) { atomicOr(status, 1); if (prod >= 2147483648.0f) b = 0x7fffffff; }
        s_gidx[t] = b;
        s_score[t] = sc;
        gidx[t] = b;
    }
    __syncthreads();
    for (int i = t; i < G * V; i += 64) {
        const int g = i / V, v = i - g * V;
        scheme[i] = (s_gidx[v] == g) ? 1 : 0;
    }
    for (int g = t; g < G; g += 64) {
        int cnt = 0;
        float sum = 0.f;
        for (int v = 0; v < V; ++v)
            if (s_gidx[v] == g) { ++cnt; sum = __fadd_rn(sum, s_score[v]); }      // view order: reproducible
        weight[g] = weight_mode == GV_WEIGHT_COUNT ? (float)(1 + cnt) : (cnt ? __fdiv_rn(sum, (float)cnt) : 0.f);
    }
}

extern "C" int gv_group_assign_per_shape(const float* scores, int32_t num_shapes, int32_t num_views,
                                         int32_t num_groups, int32_t num_bins, int32_t weight_mode, int32_t* gidx,
                                         int32_t* scheme, float* weight, int32_t* status, void* stream) {
    if (!scores || !gidx || !scheme || !weight || !status) return GV_E_BADARG;
    if (num_shapes <= 0 || num_views <= 0 || num_groups <= 0 || num_bins <= 0) return GV_E_BADARG;
    if (weight_mode != GV_WEIGHT_COUNT && weight_mode != GV_WEIGHT_MEAN_SCORE) return GV_E_BADARG;
    if (num_views > 64 || num_groups > 64) return GV_E_UNSUPPORTED;
    GV_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(int32_t), (hipStream_t)stream));
    hipLaunchKernelGGL(group_assign_per_shape_kernel, dim3(num_shapes), dim3(64), 0, (hipStream_t)stream, scores,
                       num_views, num_groups, num_bins, weight_mode, gidx, scheme, weight, status);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

// eval.py:94-99: argmax, correct count, confusion matrix; one thread per shape
__global__ void eval_metrics_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int n,
                                    int c, long long* __restrict__ prediction, int* __restrict__ confusion,
                                    int* __restrict__ correct) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* row = logits + (size_t)i * c;
    int best = 0;
    float bv = row[0];
    for (int k = 1; k < c; ++k)
        if (row[k] > bv) { bv = row[k]; best = k; }               // strict >: the first maximum wins
    prediction[i] = best;
    const long long l = labels[i];
    if (l >= 0 && l < c) {
        atomicAdd(&confusion[(size_t)l * c + best], 1);
        if (l == best) atomicAdd(correct, 1);
    }
}

extern "C" int gv_eval_metrics(const float* logits, const int64_t* labels, int32_t n, int32_t c,
                               int64_t* prediction, int32_t* confusion, int32_t* correct, void* stream) {
    if (!logits || !labels || !prediction || !confusion || !correct || n <= 0 || c <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(eval_metrics_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, logits,
                       (const long long*)labels, n, c, (long long*)prediction, confusion, correct);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_dense_fwd(const float* x, int32_t n, int32_t f, const float* kernel,
                            const float* bias, int32_t c, float* y, void* stream) {
    if (!x || !kernel || !bias || !y || n <= 0 || f <= 0 || c <= 0) return GV_E_BADARG;
    if (c > 65535) return GV_E_UNSUPPORTED;
    hipLaunchKernelGGL(dense_f32, dim3(n, c), dim3(256), 0, (hipStream_t)stream, x, f, kernel, bias, c, y);
    GV_LAUNCH_CHECK();
    return GV_OK;
}
