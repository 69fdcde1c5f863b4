// train.hip — kernels that exist only for the training step of the hot path:
// batch-statistics BatchNorm grouped by view (the reference builds one graph copy per view, so every
// BN normalises over the N*h*w values of ONE view: nets/model.py:129-141 + slim.batch_norm with
// is_training=True), the backward of every forward op, the loss (train.py:145) and the Momentum
// update (train.py:171).  fp32 storage; reductions accumulate in fp64 (HBM-bound kernels, the fp64
// adds are free) so that per-(view, channel) sums over up to 1.5e5 pixels are reproducible to fp32.
#include <math.h>

#include "gv_common.h"
#include "lowp.h"

namespace {

// dst[p][c] += src[p][c]  (gradient fan-in of the residual add, nets/resnet_v2.py:91)
__global__ __launch_bounds__(256) void accumulate_f32(const float* __restrict__ src, int src_ld,
                                                      float* __restrict__ dst, int dst_ld, int64_t npix, int c) {
    const int64_t total = npix * c;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        const int64_t pix = idx / c;
        dst[pix * dst_ld + ch] += src[pix * src_ld + ch];
    }
}

// ---- grouped per-(group, channel) sums -----------------------------------------------------------
// acc[g][c][0] += sum u,  acc[g][c][1] += sum u*w   over the pixels of images b with b % G == g.
//   MODE 0 (BN forward stats):  u = z,            w = z                       -> sum z, sum z^2
//   MODE 1 (BN backward):       u = dy*[y>0],     w = (z-mean)*inv            -> sum g, sum g*zhat
//   MODE 2 (bias gradient):     u = dz,           w = 0
// Thread layout: VEC consecutive channels per thread (16-byte loads when VEC = 4), 64/VEC... threads span
// 64 channels, the remaining threads of the 256 are pixel lanes; grid (channel blocks, pixel splits, G).
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void grouped_sums_f32(const float* __restrict__ z, int z_ld,
                                                        const float* __restrict__ dy, int dy_ld,
                                                        const float* __restrict__ y, int y_ld,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ inv, int nb, int hw, int c,
                                                        int G, double* __restrict__ acc,
                                                        const float* __restrict__ scale = nullptr,
                                                        const float* __restrict__ shift = nullptr) {
    constexpr int TPC = 64 / VEC;                          // threads across the 64-channel block
    constexpr int PL = 256 / TPC;                          // pixel lanes
    const int cl = threadIdx.x % TPC;
    const int pl = threadIdx.x / TPC;
    const int ch = blockIdx.x * 64 + cl * VEC;
    const int g = blockIdx.z;
    const int nimg = (nb - g + G - 1) / G;                 // images of this group
    const int64_t npix = (int64_t)nimg * hw;
    const int64_t per = (npix + gridDim.y - 1) / gridDim.y;
    const int64_t p0 = (int64_t)blockIdx.y * per;
    const int64_t p1 = p0 + per < npix ? p0 + per : npix;
    double s0[VEC], s1[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s0[e] = s1[e] = 0.0;
    if (ch < c) {                                          // c % VEC == 0 (checked by the launcher)
        float mu[VEC], iv[VEC], sc[VEC], sh[VEC];
        const bool rmask = MODE == 1 && !y && scale;          // ReLU mask recomputed from z: [z*scale + shift > 0]
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            mu[e] = MODE == 1 ? mean[g * c + ch + e] : 0.f;
            iv[e] = MODE == 1 ? inv[g * c + ch + e] : 0.f;
            sc[e] = rmask ? scale[g * c + ch + e] : 0.f;
            sh[e] = rmask ? shift[g * c + ch + e] : 0.f;
        }
        // division-free walk over the group's pixels (images g, g+G, ... of the batch): one division to start
        int64_t pw = p0 + pl;
        int kw_ = (int)(pw / hw);
        int rw = (int)(pw - (int64_t)kw_ * hw);
        int64_t pix = (int64_t)(kw_ * G + g) * hw + rw;
        const int64_t step_img = (int64_t)(G - 1) * hw;
        for (int64_t p = pw; p < p1; p += PL) {
            float zv[VEC], gv[VEC], yv[VEC];
            if constexpr (VEC == 4) {
                if (MODE != 2) *reinterpret_cast<f32x4*>(zv) = *reinterpret_cast<const f32x4*>(z + pix * z_ld + ch);
                if (MODE != 0) *reinterpret_cast<f32x4*>(gv) = *reinterpret_cast<const f32x4*>(dy + pix * dy_ld + ch);
                if (MODE == 1 && y) *reinterpret_cast<f32x4*>(yv) = *reinterpret_cast<const f32x4*>(y + pix * y_ld + ch);
            } else {
                if (MODE != 2) zv[0] = z[pix * z_ld + ch];
                if (MODE != 0) gv[0] = dy[pix * dy_ld + ch];
                if (MODE == 1 && y) yv[0] = y[pix * y_ld + ch];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (MODE == 0) {
                    s0[e] += zv[e];
                    s1[e] += (double)zv[e] * zv[e];
                } else if (MODE == 1) {
                    float gr = gv[e];
                    if (y && !(yv[e] > 0.f)) gr = 0.f;
                    if (rmask && !(zv[e] * sc[e] + sh[e] > 0.f)) gr = 0.f;
                    s0[e] += gr;
                    s1[e] += (double)gr * ((zv[e] - mu[e]) * iv[e]);
                } else {
                    s0[e] += gv[e];
                }
            }
            pix += PL;
            if (step_img) {                                   // (one group: the pixels are contiguous, nothing to skip)
                rw += PL;
                while (rw >= hw) { rw -= hw; pix += step_img; }
            }
        }
    }
    __shared__ double red[2][PL][64];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        red[0][pl][cl * VEC + e] = s0[e];
        red[1][pl][cl * VEC + e] = s1[e];
    }
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < c) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < PL; ++q) { a += red[0][q][threadIdx.x]; b += red[1][q][threadIdx.x]; }
        const size_t o = ((size_t)g * c + blockIdx.x * 64 + threadIdx.x) * 2;
        // the block's partial on fixed grids (2^-40 / 2^-36 forward, 2^-52 backward: grouped_sums_v8 in train_lp.hip says
        // why): exact fp64 additions within their range, so the totals do not depend on the order the blocks arrive in
        const double q0 = MODE == 0 ? 1099511627776.0 : 4503599627370496.0, q1 = MODE == 0 ? 68719476736.0 : 4503599627370496.0;
        atomicAdd(&acc[o], rint(a * q0) / q0);
        if (MODE != 2) atomicAdd(&acc[o + 1], rint(b * q1) / q1);
    }
}

template <int MODE>
void launch_grouped_sums(dim3 grid, hipStream_t st, bool vec, const float* z, int z_ld, const float* dy, int dy_ld,
                         const float* y, int y_ld, const float* mean, const float* inv, int nb, int hw, int c, int G,
                         double* acc, const float* scale = nullptr, const float* shift = nullptr) {
    if (vec)
        hipLaunchKernelGGL((grouped_sums_f32<MODE, 4>), grid, dim3(256), 0, st, z, z_ld, dy, dy_ld, y, y_ld, mean, inv,
                           nb, hw, c, G, acc, scale, shift);
    else
        hipLaunchKernelGGL((grouped_sums_f32<MODE, 1>), grid, dim3(256), 0, st, z, z_ld, dy, dy_ld, y, y_ld, mean, inv,
                           nb, hw, c, G, acc, scale, shift);
}

inline bool vec_ok(const void* p, int ld) { return p == nullptr || ((((uintptr_t)p) & 15u) == 0 && (ld & 3) == 0); }

// mean/var (biased) + folded scale/shift per (group, channel)
__global__ void bn_finalize_grouped(const double* __restrict__ acc, int G, int c, const int* __restrict__ counts,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                    float* __restrict__ mean, float* __restrict__ var, float* __restrict__ inv,
                                    float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * c) return;
    const int g = i / c, ch = i - g * c;
    const double m = (double)counts[g];
    const double mu = acc[(size_t)i * 2] / m;
    double v = acc[(size_t)i * 2 + 1] / m - mu * mu;
    if (v < 0.0) v = 0.0;
    const float iv = (float)(1.0 / sqrt(v + (double)eps));
    const float ga = gamma ? gamma[ch] : 1.f;
    mean[i] = (float)mu;
    var[i] = (float)v;
    inv[i] = iv;
    scale[i] = iv * ga;
    shift[i] = beta[ch] - (float)mu * iv * ga;
}

// y = act(x*scale[g][c] + shift[g][c]),  g = image % G
__global__ __launch_bounds__(256) void scale_shift_act_grouped_f32(const float* __restrict__ x, int nb, int hw,
                                                                   int c, int x_ld,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, int G,
                                                                   int relu, float* __restrict__ y, int y_ld) {
    const int cg = c >> 2;
    const int64_t total = (int64_t)nb * hw * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        const int64_t pix = idx / cg;
        const int g = (int)((pix / hw) % G);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * x_ld + 4 * q);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (size_t)g * c + 4 * q);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (size_t)g * c + 4 * q);
        v = v * sc + sh;
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(y + pix * y_ld + 4 * q) = v;
    }
}

// dz += gamma*inv * (g - s1/m - zhat*s2/m),  g = dy*[y>0]   (train-mode BN + ReLU backward)
__global__ __launch_bounds__(256) void bn_bwd_apply_grouped_f32(
    const float* __restrict__ dy, int dy_ld, const float* __restrict__ y, int y_ld, const float* __restrict__ z,
    int z_ld, const float* __restrict__ mean, const float* __restrict__ inv, const float* __restrict__ gamma,
    const double* __restrict__ acc, const int* __restrict__ counts, int nb, int hw, int c, int G,
    float* __restrict__ dz, int dz_ld, const float* __restrict__ scale = nullptr,
    const float* __restrict__ shift = nullptr, int accumulate = 1) {
    const int64_t total = (int64_t)nb * hw * c;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        const int64_t pix = idx / c;
        const int g = (int)((pix / hw) % G);
        const int gi = g * c + ch;
        float gr = dy[pix * dy_ld + ch];
        if (y && !(y[pix * y_ld + ch] > 0.f)) gr = 0.f;
        const float zval = z[pix * z_ld + ch];
        if (!y && scale && !(zval * scale[gi] + shift[gi] > 0.f)) gr = 0.f;
        const float iv = inv[gi];
        const float zh = (zval - mean[gi]) * iv;
        const float m = (float)counts[g];
        const float s1 = (float)acc[(size_t)gi * 2], s2 = (float)acc[(size_t)gi * 2 + 1];
        const float coef = (gamma ? gamma[ch] : 1.f) * iv;
        const float upd = coef * (gr - s1 / m - zh * s2 / m);
        dz[pix * dz_ld + ch] = accumulate ? dz[pix * dz_ld + ch] + upd : upd;
    }
}

// per-channel parameter gradients: dbeta[c] += sum_g acc[g][c][0], dgamma[c] += sum_g acc[g][c][1]
__global__ void bn_param_grads(const double* __restrict__ acc, int G, int c, float* __restrict__ dbeta,
                               float* __restrict__ dgamma) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double a = 0.0, b = 0.0;
    for (int g = 0; g < G; ++g) { a += acc[((size_t)g * c + ch) * 2]; b += acc[((size_t)g * c + ch) * 2 + 1]; }
    if (dbeta) dbeta[ch] += (float)a;
    if (dgamma) dgamma[ch] += (float)b;
}

// ---- pooling backward ------------------------------------------------------------------------------
// max: the gradient of a window goes to its first maximum in scan order (tf MaxPoolGrad / torch);
// scatter with atomics because 3x3/2 windows overlap.  avg: dy / (#valid taps) to every valid tap.
__global__ __launch_bounds__(256) void pool2d_bwd_f32(const float* __restrict__ x, int x_ld,
                                                      const float* __restrict__ dy, int dy_ld, int nb, int ih,
                                                      int iw, int c, int kh, int kw, int stride, int pad_t,
                                                      int pad_l, int oh, int ow, int mode,
                                                      float* __restrict__ dx, int dx_ld) {
    const int64_t total = (int64_t)nb * oh * ow * c;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        const int64_t pix = idx / c;
        const int ox = (int)(pix % ow);
        const int64_t t = pix / ow;
        const int oy = (int)(t % oh);
        const int n = (int)(t / oh);
        const float g = dy[pix * dy_ld + ch];
        if (mode == GV_POOL_MAX) {
            float best = -INFINITY;
            int64_t arg = -1;
            for (int r = 0; r < kh; ++r) {
                const int iy = oy * stride + r - pad_t;
                if ((unsigned)iy >= (unsigned)ih) continue;
                for (int s = 0; s < kw; ++s) {
                    const int ix = ox * stride + s - pad_l;
                    if ((unsigned)ix >= (unsigned)iw) continue;
                    const int64_t ip = ((int64_t)n * ih + iy) * iw + ix;
                    const float v = x[ip * x_ld + ch];
                    if (v > best || arg < 0) { best = v; arg = ip; }
                }
            }
            if (arg >= 0) atomicAdd(&dx[arg * dx_ld + ch], g);
        } else {
            int cnt = 0;
            for (int r = 0; r < kh; ++r) {
                const int iy = oy * stride + r - pad_t;
                if ((unsigned)iy >= (unsigned)ih) continue;
                for (int s = 0; s < kw; ++s) cnt += (unsigned)(ox * stride + s - pad_l) < (unsigned)iw;
            }
            const float gv = g / (float)cnt;
            for (int r = 0; r < kh; ++r) {
                const int iy = oy * stride + r - pad_t;
                if ((unsigned)iy >= (unsigned)ih) continue;
                for (int s = 0; s < kw; ++s) {
                    const int ix = ox * stride + s - pad_l;
                    if ((unsigned)ix >= (unsigned)iw) continue;
                    atomicAdd(&dx[(((int64_t)n * ih + iy) * iw + ix) * dx_ld + ch], gv);
                }
            }
        }
    }
}

// ---- grouping module backward (nets/model.py:44-102) ------------------------------------------------
// dF[v] += w_g/sum(w) * dS / (#views of g tied at the maximum) for every view of g attaining the maximum
// (tf.reduce_max splits the gradient equally among ties); mean pooling: w_g/sum(w) * dS / |g|.
__global__ __launch_bounds__(256) void view_pool_fuse_bwd_f32(
    const float* __restrict__ F, const float* __restrict__ dS, int V, int N, int64_t E, int64_t view_stride,
    int64_t shape_stride, const int* __restrict__ scheme, int G, const float* __restrict__ weight, int mode,
    float* __restrict__ dF, int64_t scheme_stride, int64_t weight_stride) {
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
        for (int g = 0; g < G; ++g) ws += s_w[g];
        s_wsum = ws;
    }
    __syncthreads();
    if (s_wsum == 0.f) return;                      // (per-shape, mean-score weights) S = 0 for this shape: no gradient
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const size_t base = (size_t)n * shape_stride + e;
        const float ds = dS[(size_t)n * E + e];
        for (int g = 0; g < G; ++g) {
            const unsigned long long m0 = s_mask[g];
            if (m0 == 0) continue;
            const float coef = s_w[g] / s_wsum * ds;
            if (mode == GV_VIEWPOOL_MEAN) {
                const float each = coef / (float)__popcll(m0);
                for (unsigned long long m = m0; m; m &= m - 1)
                    dF[base + (size_t)(__ffsll((long long)m) - 1) * view_stride] += each;
            } else {
                float best = -INFINITY;
                for (unsigned long long m = m0; m; m &= m - 1)
                    best = fmaxf(best, F[base + (size_t)(__ffsll((long long)m) - 1) * view_stride]);
                int ties = 0;
                for (unsigned long long m = m0; m; m &= m - 1)
                    ties += F[base + (size_t)(__ffsll((long long)m) - 1) * view_stride] == best;
                const float each = coef / (float)ties;
                for (unsigned long long m = m0; m; m &= m - 1) {
                    const size_t a = base + (size_t)(__ffsll((long long)m) - 1) * view_stride;
                    if (F[a] == best) dF[a] += each;
                }
            }
        }
    }
}

// dx[b][p][c] += dgap[b][c] / hw
__global__ __launch_bounds__(256) void global_avg_pool_bwd_f32(const float* __restrict__ dgap, int nb, int hw,
                                                               int c, float* __restrict__ dx, int dx_ld) {
    const int64_t total = (int64_t)nb * hw * c;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        const int64_t pix = idx / c;
        dx[pix * dx_ld + ch] += dgap[(pix / hw) * c + ch] / (float)hw;
    }
}

// mean sparse-softmax cross-entropy (train.py:145): loss, dlogits = (softmax - onehot)/N; one block
__global__ __launch_bounds__(256) void softmax_ce_f32(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                      int n, int c, float* __restrict__ loss,
                                                      float* __restrict__ dlogits) {
    __shared__ float part[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float* l = logits + (size_t)i * c;
        float mx = -INFINITY;
        for (int j = 0; j < c; ++j) mx = fmaxf(mx, l[j]);
        float se = 0.f;
        for (int j = 0; j < c; ++j) se += expf(l[j] - mx);
        const float lse = logf(se) + mx;
        // a label outside [0, c) (or an int64 that does not fit an int) never indexes the logits: like TensorFlow's
        // GPU kernel the row's loss and gradient become NaN, which the trainer's check_numerics (train.py:175) raises on
        const int64_t lab64 = labels[i];
        const bool ok = lab64 >= 0 && lab64 < (int64_t)c;
        const int lab = ok ? (int)lab64 : 0;
        acc += ok ? lse - l[lab] : NAN;
        for (int j = 0; j < c; ++j)
            dlogits[(size_t)i * c + j] = ok ? (expf(l[j] - lse) - (j == lab ? 1.f : 0.f)) / (float)n : NAN;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = part[0] / (float)n;
}

// Dense backward: dx[n][f] = sum_c dy[n][c] W[f][c];  dW[f][c] += sum_n x[n][f] dy[n][c];  db[c] += sum_n dy[n][c]
__global__ __launch_bounds__(256) void dense_bwd_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ W, int n, int f, int c,
                                                     float* __restrict__ dx, float* __restrict__ dW,
                                                     float* __restrict__ db) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)n * f) {
        const int row = (int)(i / f), col = (int)(i % f);
        float s = 0.f;
        for (int j = 0; j < c; ++j) s += dy[(size_t)row * c + j] * W[(size_t)col * c + j];
        dx[i] = s;
    }
    if (i < (int64_t)f * c) {
        const int ff = (int)(i / c), cc = (int)(i % c);
        float s = 0.f;
        for (int r = 0; r < n; ++r) s += x[(size_t)r * f + ff] * dy[(size_t)r * c + cc];
        dW[i] += s;
    }
    if (i < c) {
        float s = 0.f;
        for (int r = 0; r < n; ++r) s += dy[(size_t)r * c + i];
        db[i] += s;
    }
}

// MomentumOptimizer(lr, 0.9), non-Nesterov, with the slim L2 term wd*w added to the gradient
// (train.py:171, train_utils.py:121-161):  m = mu*m + (g + wd*w);  w -= lr*m
__global__ void sgd_momentum_f32(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                 int64_t n, float lr, float mu, float wd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mm = mu * m[i] + (g[i] + wd * w[i]);
    m[i] = mm;
    w[i] -= lr * mm;
}

// ---- filter gradient: dW[r][s][ci][co] += sum_m X[shift_{r,s}(m)][ci] * dZ[m][co] -------------------
// One TN GEMM per filter tap between two pixel-major matrices (the shifted input and dZ), reduction
// over the output pixels, on the exact fp32 MFMA (32x32x2).  A workgroup owns a 64(ci) x 64(co) tile of
// one tap and a slice of the pixels; slices are combined with fp32 atomic adds (the tile is 16 KB).
// LDS keeps both operands pixel-major, which is exactly the k-major image the 32x32x2 operand map
// wants (lane i reads element i of pixel row m): conflict-free ds_read_b32, no transposes.
__global__ __launch_bounds__(256) void conv_wgrad_f32(const float* __restrict__ x, int x_ld,
                                                      const float* __restrict__ dz, int dz_ld, int nb, int ih,
                                                      int iw, int cin, int kh, int kw, int stride, int pad_t,
                                                      int pad_l, int oh, int ow, int cout, int64_t M,
                                                      int64_t m_per_block, const GvDw dw) {
    constexpr int PT = 32;                              // pixels per LDS tile
    __shared__ __attribute__((aligned(16))) float sX[PT][64 + 4];
    __shared__ __attribute__((aligned(16))) float sZ[PT][64 + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;            // 2x2 waves, 32x32 each
    const int ntile_co = (cout + 63) / 64, ntile_ci = (cin + 63) / 64;
    int b = blockIdx.x;
    const int tco = b % ntile_co; b /= ntile_co;
    const int tci = b % ntile_ci; b /= ntile_ci;
    const int tap = b;                                  // r*kw + s
    const int fr = tap / kw, fs = tap - fr * kw;
    const int ci0 = tci * 64, co0 = tco * 64;
    const int64_t m0 = (int64_t)blockIdx.y * m_per_block;
    const int64_t m1 = m0 + m_per_block < M ? m0 + m_per_block : M;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ohow = oh * ow;
    const bool xvec = (x_ld & 3) == 0 && ((((uintptr_t)x) & 15u) == 0);      // 16-byte loads when rows are aligned
    const bool zvec = (dz_ld & 3) == 0 && ((((uintptr_t)dz) & 15u) == 0);
    // loader: thread -> (pixel row p = tid/16 (+16), 4 channels q = tid%16)
    const int lp = tid >> 4, lq = (tid & 15) * 4;
    for (int64_t mt = m0; mt < m1; mt += PT) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = lp + 16 * h;
            const int64_t m = mt + p;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f};
            if (m < m1) {
                const int n = (int)(m / ohow);
                const int rem = (int)(m - (int64_t)n * ohow);
                const int oy = rem / ow, ox = rem - oy * ow;
                const int iy = oy * stride + fr - pad_t, ix = ox * stride + fs - pad_l;
                if ((unsigned)iy < (unsigned)ih && (unsigned)ix < (unsigned)iw) {
                    const float* xp = x + (((size_t)n * ih + iy) * iw + ix) * x_ld + ci0 + lq;
                    if (xvec && ci0 + lq + 3 < cin) {
                        xv = *reinterpret_cast<const f32x4*>(xp);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (ci0 + lq + e < cin) xv[e] = xp[e];
                    }
                }
                const float* zp = dz + (size_t)m * dz_ld + co0 + lq;
                if (zvec && co0 + lq + 3 < cout) {
                    zv = *reinterpret_cast<const f32x4*>(zp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (co0 + lq + e < cout) zv[e] = zp[e];
                }
            }
            *reinterpret_cast<f32x4*>(&sX[p][lq]) = xv;
            *reinterpret_cast<f32x4*>(&sZ[p][lq]) = zv;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PT; k += 2) {
            const float av = sX[k + (lane >> 5)][wi * 32 + (lane & 31)];
            const float bv = sZ[k + (lane >> 5)][wj * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = co0 + wj * 32 + (lane & 31);
    if (col < cout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (ci < cin) gv_dw_put(dw, blockIdx.y, ((size_t)tap * cin + ci) * cout + col, acc[r]);
        }
    }
}

inline unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int64_t cap = 256 * 16;
    return (unsigned)(b < cap ? (b > 0 ? b : 1) : cap);
}


// Storage-typed loads of the filter-gradient kernels: S = float, __bf16 or _Float16 in HBM, fp32 in registers.
template <typename S>
__device__ __forceinline__ f32x4 ld4(const S* p) {
    if constexpr (sizeof(S) == 4) {
        return *reinterpret_cast<const f32x4*>(p);
    } else {
        const uint2 raw = *reinterpret_cast<const uint2*>(p);
        f32x4 v;
        v[0] = (float)__builtin_bit_cast(S, (unsigned short)(raw.x & 0xffffu));
        v[1] = (float)__builtin_bit_cast(S, (unsigned short)(raw.x >> 16));
        v[2] = (float)__builtin_bit_cast(S, (unsigned short)(raw.y & 0xffffu));
        v[3] = (float)__builtin_bit_cast(S, (unsigned short)(raw.y >> 16));
        return v;
    }
}
template <typename S>
__device__ __forceinline__ bool ld4_ok(const S* p, int ld) {
    return (ld & 3) == 0 && ((((uintptr_t)p) & (4 * sizeof(S) - 1)) == 0);
}

// Filter gradient, second generation: workgroup tile (64*TI) input channels x (64*TO) output channels for one
// filter tap, 2x2 waves of TI x TO MFMA tiles (v_mfma_f32_32x32x2_f32: exact fp32; the k axis of this GEMM is
// the PIXEL axis, and pixel-major LDS rows are exactly the k-major operand image this instruction wants:
// lane (i, h) reads element [pixel k + h][channel i], 32 consecutive dwords per half-wave).  16 pixels per
// step, LDS double buffered, the next step's global loads are issued before this step's MFMAs, one barrier
// per step.  Pixel slices (blockIdx.y) are combined with fp32 atomics.
template <typename S, int TI, int TO>
__global__ __launch_bounds__(256) void conv_wgrad2_f32(const S* __restrict__ x, int x_ld,
                                                       const S* __restrict__ dz, int dz_ld, int nb, int ih,
                                                       int iw, int cin, int kh, int kw, int stride, int pad_t,
                                                       int pad_l, int oh, int ow, int cout, int64_t M,
                                                       int64_t m_per_block, const GvDw dw) {
    constexpr int PT = (TI * TO == 1) ? 32 : 16, BI = 64 * TI, BO = 64 * TO;   // pixels per step
    constexpr int XV = PT * BI / 4 / 256, ZV = PT * BO / 4 / 256;      // float4 loads per thread and step
    __shared__ __attribute__((aligned(16))) float sX[2][PT][BI];
    __shared__ __attribute__((aligned(16))) float sZ[2][PT][BO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int ntile_co = (cout + BO - 1) / BO, ntile_ci = (cin + BI - 1) / BI;
    // 1-D grid, XCD-aware: the tiles of one pixel slice re-read the same rows, so they share an XCD's L2
    const int tiles = ntile_co * ntile_ci * kh * kw;
    const int logical = gv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int b = logical % tiles;
    const int tco = b % ntile_co; b /= ntile_co;
    const int tci = b % ntile_ci; b /= ntile_ci;
    const int tap = b;
    const int fr = tap / kw, fs = tap - fr * kw;
    const int ci0 = tci * BI, co0 = tco * BO;
    const int64_t m0 = (int64_t)(logical / tiles) * m_per_block;
    const int64_t m1 = m0 + m_per_block < M ? m0 + m_per_block : M;
    f32x16 acc[TI][TO];
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int u = 0; u < TO; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const int ohow = oh * ow;
    const bool xvec = (cin & 3) == 0 && ld4_ok(x, x_ld);
    const bool zvec = (cout & 3) == 0 && ld4_ok(dz, dz_ld);
    f32x4 xr[XV], zr[ZV];
    auto load = [&](int64_t mt) {
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int idx = tid + j * 256;
            const int p = idx / (BI / 4), c = ci0 + (idx % (BI / 4)) * 4;
            const int64_t m = mt + p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m1 && c < cin) {
                const int n = (int)(m / ohow);
                const int rem = (int)(m - (int64_t)n * ohow);
                const int oy = rem / ow, ox = rem - oy * ow;
                const int iy = oy * stride + fr - pad_t, ix = ox * stride + fs - pad_l;
                if ((unsigned)iy < (unsigned)ih && (unsigned)ix < (unsigned)iw) {
                    const S* xp = x + (((size_t)n * ih + iy) * iw + ix) * x_ld + c;
                    if (xvec) {
                        v = ld4(xp);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (c + e < cin) v[e] = (float)xp[e];
                    }
                }
            }
            xr[j] = v;
        }
#pragma unroll
        for (int j = 0; j < ZV; ++j) {
            const int idx = tid + j * 256;
            const int p = idx / (BO / 4), c = co0 + (idx % (BO / 4)) * 4;
            const int64_t m = mt + p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m1 && c < cout) {
                const S* zp = dz + (size_t)m * dz_ld + c;
                if (zvec) {
                    v = ld4(zp);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (c + e < cout) v[e] = (float)zp[e];
                }
            }
            zr[j] = v;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int idx = tid + j * 256;
            *reinterpret_cast<f32x4*>(&sX[buf][idx / (BI / 4)][(idx % (BI / 4)) * 4]) = xr[j];
        }
#pragma unroll
        for (int j = 0; j < ZV; ++j) {
            const int idx = tid + j * 256;
            *reinterpret_cast<f32x4*>(&sZ[buf][idx / (BO / 4)][(idx % (BO / 4)) * 4]) = zr[j];
        }
    };
    load(m0);
    store(0);
    __syncthreads();
    int buf = 0;
    const int li = lane & 31, lh = lane >> 5;
    for (int64_t mt = m0; mt < m1; mt += PT) {
        const bool more = mt + PT < m1;
        if (more) load(mt + PT);
#pragma unroll
        for (int k = 0; k < PT; k += 2) {
            float av[TI], bv[TO];
#pragma unroll
            for (int t = 0; t < TI; ++t) av[t] = sX[buf][k + lh][(wi * TI + t) * 32 + li];
#pragma unroll
            for (int u = 0; u < TO; ++u) bv[u] = sZ[buf][k + lh][(wj * TO + u) * 32 + li];
#pragma unroll
            for (int t = 0; t < TI; ++t)
#pragma unroll
                for (int u = 0; u < TO; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[u], acc[t][u], 0, 0, 0);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int u = 0; u < TO; ++u) {
        const int col = co0 + (wj * TO + u) * 32 + li;
        if (col >= cout) continue;
#pragma unroll
        for (int t = 0; t < TI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (wi * TI + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < cin) gv_dw_put(dw, logical / tiles, ((size_t)tap * cin + ci) * cout + col, acc[t][u][r]);
            }
    }
}


// Filter gradient of the few-channel layers (the stems: kh*kw*cin <= 288 im2col rows).  The tap-per-workgroup
// kernels above re-read x and dz once per filter tap and waste most of a 64-row tile on 3 or 32 input channels
// (Conv2d_1a/2a/2b: 12 of the 41 ms of filter gradients per step).  Here one WAVE owns a contiguous pixel range and
// ALL im2col rows rho = tap*cin + ci (NRT tiles of 32 rows) x one 32-column tile of cout; the MFMA operands are
// fetched straight from global memory — the k axis of this GEMM is the pixel axis, so lane (i, h) needs
// x[pixel k+h shifted by tap(i)][ci(i)] and dz[pixel k+h][co i]: 128-byte coalesced rows, no LDS, no barrier.
// Loads run U pixel pairs ahead of the MFMAs that consume them.  Waves are combined with fp32 atomics.
template <typename S, int NRT, int U>
__global__ __launch_bounds__(256) void conv_wgrad_direct_f32(const S* __restrict__ x, int x_ld,
                                                             const S* __restrict__ dz, int dz_ld, int ih, int iw,
                                                             int cin, int kh, int kw, int stride, int pad_t,
                                                             int pad_l, int oh, int ow, int cout, int64_t M,
                                                             int64_t m_per_wave, const GvDw dw) {
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    // column tile fastest: the workgroups that share a pixel range (and re-read the same x) run side by side
    const int nct = (cout + 31) / 32;
    const int64_t w = (int64_t)(blockIdx.x / nct) * 4 + (threadIdx.x >> 6);
    const int co = (blockIdx.x % nct) * 32 + li;
    const int64_t m0 = w * m_per_wave;
    const int64_t m1 = m0 + m_per_wave < M ? m0 + m_per_wave : M;
    if (m0 >= M) return;
    const int R = kh * kw * cin;
    int fr[NRT], fs[NRT], delta[NRT];
    bool rv[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        const int rho = t * 32 + li;
        rv[t] = rho < R;
        const int tap = rv[t] ? rho / cin : 0;
        const int ci = rv[t] ? rho - tap * cin : 0;
        fr[t] = tap / kw;
        fs[t] = tap - fr[t] * kw;
        delta[t] = (fr[t] * iw + fs[t]) * x_ld + ci;
    }
    f32x16 acc[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // this lane's pixel of the current pair: m = m0 + 2*pair + lh
    int64_t m = m0 + lh;
    int n = (int)(m / ((int64_t)oh * ow));
    int rem = (int)(m - (int64_t)n * oh * ow);
    int oy = rem / ow, ox = rem - oy * ow;
    S av[2][U][NRT], bv[2][U];             // RAW storage values: converting at the load site would wait for each load
                                          // (16-bit loads) and serialise the prefetch; the conversion happens at the MFMA
    auto fetch = [&](int s) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool pok = m < m1;
            const int iy0 = oy * stride - pad_t, ix0 = ox * stride - pad_l;
            const int base = ((n * ih + iy0) * iw + ix0) * x_ld;          // < 2^31 elements (checked by the launcher)
            // (no branch around the loads: a join would force vmcnt(0) and serialise the prefetch)
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                const bool ok = pok && rv[t] && (unsigned)(iy0 + fr[t]) < (unsigned)ih &&
                                (unsigned)(ix0 + fs[t]) < (unsigned)iw;
                av[s][u][t] = ok ? x[base + delta[t]] : (S)0.f;
            }
            bv[s][u] = (pok && co < cout) ? dz[m * dz_ld + co] : (S)0.f;
            m += 2;
            ox += 2;
            while (ox >= ow) {
                ox -= ow;
                if (++oy == oh) { oy = 0; ++n; }
            }
        }
    };
    auto consume = [&](int s) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NRT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)av[s][u][t], (float)bv[s][u], acc[t], 0, 0, 0);
    };
    const int64_t pairs = (m1 - m0 + 1) / 2;
    const int64_t stages = (pairs + U - 1) / U;
    fetch(0);
    for (int64_t st = 0; st + 1 < stages; st += 2) {     // two stages per trip: static register sets
        fetch(1);
        consume(0);
        if (st + 2 < stages) fetch(0);
        consume(1);
    }
    if (stages & 1) consume(0);
    if (co < cout) {
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rho = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (rho < R) gv_dw_put(dw, w, (size_t)rho * cout + co, acc[t][r]);
            }
    }
}


__global__ void bn_update_moving(const float* __restrict__ mean, const float* __restrict__ var,
                                 const int* __restrict__ counts, int G, int c, float decay,
                                 float* __restrict__ mm, float* __restrict__ mv) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float m = mm[ch], v = mv[ch];
    for (int g = 0; g < G; ++g) {                         // one update per view graph copy, in view order
        const float n = (float)counts[g];
        const float unb = n > 1.f ? n / (n - 1.f) : 1.f;
        m = __fadd_rn(__fmul_rn(m, decay), __fmul_rn(mean[(size_t)g * c + ch], 1.f - decay));
        v = __fadd_rn(__fmul_rn(v, decay), __fmul_rn(__fmul_rn(var[(size_t)g * c + ch], unb), 1.f - decay));
    }
    mm[ch] = m;
    mv[ch] = v;
}

// every BatchNorm layer's moving-average update in ONE launch (block -> job through a table)
__global__ void bn_update_moving_batched(const gv_bn_moving_job* __restrict__ jobs, const int* __restrict__ block_job,
                                         int G, float decay) {
    const gv_bn_moving_job j = jobs[block_job[blockIdx.x]];
    const int ch = (blockIdx.x - j.first_block) * blockDim.x + threadIdx.x;
    if (ch >= j.c) return;
    float m = j.moving_mean[ch], v = j.moving_var[ch];
    const size_t ld = j.ld > 0 ? (size_t)j.ld : (size_t)j.c;
    for (int g = 0; g < G; ++g) {
        const float n = (float)j.counts[g];
        const float unb = n > 1.f ? n / (n - 1.f) : 1.f;
        m = __fadd_rn(__fmul_rn(m, decay), __fmul_rn(j.mean[g * ld + ch], 1.f - decay));
        v = __fadd_rn(__fmul_rn(v, decay), __fmul_rn(__fmul_rn(j.var[g * ld + ch], unb), 1.f - decay));
    }
    j.moving_mean[ch] = m;
    j.moving_var[ch] = v;
}

}  // namespace

extern "C" int gv_bn_update_moving_batched(const gv_bn_moving_job* jobs_dev, int32_t num_jobs,
                                           const int32_t* block_job_dev, int32_t num_blocks, int32_t num_groups,
                                           float decay, void* stream) {
    if (!jobs_dev || !block_job_dev || num_jobs <= 0 || num_blocks <= 0 || num_groups <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(bn_update_moving_batched, dim3((unsigned)num_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev,
                       block_job_dev, num_groups, decay);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_bn_sums_grouped(const float* z, int32_t nb, int32_t hw, int32_t c, int32_t z_ld,
                                  int32_t num_groups, double* accum, void* stream) {
    if (!z || !accum) return GV_E_BADARG;
    if (nb <= 0 || hw <= 0 || c <= 0 || z_ld < c || num_groups <= 0 || nb % num_groups != 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    GV_HIP_CHECK(hipMemsetAsync(accum, 0, sizeof(double) * 2 * (size_t)num_groups * c, st));
    const int64_t npix = (int64_t)(nb / num_groups) * hw;
    int splits = (int)((npix + 2047) / 2048);
    {   // at least ~1024 workgroups on the small late layers (latency-bound), down to 128 pixels per workgroup
        const int64_t want = 1024 / ((int64_t)((c + 63) / 64) * num_groups) + 1, most = (npix + 127) / 128;
        if (splits < want) splits = (int)(want < most ? want : most);
    }
    if (splits > 256) splits = 256;
    // (the streaming kernels of train_lp.hip, instantiated for fp32, when the shape is 16-byte vectorisable)
    if (gvlp::grouped_sums(GV_F32, 0, z, z_ld, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nb, hw, c,
                           num_groups, splits, accum, st) == GV_OK)
        return GV_OK;
    launch_grouped_sums<0>(dim3((c + 63) / 64, splits, num_groups), st, (c & 3) == 0 && vec_ok(z, z_ld), z, z_ld,
                           nullptr, 0, nullptr, 0, nullptr, nullptr, nb, hw, c, num_groups, accum);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_bn_finalize_grouped(const double* accum, int32_t c, int32_t num_groups, const int32_t* counts,
                                      const float* gamma, const float* beta, float eps, float* mean, float* var,
                                      float* inv, float* scale, float* shift, void* stream) {
    if (!accum || !counts || !beta || !mean || !var || !inv || !scale || !shift || c <= 0 || num_groups <= 0)
        return GV_E_BADARG;
    hipLaunchKernelGGL(bn_finalize_grouped, dim3((num_groups * c + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       accum, num_groups, c, counts, gamma, beta, eps, mean, var, inv, scale, shift);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_bn_stats_grouped(const float* z, int32_t nb, int32_t hw, int32_t c, int32_t z_ld,
                                   int32_t num_groups, const int32_t* counts, const float* gamma,
                                   const float* beta, float eps, double* accum, float* mean, float* var,
                                   float* inv, float* scale, float* shift, void* stream) {
    const int rc = gv_bn_sums_grouped(z, nb, hw, c, z_ld, num_groups, accum, stream);
    if (rc != GV_OK) return rc;
    return gv_bn_finalize_grouped(accum, c, num_groups, counts, gamma, beta, eps, mean, var, inv, scale, shift, stream);
}

extern "C" int gv_scale_shift_act_grouped(const float* x, int32_t nb, int32_t hw, int32_t c, int32_t x_ld,
                                          const float* scale, const float* shift, int32_t num_groups,
                                          int32_t relu, float* y, int32_t y_ld, void* stream) {
    if (!x || !y || !scale || !shift || nb <= 0 || hw <= 0 || c <= 0 || x_ld < c || y_ld < c || num_groups <= 0)
        return GV_E_BADARG;
    if ((c & 3) || (x_ld & 3) || (y_ld & 3) || !gv_aligned16(x) || !gv_aligned16(y) || !gv_aligned16(scale) ||
        !gv_aligned16(shift))
        return GV_E_ALIGN;
    if (gvlp::scale_shift_act_grouped(GV_F32, x, nb, hw, c, x_ld, scale, shift, num_groups, relu, y, y_ld,
                                      (hipStream_t)stream) == GV_OK)
        return GV_OK;
    hipLaunchKernelGGL(scale_shift_act_grouped_f32, dim3(grid_for((int64_t)nb * hw * (c / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, nb, hw, c, x_ld, scale, shift, num_groups, relu, y, y_ld);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

static int bn_bwd_sums_f32(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld, const float* z, int32_t z_ld,
                           const float* mean, const float* inv, int32_t nb, int32_t hw, int32_t c, int32_t num_groups,
                           double* accum, const float* scale, const float* shift, void* stream);

extern "C" int gv_bn_relu_bwd_sums_grouped(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld,
                                           const float* z, int32_t z_ld, const float* mean, const float* inv,
                                           int32_t nb, int32_t hw, int32_t c, int32_t num_groups, double* accum,
                                           void* stream) {
    return bn_bwd_sums_f32(dy, dy_ld, y, y_ld, z, z_ld, mean, inv, nb, hw, c, num_groups, accum, nullptr, nullptr, stream);
}

static int bn_bwd_sums_f32(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld, const float* z, int32_t z_ld,
                           const float* mean, const float* inv, int32_t nb, int32_t hw, int32_t c, int32_t num_groups,
                           double* accum, const float* scale, const float* shift, void* stream) {
    if (!dy || !z || !mean || !inv || !accum) return GV_E_BADARG;
    if (nb <= 0 || hw <= 0 || c <= 0 || num_groups <= 0 || nb % num_groups != 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    GV_HIP_CHECK(hipMemsetAsync(accum, 0, sizeof(double) * 2 * (size_t)num_groups * c, st));
    const int64_t npix = (int64_t)(nb / num_groups) * hw;
    int splits = (int)((npix + 2047) / 2048);
    {   // at least ~1024 workgroups on the small late layers (latency-bound), down to 128 pixels per workgroup
        const int64_t want = 1024 / ((int64_t)((c + 63) / 64) * num_groups) + 1, most = (npix + 127) / 128;
        if (splits < want) splits = (int)(want < most ? want : most);
    }
    if (splits > 256) splits = 256;
    if (gvlp::grouped_sums(GV_F32, 1, z, z_ld, dy, dy_ld, y, y_ld, mean, inv, scale, shift, nb, hw, c, num_groups, splits,
                           accum, st) == GV_OK)
        return GV_OK;
    launch_grouped_sums<1>(dim3((c + 63) / 64, splits, num_groups), st,
                           (c & 3) == 0 && vec_ok(z, z_ld) && vec_ok(dy, dy_ld) && vec_ok(y, y_ld), z, z_ld, dy, dy_ld, y,
                           y_ld, mean, inv, nb, hw, c, num_groups, accum, scale, shift);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

static int bn_bwd_apply_f32(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld, const float* z, int32_t z_ld,
                            const float* mean, const float* inv, const float* gamma, const int32_t* counts, int32_t nb,
                            int32_t hw, int32_t c, int32_t num_groups, const double* accum, float* dz, int32_t dz_ld,
                            float* dbeta, float* dgamma, const float* scale, const float* shift, int accumulate,
                            void* stream) {
    if (!dy || !z || !mean || !inv || !counts || !accum || !dz) return GV_E_BADARG;
    if (nb <= 0 || hw <= 0 || c <= 0 || num_groups <= 0 || nb % num_groups != 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    {
        bool done = false;
        if (gvlp::bn_bwd_apply_grouped(GV_F32, dy, dy_ld, y, y_ld, z, z_ld, mean, inv, gamma, accum, counts, scale, shift,
                                       accumulate, nb, hw, c, num_groups, dz, dz_ld, dbeta, dgamma, &done, st) == GV_OK)
            return GV_OK;
    }
    hipLaunchKernelGGL(bn_bwd_apply_grouped_f32, dim3(grid_for((int64_t)nb * hw * c)), dim3(256), 0, st, dy, dy_ld,
                       y, y_ld, z, z_ld, mean, inv, gamma, accum, counts, nb, hw, c, num_groups, dz, dz_ld, scale, shift, accumulate);
    if (dbeta || dgamma)
        hipLaunchKernelGGL(bn_param_grads, dim3((c + 255) / 256), dim3(256), 0, st, accum, num_groups, c, dbeta,
                           dgamma);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_bn_relu_bwd_apply_grouped(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld,
                                            const float* z, int32_t z_ld, const float* mean, const float* inv,
                                            const float* gamma, const int32_t* counts, int32_t nb, int32_t hw,
                                            int32_t c, int32_t num_groups, const double* accum, float* dz,
                                            int32_t dz_ld, float* dbeta, float* dgamma, void* stream) {
    return bn_bwd_apply_f32(dy, dy_ld, y, y_ld, z, z_ld, mean, inv, gamma, counts, nb, hw, c, num_groups, accum, dz, dz_ld,
                            dbeta, dgamma, nullptr, nullptr, 1, stream);
}

extern "C" int gv_bn_relu_bwd_grouped(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld,
                                      const float* z, int32_t z_ld, const float* mean, const float* inv,
                                      const float* gamma, const int32_t* counts, int32_t nb, int32_t hw,
                                      int32_t c, int32_t num_groups, double* accum, float* dz, int32_t dz_ld,
                                      float* dbeta, float* dgamma, void* stream) {
    const int rc = gv_bn_relu_bwd_sums_grouped(dy, dy_ld, y, y_ld, z, z_ld, mean, inv, nb, hw, c, num_groups, accum,
                                               stream);
    if (rc != GV_OK) return rc;
    return gv_bn_relu_bwd_apply_grouped(dy, dy_ld, y, y_ld, z, z_ld, mean, inv, gamma, counts, nb, hw, c, num_groups,
                                        accum, dz, dz_ld, dbeta, dgamma, stream);
}

__global__ void scale_inplace_f32(float* __restrict__ x, int64_t n, float s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s;
}

extern "C" int gv_scale(float* x, int64_t n, float s, void* stream) {
    if (!x || n <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(scale_inplace_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, s);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_accumulate(const float* src, int32_t src_ld, float* dst, int32_t dst_ld, int64_t npix,
                             int32_t c, void* stream) {
    if (!src || !dst || npix <= 0 || c <= 0 || src_ld < c || dst_ld < c) return GV_E_BADARG;
    hipLaunchKernelGGL(accumulate_f32, dim3(grid_for(npix * c)), dim3(256), 0, (hipStream_t)stream, src, src_ld,
                       dst, dst_ld, npix, c);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_bias_grad(const float* dz, int32_t dz_ld, int64_t npix, int32_t c, double* accum,
                            float* dbias, void* stream) {
    if (!dz || !accum || !dbias || npix <= 0 || c <= 0 || dz_ld < c || npix > 0x7fffffff) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    GV_HIP_CHECK(hipMemsetAsync(accum, 0, sizeof(double) * 2 * (size_t)c, st));
    int splits = (int)((npix + 2047) / 2048);
    if (splits > 1024) splits = 1024;
    launch_grouped_sums<2>(dim3((c + 63) / 64, splits, 1), st, (c & 3) == 0 && vec_ok(dz, dz_ld), nullptr, 0, dz, dz_ld,
                           nullptr, 0, nullptr, nullptr, (int)npix, 1, c, 1, accum);
    hipLaunchKernelGGL(bn_param_grads, dim3((c + 255) / 256), dim3(256), 0, st, accum, 1, c, dbias, (float*)nullptr);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

static int g_pool_scatter = 0;
extern "C" void gv_pool2d_bwd_set_scatter(int on) { g_pool_scatter = on; }

extern "C" int gv_pool2d_bwd(const gv_pool_desc* d, const void* x, const void* dy, int32_t dy_ld, void* dx,
                             int32_t dx_ld, void* stream) {
    if (!d || !dy || !dx) return GV_E_BADARG;
    const int mode = d->mode & ~GV_POOL_BWD_STORE;
    if ((mode == GV_POOL_MAX && !x) || (mode != GV_POOL_MAX && mode != GV_POOL_AVG)) return GV_E_BADARG;
    // one deterministic gather kernel family for all storage types (train_lp.hip); the atomic-scatter fp32 kernel
    // (pool2d_bwd_f32) is kept behind the tuning hook gv_pool2d_bwd_set_scatter for comparison
    if (d->dtype == GV_BF16 || d->dtype == GV_F16 || (d->dtype == GV_F32 && !g_pool_scatter))
        return gvlp::pool2d_bwd(d, x, dy, dy_ld, dx, dx_ld, (hipStream_t)stream);
    if (d->dtype != GV_F32) return GV_E_UNSUPPORTED;
    if (d->mode & GV_POOL_BWD_STORE) return GV_E_UNSUPPORTED;    // the scatter kernel only adds
    hipLaunchKernelGGL(pool2d_bwd_f32, dim3(grid_for((int64_t)d->nb * d->oh * d->ow * d->c)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)x, d->x_ld, (const float*)dy, dy_ld, d->nb, d->ih, d->iw, d->c,
                       d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, mode, (float*)dx, dx_ld);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_pool2d_fwd_argmax(const gv_pool_desc* d, const void* x, void* y, uint8_t* argmax, void* stream) {
    if (!d || !x || !y || !argmax) return GV_E_BADARG;
    if (d->nb <= 0 || d->ih <= 0 || d->iw <= 0 || d->c <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->oh <= 0 ||
        d->ow <= 0 || d->pad_t < 0 || d->pad_l < 0 || d->x_ld < d->c || d->y_ld < d->c || d->mode != GV_POOL_MAX)
        return GV_E_BADARG;
    if (!gv_pool_geometry_ok(d)) return GV_E_BADARG;
    if (d->kh * d->kw > 255) return GV_E_UNSUPPORTED;               // the tap index is one byte
    if (d->dtype != GV_F32 && d->dtype != GV_BF16 && d->dtype != GV_F16) return GV_E_UNSUPPORTED;
    return gvlp::pool2d_fwd_argmax(d, x, y, argmax, (hipStream_t)stream);
}

extern "C" int gv_bn_bwd_coeffs_t(const double* accum, const int32_t* counts, const float* mean, const float* inv,
                                  const float* gamma, int32_t c, int32_t num_groups, int32_t raw_z, float* coef_a,
                                  float* coef_b, float* coef_c, float* dbeta, float* dgamma, void* stream) {
    if (!accum || !counts || !mean || !inv || !coef_a || !coef_b || !coef_c || c <= 0 || num_groups <= 0) return GV_E_BADARG;
    return gvlp::bn_bwd_coeffs_launch(accum, counts, mean, inv, gamma, c, num_groups, raw_z ? 1 : 0, coef_a, coef_b, coef_c,
                                      dbeta, dgamma, (hipStream_t)stream);
}

extern "C" int gv_pool2d_bwd_argmax_bn(const gv_pool_desc* d, const uint8_t* argmax, const void* dy, int32_t dy_ld,
                                       const void* z, int32_t z_ld, int32_t num_groups, const float* coef_a,
                                       const float* coef_b, const float* coef_c, const float* scale, const float* shift,
                                       void* dz, int32_t dz_ld, void* stream) {
    if (!d || !argmax || !dy || !z || !dz || !coef_a || !coef_b || !coef_c || num_groups <= 0) return GV_E_BADARG;
    if ((scale == nullptr) != (shift == nullptr)) return GV_E_BADARG;
    if (d->nb <= 0 || d->ih <= 0 || d->iw <= 0 || d->c <= 0 || d->oh <= 0 || d->ow <= 0 || dy_ld < d->c || dz_ld < d->c ||
        z_ld < d->c || (d->mode & ~GV_POOL_BWD_STORE) != GV_POOL_MAX)
        return GV_E_BADARG;
    if (!gv_pool_geometry_ok(d)) return GV_E_BADARG;
    if (d->dtype != GV_BF16 && d->dtype != GV_F16) return GV_E_UNSUPPORTED;
    if (d->c % 8 != 0 || z_ld % 8 != 0 || !gv_aligned16(z)) return GV_E_UNSUPPORTED;
    return gvlp::pool2d_bwd_argmax(d, argmax, dy, dy_ld, dz, dz_ld, (hipStream_t)stream, z, z_ld, num_groups, coef_a, coef_b,
                                   coef_c, scale, shift);
}

extern "C" int gv_pool2d_bwd_argmax(const gv_pool_desc* d, const uint8_t* argmax, const void* dy, int32_t dy_ld, void* dx,
                                    int32_t dx_ld, void* stream) {
    if (!d || !argmax || !dy || !dx) return GV_E_BADARG;
    if (d->nb <= 0 || d->ih <= 0 || d->iw <= 0 || d->c <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->oh <= 0 ||
        d->ow <= 0 || d->pad_t < 0 || d->pad_l < 0 || dy_ld < d->c || dx_ld < d->c ||
        (d->mode & ~GV_POOL_BWD_STORE) != GV_POOL_MAX)
        return GV_E_BADARG;
    if (!gv_pool_geometry_ok(d)) return GV_E_BADARG;
    if (d->dtype != GV_F32 && d->dtype != GV_BF16 && d->dtype != GV_F16) return GV_E_UNSUPPORTED;
    return gvlp::pool2d_bwd_argmax(d, argmax, dy, dy_ld, dx, dx_ld, (hipStream_t)stream);
}

static int pool_fuse_bwd_launch(const float* F, const float* dS, int32_t num_views, int32_t num_shapes, int64_t E,
                                int64_t view_stride, int64_t shape_stride, const int32_t* scheme, int32_t num_groups,
                                const float* weight, int32_t mode, float* dF, void* stream, int64_t scheme_stride,
                                int64_t weight_stride) {
    if (!F || !dS || !scheme || !weight || !dF) return GV_E_BADARG;
    if (num_views <= 0 || num_shapes <= 0 || E <= 0 || num_groups <= 0) return GV_E_BADARG;
    if (num_views > 64 || num_groups > 64 || num_shapes > 65535) return GV_E_UNSUPPORTED;
    int64_t bx = (E + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(view_pool_fuse_bwd_f32, dim3((unsigned)bx, (unsigned)num_shapes), dim3(256), 0,
                       (hipStream_t)stream, F, dS, num_views, num_shapes, E, view_stride, shape_stride, scheme,
                       num_groups, weight, mode, dF, scheme_stride, weight_stride);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_view_pool_fuse_bwd(const float* F, const float* dS, int32_t num_views, int32_t num_shapes,
                                     int64_t E, int64_t view_stride, int64_t shape_stride,
                                     const int32_t* scheme, int32_t num_groups, const float* weight,
                                     int32_t mode, float* dF, void* stream) {
    return pool_fuse_bwd_launch(F, dS, num_views, num_shapes, E, view_stride, shape_stride, scheme, num_groups, weight,
                                mode, dF, stream, 0, 0);
}

extern "C" int gv_view_pool_fuse_bwd_per_shape(const float* F, const float* dS, int32_t num_views,
                                               int32_t num_shapes, int64_t E, int64_t view_stride,
                                               int64_t shape_stride, const int32_t* scheme, int32_t num_groups,
                                               const float* weight, int32_t mode, float* dF, void* stream) {
    return pool_fuse_bwd_launch(F, dS, num_views, num_shapes, E, view_stride, shape_stride, scheme, num_groups, weight,
                                mode, dF, stream, (int64_t)num_groups * num_views, num_groups);
}

extern "C" int gv_global_avg_pool_bwd(const float* dgap, int32_t nb, int32_t hw, int32_t c, float* dx,
                                      int32_t dx_ld, void* stream) {
    if (!dgap || !dx || nb <= 0 || hw <= 0 || c <= 0 || dx_ld < c) return GV_E_BADARG;
    hipLaunchKernelGGL(global_avg_pool_bwd_f32, dim3(grid_for((int64_t)nb * hw * c)), dim3(256), 0,
                       (hipStream_t)stream, dgap, nb, hw, c, dx, dx_ld);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_softmax_ce(const float* logits, const int64_t* labels, int32_t n, int32_t c, float* loss,
                             float* dlogits, void* stream) {
    if (!logits || !labels || !loss || !dlogits || n <= 0 || c <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(softmax_ce_f32, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, n, c, loss,
                       dlogits);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_dense_bwd(const float* x, const float* dy, const float* kernel, int32_t n, int32_t f,
                            int32_t c, float* dx, float* dkernel, float* dbias, void* stream) {
    if (!x || !dy || !kernel || !dx || !dkernel || !dbias || n <= 0 || f <= 0 || c <= 0) return GV_E_BADARG;
    int64_t total = (int64_t)n * f;
    if ((int64_t)f * c > total) total = (int64_t)f * c;
    hipLaunchKernelGGL(dense_bwd_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       dy, kernel, n, f, c, dx, dkernel, dbias);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_sgd_momentum(float* w, const float* g, float* m, int64_t n, float lr, float mu, float wd,
                               void* stream) {
    if (!w || !g || !m || n <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(sgd_momentum_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, g,
                       m, n, lr, mu, wd);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_bn_update_moving(const float* mean, const float* var, const int32_t* counts, int32_t num_groups,
                                   int32_t c, float decay, float* moving_mean, float* moving_var, void* stream) {
    if (!mean || !var || !counts || !moving_mean || !moving_var || num_groups <= 0 || c <= 0) return GV_E_BADARG;
    hipLaunchKernelGGL(bn_update_moving, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, var,
                       counts, num_groups, c, decay, moving_mean, moving_var);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

static int g_wgrad_v1 = 0;
extern "C" void gv_conv2d_wgrad_set_v1(int on) { g_wgrad_v1 = on; }

// few-channel stem layers: all taps in one wave, operands straight from global memory
static bool wgrad_direct_ok(const gv_conv_desc* d) {
    const int64_t M = (int64_t)d->nb * d->oh * d->ow;
    return d->kh * d->kw * d->cin <= 288 && d->cin <= 32 && M >= 200000 &&
           (int64_t)d->nb * d->ih * d->iw * d->x_ld < 0x7fffffffll;
}

// the fp32-MFMA filter gradient for storage type S (float; 16-bit: stems and shapes the 16-bit MFMA kernel
// does not take)
template <typename S>
static int wgrad_f32mfma(const gv_conv_desc* d, const S* x, const S* dz, int32_t dz_ld, const GvDw& dw, hipStream_t st) {
    const int64_t M = (int64_t)d->nb * d->oh * d->ow;
    const int R = d->kh * d->kw * d->cin;
    const size_t elems = (size_t)R * d->cout;
    if (wgrad_direct_ok(d)) {
        const int nrt = (R + 31) / 32;
        const int64_t waves = gv_dw_clamp(dw, elems, 256 * 4 * (nrt == 1 ? 6 : (nrt <= 5 ? 2 : 1)));   // (a wave is a slice)
        int64_t per = (M + waves - 1) / waves;
        per = (per + 1) / 2 * 2;
        const int64_t nw = (M + per - 1) / per;
        const GvDw sink = gv_dw_sink(dw, elems, nw);
        const dim3 grid((unsigned)(((nw + 3) / 4) * ((d->cout + 31) / 32)));
#define GV_WGRAD_D(NRT, U)                                                                                          \
        hipLaunchKernelGGL((conv_wgrad_direct_f32<S, NRT, U>), grid, dim3(256), 0, st, x, d->x_ld, dz, dz_ld, d->ih,  \
                           d->iw, d->cin, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, d->cout, M, per,  \
                           sink)
        if (nrt == 1) GV_WGRAD_D(1, 8);
        else if (nrt <= 5) GV_WGRAD_D(5, 4);
        else GV_WGRAD_D(9, 8);
#undef GV_WGRAD_D
        GV_LAUNCH_CHECK();
        return gv_dw_finish(dw, elems, nw, st);
    }
    // 128 channels on a side only where that wastes no more rows than 64-wide tiles would
    const int ti = (d->cin + 127) / 128 * 128 == (d->cin + 63) / 64 * 64 ? 2 : 1;
    const int to = (d->cout + 127) / 128 * 128 == (d->cout + 63) / 64 * 64 ? 2 : 1;
    const int tiles = d->kh * d->kw * ((d->cin + 64 * ti - 1) / (64 * ti)) * ((d->cout + 64 * to - 1) / (64 * to));
    int64_t splits = (2048 + tiles - 1) / tiles;            // ~2k workgroups: 256 CUs x 2-4 resident, 2+ rounds
    const int64_t max_splits = (M + 511) / 512;             // at least 512 pixels per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    splits = gv_dw_clamp(dw, elems, splits);
    int64_t per = (M + splits - 1) / splits;
    per = (per + 31) / 32 * 32;
    splits = (M + per - 1) / per;
    const dim3 grid((unsigned)(tiles * splits));
    const GvDw sink = gv_dw_sink(dw, elems, splits);
#define GV_WGRAD2(TI, TO)                                                                                          \
    hipLaunchKernelGGL((conv_wgrad2_f32<S, TI, TO>), grid, dim3(256), 0, st, x, d->x_ld, dz, dz_ld, d->nb, d->ih,    \
                       d->iw, d->cin, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, d->cout, M, per,     \
                       sink)
    if (ti == 2 && to == 2) GV_WGRAD2(2, 2);
    else if (ti == 2) GV_WGRAD2(2, 1);
    else if (to == 2) GV_WGRAD2(1, 2);
    else GV_WGRAD2(1, 1);
#undef GV_WGRAD2
    GV_LAUNCH_CHECK();
    return gv_dw_finish(dw, elems, splits, st);
}

/* tuning hook: number of launch configurations gv_conv2d_wgrad accepts in gv_conv_desc.tile_cfg for `dtype` */
extern "C" int gv_conv2d_wgrad_num_cfgs(int dtype) { return dtype == GV_BF16 || dtype == GV_F16 ? 30 + gvlp::wgrad_dma_num_cfgs() + 5 : 0; }

static int g_wgrad_lp_f32 = 0;
/* tuning hook: 16-bit storage filter gradients on the fp32 MFMA (typed loads) instead of the 16-bit MFMA kernel */
extern "C" void gv_conv2d_wgrad_set_lp_f32(int on) { g_wgrad_lp_f32 = on; }

// ---- the slices of a filter gradient, added in slice order (the deterministic form: gv_common.h, GvDw) ----------------
// dw[i] += part[0][i] + part[1][i] + ... : a workgroup owns 64 (float4) or 256 (scalar) consecutive elements and splits the
// slices over its four waves (wave w: slices w, w + 4, ...), whose sums are added in wave order — a fixed tree, whatever
// the dispatch order.  Reads are 1 KiB per wave-instruction, 4 in flight per lane.
template <typename V>
__global__ __launch_bounds__(256) void dw_reduce_slices(const V* __restrict__ part, size_t stride, int splits,
                                                        V* __restrict__ dw, size_t n) {
    const int cl = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + cl;
    V s0 = {}, s1 = {};
    if (i < n) {
        int k = sg;
        for (; k + 4 < splits; k += 8) {
            const V a = part[(size_t)k * stride + i], b = part[(size_t)(k + 4) * stride + i];
            s0 += a;
            s1 += b;
        }
        if (k < splits) s0 += part[(size_t)k * stride + i];
    }
    __shared__ V red[4][64];
    red[sg][cl] = s0 + s1;
    __syncthreads();
    if (sg == 0 && i < n) dw[i] += (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

int gv_dw_finish(const GvDw& o, size_t elems, int64_t splits, hipStream_t st) {
    if (!o.part || splits <= 1) return GV_OK;
    const size_t stride = (elems + 3) / 4 * 4;
    if (splits > 0x7fffffff) return GV_E_UNSUPPORTED;
    if (elems % 4 == 0 && gv_aligned16(o.dw) && gv_aligned16(o.part)) {
        const size_t n = elems / 4;
        hipLaunchKernelGGL(dw_reduce_slices<f32x4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st,
                           (const f32x4*)o.part, stride / 4, (int)splits, (f32x4*)o.dw, n);
    } else {
        hipLaunchKernelGGL(dw_reduce_slices<float>, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, st,
                           (const float*)o.part, stride, (int)splits, o.dw, elems);
    }
    GV_LAUNCH_CHECK();
    return GV_OK;
}

static int wgrad_dispatch(const gv_conv_desc* d, const void* x, const void* dz, int32_t dz_ld, const GvDw& dw,
                          hipStream_t st) {
    if (!d || !x || !dz || !dw.dw) return GV_E_BADARG;
    if (d->nb <= 0 || d->cin <= 0 || d->cout <= 0 || dz_ld < d->cout || d->x_ld < d->cin) return GV_E_BADARG;
    if (d->dtype == GV_BF16 || d->dtype == GV_F16) {
        // (the 16-bit MFMA kernel also takes the 32-channel stem layers: the direct kernel's scalar 16-bit loads
        // lose the prefetch, 7-13 ms against ~1 ms)
        if (!g_wgrad_lp_f32 && gvlp::wgrad_mfma_ok(d, x, dz, dz_ld))
            return gvlp::conv_wgrad(d, x, dz, dz_ld, dw, st);
        if (!g_wgrad_lp_f32 && d->cin <= 4) {             // the 3-channel stems: row strips on the 16-bit MFMA
            const int rc = gvlp::conv_wgrad_stem(d, x, dz, dz_ld, dw, st);
            if (rc != GV_E_UNSUPPORTED) return rc;
        }
        if (d->dtype == GV_BF16) return wgrad_f32mfma<__bf16>(d, (const __bf16*)x, (const __bf16*)dz, dz_ld, dw, st);
        return wgrad_f32mfma<_Float16>(d, (const _Float16*)x, (const _Float16*)dz, dz_ld, dw, st);
    }
    if (d->dtype != GV_F32) return GV_E_UNSUPPORTED;
    if (!g_wgrad_v1) return wgrad_f32mfma<float>(d, (const float*)x, (const float*)dz, dz_ld, dw, st);
    const int64_t M = (int64_t)d->nb * d->oh * d->ow;
    const int tiles = d->kh * d->kw * ((d->cin + 63) / 64) * ((d->cout + 63) / 64);
    const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    int64_t splits = (4096 + tiles - 1) / tiles;                // ~4k workgroups in flight
    const int64_t max_splits = (M + 255) / 256;                 // at least 256 pixels per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    splits = gv_dw_clamp(dw, elems, splits);
    int64_t per = (M + splits - 1) / splits;
    per = (per + 31) / 32 * 32;
    splits = (M + per - 1) / per;
    hipLaunchKernelGGL(conv_wgrad_f32, dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, st, (const float*)x,
                       d->x_ld, (const float*)dz, dz_ld, d->nb, d->ih, d->iw, d->cin, d->kh, d->kw, d->stride, d->pad_t,
                       d->pad_l, d->oh, d->ow, d->cout, M, per, gv_dw_sink(dw, elems, splits));
    GV_LAUNCH_CHECK();
    return gv_dw_finish(dw, elems, splits, st);
}

extern "C" int gv_conv2d_wgrad(const gv_conv_desc* d, const void* x, const void* dz, int32_t dz_ld,
                               float* dw_hwio, void* stream) {
    return wgrad_dispatch(d, x, dz, dz_ld, gv_dw_plain(dw_hwio), (hipStream_t)stream);
}

/* The deterministic form: pixel slices store their partial tiles into `workspace` and are added in slice order. */
extern "C" int gv_conv2d_wgrad_ws(const gv_conv_desc* d, const void* x, const void* dz, int32_t dz_ld,
                                  float* dw_hwio, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < 0 || !gv_aligned16(workspace)) return GV_E_BADARG;
    return wgrad_dispatch(d, x, dz, dz_ld, GvDw{dw_hwio, (float*)workspace, 0, (size_t)workspace_bytes},
                          (hipStream_t)stream);
}

// ---- storage-typed forms (16-bit training step; GV_F32 forwards to the fp32 entry points) ---------------------------
static inline bool lp_type(int dtype) { return dtype == GV_BF16 || dtype == GV_F16; }

// pixel splits of a sums launch: 2048 pixels per workgroup on the big layers, but at least ~1024 workgroups in total
// (down to 128 pixels per workgroup) on the small, latency-bound ones
static inline int sums_splits(int64_t npix, int cap, int c = 64, int G = 1) {
    int64_t splits = (npix + 2047) / 2048;
    const int64_t want = 1024 / ((int64_t)((c + 63) / 64) * G) + 1, most = (npix + 127) / 128;
    if (splits < want) splits = want < most ? want : most;
    return (int)(splits > cap ? cap : (splits < 1 ? 1 : splits));
}

extern "C" int gv_bn_sums_grouped_t(const void* z, int32_t nb, int32_t hw, int32_t c, int32_t z_ld,
                                    int32_t num_groups, double* accum, int32_t dtype, void* stream) {
    const bool zeroed = (dtype & GV_ACCUM_ZEROED) != 0;          // the caller keeps a pre-zeroed accumulator per layer
    dtype &= ~GV_ACCUM_ZEROED;
    if (dtype == GV_F32) return gv_bn_sums_grouped((const float*)z, nb, hw, c, z_ld, num_groups, accum, stream);
    if (!lp_type(dtype)) return GV_E_UNSUPPORTED;
    if (!z || !accum) return GV_E_BADARG;
    if (nb <= 0 || hw <= 0 || c <= 0 || z_ld < c || num_groups <= 0 || nb % num_groups != 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (!zeroed) GV_HIP_CHECK(hipMemsetAsync(accum, 0, sizeof(double) * 2 * (size_t)num_groups * c, st));
    return gvlp::grouped_sums(dtype, 0, z, z_ld, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nb, hw, c,
                              num_groups, sums_splits((int64_t)(nb / num_groups) * hw, 256, c, num_groups), accum, st);
}

extern "C" int gv_scale_shift_act_grouped_t(const void* x, int32_t nb, int32_t hw, int32_t c, int32_t x_ld,
                                            const float* scale, const float* shift, int32_t num_groups,
                                            int32_t relu, void* y, int32_t y_ld, int32_t dtype, void* stream) {
    if (dtype == GV_F32)
        return gv_scale_shift_act_grouped((const float*)x, nb, hw, c, x_ld, scale, shift, num_groups, relu, (float*)y,
                                          y_ld, stream);
    if (!lp_type(dtype)) return GV_E_UNSUPPORTED;
    if (!x || !y || !scale || !shift || nb <= 0 || hw <= 0 || c <= 0 || x_ld < c || y_ld < c || num_groups <= 0)
        return GV_E_BADARG;
    return gvlp::scale_shift_act_grouped(dtype, x, nb, hw, c, x_ld, scale, shift, num_groups, relu, y, y_ld,
                                         (hipStream_t)stream);
}

extern "C" int gv_bn_finalize_apply_grouped_t(const double* accum, const int32_t* counts, const float* gamma,
                                              const float* beta, float eps, const void* x, int32_t nb, int32_t hw,
                                              int32_t c, int32_t x_ld, int32_t num_groups, int32_t relu, void* y,
                                              int32_t y_ld, float* mean, float* var, float* inv, float* scale,
                                              float* shift, int32_t dtype, void* stream) {
    if (!accum || !counts || !beta || !x || !y || !mean || !var || !inv || !scale || !shift) return GV_E_BADARG;
    if (nb <= 0 || hw <= 0 || c <= 0 || x_ld < c || y_ld < c || num_groups <= 0) return GV_E_BADARG;
    if (lp_type(dtype) || dtype == GV_F32) {
        const int rc = gvlp::bn_finalize_apply_grouped(dtype, accum, counts, gamma, beta, eps, x, nb, hw, c, x_ld,
                                                       num_groups, relu, y, y_ld, mean, var, inv, scale, shift,
                                                       (hipStream_t)stream);
        if (rc != GV_E_UNSUPPORTED) return rc;
    }
    const int rc = gv_bn_finalize_grouped(accum, c, num_groups, counts, gamma, beta, eps, mean, var, inv, scale, shift,
                                          stream);
    if (rc != GV_OK) return rc;
    return gv_scale_shift_act_grouped_t(x, nb, hw, c, x_ld, scale, shift, num_groups, relu, y, y_ld, dtype, stream);
}

extern "C" int gv_bn_relu_bwd_sums_grouped_t(const void* dy, int32_t dy_ld, const void* y, int32_t y_ld,
                                             const void* z, int32_t z_ld, const float* mean, const float* inv,
                                             int32_t nb, int32_t hw, int32_t c, int32_t num_groups, double* accum,
                                             const float* scale, const float* shift, int32_t dtype, void* stream) {
    const bool zeroed = (dtype & GV_ACCUM_ZEROED) != 0;
    dtype &= ~GV_ACCUM_ZEROED;
    if (dtype == GV_F32) {
        if ((scale == nullptr) != (shift == nullptr)) return GV_E_BADARG;
        return bn_bwd_sums_f32((const float*)dy, dy_ld, (const float*)y, y_ld, (const float*)z, z_ld, mean, inv, nb, hw, c,
                               num_groups, accum, scale, shift, stream);
    }
    if ((scale == nullptr) != (shift == nullptr)) return GV_E_BADARG;
    if (!lp_type(dtype)) return GV_E_UNSUPPORTED;
    if (!dy || !z || !mean || !inv || !accum) return GV_E_BADARG;
    if (nb <= 0 || hw <= 0 || c <= 0 || num_groups <= 0 || nb % num_groups != 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (!zeroed) GV_HIP_CHECK(hipMemsetAsync(accum, 0, sizeof(double) * 2 * (size_t)num_groups * c, st));
    return gvlp::grouped_sums(dtype, 1, z, z_ld, dy, dy_ld, y, y_ld, mean, inv, scale, shift, nb, hw, c, num_groups,
                              sums_splits((int64_t)(nb / num_groups) * hw, 256, c, num_groups), accum, st);
}

extern "C" int gv_bn_relu_bwd_apply_grouped_t(const void* dy, int32_t dy_ld, const void* y, int32_t y_ld,
                                              const void* z, int32_t z_ld, const float* mean, const float* inv,
                                              const float* gamma, const int32_t* counts, int32_t nb, int32_t hw,
                                              int32_t c, int32_t num_groups, const double* accum, void* dz,
                                              int32_t dz_ld, float* dbeta, float* dgamma, const float* scale,
                                              const float* shift, int32_t accumulate, int32_t dtype, void* stream) {
    const int raw_z = (dtype & GV_ACCUM_RAW_Z) ? 1 : 0;
    dtype &= ~GV_ACCUM_RAW_Z;
    if (dtype == GV_F32) {
        if ((scale == nullptr) != (shift == nullptr)) return GV_E_BADARG;
        if (raw_z) return GV_E_UNSUPPORTED;
        return bn_bwd_apply_f32((const float*)dy, dy_ld, (const float*)y, y_ld, (const float*)z, z_ld, mean, inv, gamma,
                                counts, nb, hw, c, num_groups, accum, (float*)dz, dz_ld, dbeta, dgamma, scale, shift,
                                accumulate, stream);
    }
    if ((scale == nullptr) != (shift == nullptr)) return GV_E_BADARG;
    if (!lp_type(dtype)) return GV_E_UNSUPPORTED;
    if (!dy || !z || !mean || !inv || !counts || !accum || !dz) return GV_E_BADARG;
    if (nb <= 0 || hw <= 0 || c <= 0 || num_groups <= 0 || nb % num_groups != 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    bool done = false;                                           // (the streaming kernel also sums dbeta / dgamma)
    const int rc = gvlp::bn_bwd_apply_grouped(dtype, dy, dy_ld, y, y_ld, z, z_ld, mean, inv, gamma, accum, counts, scale,
                                              shift, accumulate, nb, hw, c, num_groups, dz, dz_ld, dbeta, dgamma, &done,
                                              st, raw_z);
    if (rc != GV_OK) return rc;
    if ((dbeta || dgamma) && !done)
        hipLaunchKernelGGL(bn_param_grads, dim3((c + 255) / 256), dim3(256), 0, st, accum, num_groups, c, dbeta,
                           dgamma);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_accumulate_t(const void* src, int32_t src_ld, void* dst, int32_t dst_ld, int64_t npix, int32_t c,
                               int32_t dtype, void* stream) {
    if (dtype == GV_F32) return gv_accumulate((const float*)src, src_ld, (float*)dst, dst_ld, npix, c, stream);
    if (!lp_type(dtype)) return GV_E_UNSUPPORTED;
    if (!src || !dst || npix <= 0 || c <= 0 || src_ld < c || dst_ld < c) return GV_E_BADARG;
    return gvlp::accumulate(dtype, src, src_ld, dst, dst_ld, npix, c, (hipStream_t)stream);
}

extern "C" int gv_bias_grad_t(const void* dz, int32_t dz_ld, int64_t npix, int32_t c, double* accum, float* dbias,
                              int32_t dtype, void* stream) {
    if (dtype == GV_F32) return gv_bias_grad((const float*)dz, dz_ld, npix, c, accum, dbias, stream);
    if (!lp_type(dtype)) return GV_E_UNSUPPORTED;
    if (!dz || !accum || !dbias || npix <= 0 || c <= 0 || dz_ld < c || npix > 0x7fffffff) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    GV_HIP_CHECK(hipMemsetAsync(accum, 0, sizeof(double) * 2 * (size_t)c, st));
    const int rc = gvlp::grouped_sums(dtype, 2, nullptr, 0, dz, dz_ld, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                                      (int)npix, 1, c, 1, sums_splits(npix, 1024), accum, st);
    if (rc != GV_OK) return rc;
    hipLaunchKernelGGL(bn_param_grads, dim3((c + 255) / 256), dim3(256), 0, st, accum, 1, c, dbias, (float*)nullptr);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_view_pool_fuse_bwd_t(const void* F, const float* dS, int32_t num_views, int32_t num_shapes,
                                       int64_t E, int64_t view_stride, int64_t shape_stride, const int32_t* scheme,
                                       int32_t num_groups, const float* weight, int32_t mode, void* dF,
                                       int32_t per_shape, int32_t dtype, void* stream) {
    const int64_t ss = per_shape ? (int64_t)num_groups * num_views : 0, ws = per_shape ? num_groups : 0;
    if (dtype == GV_F32)
        return pool_fuse_bwd_launch((const float*)F, dS, num_views, num_shapes, E, view_stride, shape_stride, scheme,
                                    num_groups, weight, mode, (float*)dF, stream, ss, ws);
    if (!lp_type(dtype)) return GV_E_UNSUPPORTED;
    if (!F || !dS || !scheme || !weight || !dF) return GV_E_BADARG;
    if (num_views <= 0 || num_shapes <= 0 || E <= 0 || num_groups <= 0) return GV_E_BADARG;
    if (num_views > 64 || num_groups > 64 || num_shapes > 65535) return GV_E_UNSUPPORTED;
    return gvlp::view_pool_fuse_bwd(dtype, F, dS, num_views, num_shapes, E, view_stride, shape_stride, scheme,
                                    num_groups, weight, mode, dF, (hipStream_t)stream, ss, ws);
}
