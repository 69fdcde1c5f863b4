// train_lp.hip — the training step on 16-bit storage (configs[2]: bf16 forward + backward).
// Activations and activation gradients live in HBM as GV_BF16 / GV_F16; batch statistics accumulate in fp64,
// parameter gradients (dW, dbeta, dgamma, dbias) and the optimizer state stay fp32, every elementwise result is
// computed in fp32 and rounded to the storage type once.  Same algorithms as train.hip (the fp32 step); a thread
// owns 8 consecutive channels (one 16-byte load) wherever the channel count allows.
//
// The filter gradient runs on v_mfma_f32_32x32x16_{bf16,f16}.  Its reduction axis is the PIXEL axis while both
// operands (the shifted input and dZ) are channel-contiguous in HBM, i.e. k-strided for the MFMA: the tiles are
// written to LDS as they arrive ([pixel][channel], ds_write_b128) and read back with gfx950's transposing
// ds_read_b64_tr_b16, which hands lane (channel i) four consecutive pixels of its column.
#include <math.h>

#include "conv_stats.h"
#include "lowp.h"
#include "lp_elem.h"

namespace {

using namespace gvlp_elem;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

inline unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int64_t cap = 256 * 16;
    return (unsigned)(b < cap ? (b > 0 ? b : 1) : cap);
}

// dst[p][c] += src[p][c]
template <typename T, int VEC>
__global__ __launch_bounds__(256) void accumulate_lp(const unsigned short* __restrict__ src, int src_ld,
                                                     unsigned short* __restrict__ dst, int dst_ld, int64_t npix,
                                                     int c) {
    const int cg = c / VEC;
    const int64_t total = npix * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        const int64_t pix = idx / cg;
        float a[8], b[8];
        load_v<T, VEC>(src + pix * src_ld + q * VEC, a);
        load_v<T, VEC>(dst + pix * dst_ld + q * VEC, b);
#pragma unroll
        for (int e = 0; e < VEC; ++e) b[e] += a[e];
        store_v<T, VEC>(dst + pix * dst_ld + q * VEC, b);
    }
}

// Per-(group, channel) sums, see grouped_sums_f32 in train.hip.  MODE 0: sum z, sum z^2; MODE 1: sum g, sum g*zhat with
// g = dy*[y>0]; MODE 2: sum dz.  A thread owns VEC channels; 64 channels per block; grid (channel blocks, splits, G).
template <typename T, int MODE, int VEC>
__global__ __launch_bounds__(256) void grouped_sums_lp(const unsigned short* __restrict__ z, int z_ld,
                                                       const unsigned short* __restrict__ dy, int dy_ld,
                                                       const unsigned short* __restrict__ y, int y_ld,
                                                       const float* __restrict__ mean, const float* __restrict__ inv,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int nb, int hw, int c, int G,
                                                       double* __restrict__ acc) {
    constexpr int TPC = 64 / VEC;
    constexpr int PL = 256 / TPC;
    const int cl = threadIdx.x % TPC;
    const int pl = threadIdx.x / TPC;
    const int ch = blockIdx.x * 64 + cl * VEC;
    const int g = blockIdx.z;
    const int nimg = (nb - g + G - 1) / G;
    const int64_t npix = (int64_t)nimg * hw;
    const int64_t per = (npix + gridDim.y - 1) / gridDim.y;
    const int64_t p0 = (int64_t)blockIdx.y * per;
    const int64_t p1 = p0 + per < npix ? p0 + per : npix;
    double s0[VEC], s1[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s0[e] = s1[e] = 0.0;
    if (ch < c) {
        float mu[VEC], iv[VEC], sc[VEC], sh[VEC];
        const bool rmask = MODE == 1 && !y && scale;          // ReLU mask recomputed from z: y = z*scale + shift > 0
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            mu[e] = MODE == 1 ? mean[g * c + ch + e] : 0.f;
            iv[e] = MODE == 1 ? inv[g * c + ch + e] : 0.f;
            sc[e] = rmask ? scale[g * c + ch + e] : 0.f;
            sh[e] = rmask ? shift[g * c + ch + e] : 0.f;
        }
        // fp32 partial sums over short runs (exact enough for 16-bit inputs), folded into fp64 every 16 pixels
        for (int64_t pb = p0 + pl; pb < p1; pb += (int64_t)PL * 16) {
            float f0[VEC], f1[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) f0[e] = f1[e] = 0.f;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int64_t p = pb + (int64_t)it * PL;
                if (p >= p1) break;
                const int k = (int)(p / hw);
                const int64_t pix = (int64_t)(k * G + g) * hw + (p - (int64_t)k * hw);
                float zv[8], gv[8], yv[8];
                if (MODE != 2) load_v<T, VEC>(z + pix * z_ld + ch, zv);
                if (MODE != 0) load_v<T, VEC>(dy + pix * dy_ld + ch, gv);
                if (MODE == 1 && y) load_v<T, VEC>(y + pix * y_ld + ch, yv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (MODE == 0) {
                        f0[e] += zv[e];
                        f1[e] = fmaf(zv[e], zv[e], f1[e]);
                    } else if (MODE == 1) {
                        float gr = gv[e];
                        if (y && !(yv[e] > 0.f)) gr = 0.f;
                        if (rmask && !(fmaf(zv[e], sc[e], sh[e]) > 0.f)) gr = 0.f;
                        f0[e] += gr;
                        f1[e] = fmaf(gr, (zv[e] - mu[e]) * iv[e], f1[e]);
                    } else {
                        f0[e] += gv[e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) { s0[e] += f0[e]; s1[e] += f1[e]; }
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
        // the block's partial on the fixed grid of conv_stats.h: the fp64 additions are then exact, so the sums do not
        // depend on the order the blocks arrive in (bitwise reproducible run to run)
        const double q0 = MODE == 0 ? gvconv::STAT_Q_FWD0 : gvconv::STAT_Q_BWD, q1 = MODE == 0 ? gvconv::STAT_Q_FWD1 : gvconv::STAT_Q_BWD;
        atomicAdd(&acc[o], rint(a * q0) / q0);
        if (MODE != 2) atomicAdd(&acc[o + 1], rint(b * q1) / q1);
    }
}

// y = act(x*scale[g][c] + shift[g][c]),  g = image % G
template <typename T, int VEC>
__global__ __launch_bounds__(256) void scale_shift_act_grouped_lp(const unsigned short* __restrict__ x, int nb, int hw,
                                                                  int c, int x_ld, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int G, int relu,
                                                                  unsigned short* __restrict__ y, int y_ld) {
    const int cg = c / VEC;
    const int64_t total = (int64_t)nb * hw * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        const int64_t pix = idx / cg;
        const int g = (int)((pix / hw) % G);
        float v[8];
        load_v<T, VEC>(x + pix * x_ld + q * VEC, v);
        const float* sc = scale + (size_t)g * c + q * VEC;
        const float* sh = shift + (size_t)g * c + q * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[e] = v[e] * sc[e] + sh[e];
            if (relu) v[e] = fmaxf(v[e], 0.f);
        }
        store_v<T, VEC>(y + pix * y_ld + q * VEC, v);
    }
}

// dz += gamma*inv * (g - s1/m - zhat*s2/m),  g = dy*[y>0]
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_grouped_lp(
    const unsigned short* __restrict__ dy, int dy_ld, const unsigned short* __restrict__ y, int y_ld,
    const unsigned short* __restrict__ z, int z_ld, const float* __restrict__ mean, const float* __restrict__ inv,
    const float* __restrict__ gamma, const double* __restrict__ acc, const int* __restrict__ counts,
    const float* __restrict__ scale, const float* __restrict__ shift, int accumulate, int nb, int hw, int c, int G,
    unsigned short* __restrict__ dz, int dz_ld) {
    const int cg = c / VEC;
    const int64_t total = (int64_t)nb * hw * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        const int64_t pix = idx / cg;
        const int g = (int)((pix / hw) % G);
        const int gi = g * c + q * VEC;
        float gr[8], yv[8], zv[8], dv[8];
        load_v<T, VEC>(dy + pix * dy_ld + q * VEC, gr);
        if (y) load_v<T, VEC>(y + pix * y_ld + q * VEC, yv);
        load_v<T, VEC>(z + pix * z_ld + q * VEC, zv);
        if (accumulate) load_v<T, VEC>(dz + pix * dz_ld + q * VEC, dv);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) dv[e] = 0.f;
        const float rm = 1.f / (float)counts[g];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float gg = gr[e];
            if (y && !(yv[e] > 0.f)) gg = 0.f;
            if (!y && scale && !(fmaf(zv[e], scale[gi + e], shift[gi + e]) > 0.f)) gg = 0.f;
            const float iv = inv[gi + e];
            const float zh = (zv[e] - mean[gi + e]) * iv;
            const float s1 = (float)acc[(size_t)(gi + e) * 2], s2 = (float)acc[(size_t)(gi + e) * 2 + 1];
            const float coef = (gamma ? gamma[q * VEC + e] : 1.f) * iv;
            dv[e] += coef * (gg - s1 * rm - zh * s2 * rm);
        }
        store_v<T, VEC>(dz + pix * dz_ld + q * VEC, dv);
    }
}

// ---- the three BatchNorm passes, 8 channels per thread, streaming form ------------------------------------------
// Block = 8 channel threads (64 channels, 128 contiguous bytes per pixel) x 32 pixel lanes; grid (channel blocks,
// pixel splits, G).  A thread keeps its channels for the whole pixel range, so everything per (group, channel) —
// mean, inv, the folded backward coefficients — is computed once, and the walk over the group's pixels (images
// g, g+G, g+2G, ... of the batch) is incremental: no division in the loop, four 16-byte loads in flight per stream.
struct GroupWalk {
    int r, hw, step_img;
    int64_t pix;
    __device__ __forceinline__ void init(int p, int hw_, int G, int g) {
        hw = hw_;
        const int k = p / hw_;
        r = p - k * hw_;
        pix = (int64_t)(k * G + g) * hw_ + r;
        step_img = (G - 1) * hw_;
    }
    __device__ __forceinline__ void advance(int n) {
        pix += n;
        if (step_img) {                                          // (one group: the pixels are contiguous)
            r += n;
            while (r >= hw) { r -= hw; pix += step_img; }
        }
    }
};

template <typename E, typename T, int MODE>
__global__ __launch_bounds__(256) void grouped_sums_v8(const E* __restrict__ z, int z_ld,
                                                       const E* __restrict__ dy, int dy_ld,
                                                       const E* __restrict__ y, int y_ld,
                                                       const float* __restrict__ mean, const float* __restrict__ inv,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int nb, int hw, int c, int G,
                                                       double* __restrict__ acc) {
    constexpr int NE = 16 / sizeof(E), TPC = 64 / NE, PL = 256 / TPC, U = 4;   // elements per quad, channel threads, pixel lanes
    const int cl = threadIdx.x % TPC, pl = threadIdx.x / TPC;
    const int ch = blockIdx.x * 64 + cl * NE;
    const int g = blockIdx.z;
    const int nimg = (nb - g + G - 1) / G;
    const int npix = nimg * hw;                                  // < 2^31 (checked by the launcher)
    const int per = (npix + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per;
    const int p1 = p0 + per < npix ? p0 + per : npix;
    double s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < NE; ++e) s0[e] = s1[e] = 0.0;
    if (ch < c && p0 + pl < p1) {
        float mu[8], iv[8], sc[8], sh[8];
        const bool rmask = MODE == 1 && !y && scale;          // ReLU mask recomputed from z: y = z*scale + shift > 0
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            mu[e] = MODE == 1 ? mean[g * c + ch + e] : 0.f;
            iv[e] = MODE == 1 ? inv[g * c + ch + e] : 0.f;
            sc[e] = rmask ? scale[g * c + ch + e] : 0.f;
            sh[e] = rmask ? shift[g * c + ch + e] : 0.f;
        }
        GroupWalk w;
        w.init(p0 + pl, hw, G, g);
        const int n_it = (p1 - p0 - pl + PL - 1) / PL;
        float f0[8], f1[8];
        auto fold = [&]() {
#pragma unroll
            for (int e = 0; e < NE; ++e) { s0[e] += f0[e]; s1[e] += f1[e]; f0[e] = f1[e] = 0.f; }
        };
        auto add = [&](const u32x4 zq, const u32x4 gq, const u32x4 yq) {
            float zv[8], gv[8], yv[8];
            if (MODE != 2) unpackq<E, T>(zq, zv);
            if (MODE != 0) unpackq<E, T>(gq, gv);
            if (MODE == 1 && y) unpackq<E, T>(yq, yv);
            if constexpr (sizeof(E) == 4) {                      // fp32 storage: straight into fp64 (fp32 runs would cost
#pragma unroll                                                   // 1e-7 of the statistics; 16-bit inputs lose nothing)
                for (int e = 0; e < NE; ++e) {
                    if (MODE == 0) {
                        s0[e] += zv[e];
                        s1[e] += (double)zv[e] * zv[e];
                    } else if (MODE == 1) {
                        float gr = gv[e];
                        if (y && !(yv[e] > 0.f)) gr = 0.f;
                        if (rmask && !(fmaf(zv[e], sc[e], sh[e]) > 0.f)) gr = 0.f;
                        s0[e] += gr;
                        s1[e] += (double)gr * ((zv[e] - mu[e]) * iv[e]);
                    } else {
                        s0[e] += gv[e];
                    }
                }
                return;
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                if (MODE == 0) {
                    f0[e] += zv[e];
                    f1[e] = fmaf(zv[e], zv[e], f1[e]);
                } else if (MODE == 1) {
                    float gr = gv[e];
                    if (y && !(yv[e] > 0.f)) gr = 0.f;
                    if (rmask && !(fmaf(zv[e], sc[e], sh[e]) > 0.f)) gr = 0.f;
                    f0[e] += gr;
                    f1[e] = fmaf(gr, (zv[e] - mu[e]) * iv[e], f1[e]);
                } else {
                    f0[e] += gv[e];
                }
            }
        };
#pragma unroll
        for (int e = 0; e < NE; ++e) f0[e] = f1[e] = 0.f;
        int it = 0;
        for (; it + U <= n_it; it += U) {
            u32x4 zq[U], gq[U], yq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE != 2) zq[u] = *reinterpret_cast<const u32x4*>(z + w.pix * z_ld + ch);
                if (MODE != 0) gq[u] = *reinterpret_cast<const u32x4*>(dy + w.pix * dy_ld + ch);
                if (MODE == 1 && y) yq[u] = *reinterpret_cast<const u32x4*>(y + w.pix * y_ld + ch);
                w.advance(PL);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) add(zq[u], gq[u], yq[u]);
            if ((it & 12) == 12) fold();                         // fp32 runs of 16 values, then fp64
        }
        for (; it < n_it; ++it) {
            u32x4 zq = {0u, 0u, 0u, 0u}, gq = zq, yq = zq;
            if (MODE != 2) zq = *reinterpret_cast<const u32x4*>(z + w.pix * z_ld + ch);
            if (MODE != 0) gq = *reinterpret_cast<const u32x4*>(dy + w.pix * dy_ld + ch);
            if (MODE == 1 && y) yq = *reinterpret_cast<const u32x4*>(y + w.pix * y_ld + ch);
            w.advance(PL);
            add(zq, gq, yq);
        }
        fold();
    }
    __shared__ double red[2][PL][64];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        red[0][pl][cl * NE + e] = s0[e];
        red[1][pl][cl * NE + e] = s1[e];
    }
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < c) {
        double a = 0.0, b = 0.0;
        for (int q = 0; q < PL; ++q) { a += red[0][q][threadIdx.x]; b += red[1][q][threadIdx.x]; }
        const size_t o = ((size_t)g * c + blockIdx.x * 64 + threadIdx.x) * 2;
        // the block's partial on the fixed grid of conv_stats.h: the fp64 additions are then exact, so the sums do not
        // depend on the order the blocks arrive in (bitwise reproducible run to run)
        // (fp32 storage: grids 2^-40 / 2^-36 forward, 2^-52 backward — far below fp32 resolution, and they have to be:
        // with the 16-bit grids, 2^-30 / 2^-24 / 2^-40, a layer whose activations are ~1e-2 gets its variance to a relative
        // 1e-7 only and a randomly initialised ResNet's gradient sums of ~1e-8 to 5e-5, which 50 train-mode BatchNorm layers
        // amplify past the 2e-4 the sharded-engine tests of test_gpu_train.py hold.  The price is the exactness range:
        // totals below 2^13 / 2^17 / 2 add exactly (order-independent, bitwise reproducible); beyond them the additions
        // are ordinary fp64 additions, reproducible to 1e-16 relative.  Round 3's backward grid, 2^-60, was exact only
        // below 2^-7.)
        const double q0 = sizeof(E) == 4 ? (MODE == 0 ? 1099511627776.0 : 4503599627370496.0)
                                         : (MODE == 0 ? gvconv::STAT_Q_FWD0 : gvconv::STAT_Q_BWD);
        const double q1 = sizeof(E) == 4 ? (MODE == 0 ? 68719476736.0 : 4503599627370496.0)
                                         : (MODE == 0 ? gvconv::STAT_Q_FWD1 : gvconv::STAT_Q_BWD);
        atomicAdd(&acc[o], rint(a * q0) / q0);
        if (MODE != 2) atomicAdd(&acc[o + 1], rint(b * q1) / q1);
    }
}

// FWD: y = act(x*scale + shift).  BWD: dz += A*g + B*z + C with g = dy*[y>0] and, per (group, channel),
// A = gamma*inv, B = -A*inv*s2/m, C = A*(mean*inv*s2 - s1)/m   (= gamma*inv*(g - s1/m - zhat*s2/m)).
// Optional work folded into the streaming kernels (saves two tiny launches per BatchNorm):
//   forward : the finalize step — mean/var/inv and the folded scale/shift from the fp64 sums (exactly the arithmetic of
//             bn_finalize_grouped), computed by every thread for its channels, stored by one thread per (group, channel);
//   backward: dbeta / dgamma = the sums over the groups (bn_param_grads), by one thread per channel.
struct BnExtra {
    const double* fin_acc;
    const float* beta;
    float eps;
    float *mean, *var, *inv, *scale_out, *shift_out;
    float *dbeta, *dgamma;
};

template <typename E, typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_stream_v8(const E* __restrict__ x, int x_ld,
                                                    const E* __restrict__ dy, int dy_ld,
                                                    const E* __restrict__ yact, int y_ld,
                                                    const float* __restrict__ p0f, const float* __restrict__ p1f,
                                                    const float* __restrict__ gamma, const double* __restrict__ acc,
                                                    const int* __restrict__ counts, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int accumulate, int nb, int hw,
                                                    int c, int G, int relu, E* __restrict__ out,
                                                    int out_ld, BnExtra ex) {
    constexpr int NE = 16 / sizeof(E), TPC = 64 / NE, PL = 256 / TPC, U = 4;   // elements per quad, channel threads, pixel lanes
    const int cl = threadIdx.x % TPC, pl = threadIdx.x / TPC;
    const int ch = blockIdx.x * 64 + cl * NE;
    const int g = blockIdx.z;
    const int nimg = (nb - g + G - 1) / G;
    const int npix = nimg * hw;
    const int per = (npix + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per;
    const int p1 = p0 + per < npix ? p0 + per : npix;
    // forward with finalize: ONE thread per channel of the block does the fp64 arithmetic (64 instead of 2048 fp64
    // rsqrt per block), the others pick the result up from LDS
    __shared__ float sA[64], sB[64];
    if constexpr (!BWD) {
        if (ex.fin_acc) {
            const int cht = blockIdx.x * 64 + threadIdx.x;
            if (threadIdx.x < 64 && cht < c) {
                const int gi = g * c + cht;
                const double m = (double)counts[g];
                const double mu = ex.fin_acc[(size_t)gi * 2] / m;
                double v = ex.fin_acc[(size_t)gi * 2 + 1] / m - mu * mu;
                if (v < 0.0) v = 0.0;
                const float iv = (float)(1.0 / sqrt(v + (double)ex.eps));
                const float ga = gamma ? gamma[cht] : 1.f;
                const float a_ = iv * ga, b_ = ex.beta[cht] - (float)mu * iv * ga;
                sA[threadIdx.x] = a_;
                sB[threadIdx.x] = b_;
                if (blockIdx.y == 0) {
                    ex.mean[gi] = (float)mu;
                    ex.var[gi] = (float)v;
                    ex.inv[gi] = iv;
                    ex.scale_out[gi] = a_;
                    ex.shift_out[gi] = b_;
                }
            }
            __syncthreads();
        }
    }
    if constexpr (BWD) {                                         // dbeta / dgamma: one thread per channel, once per launch
        const int cht = blockIdx.x * 64 + threadIdx.x;
        if ((ex.dbeta || ex.dgamma) && blockIdx.y == 0 && g == 0 && threadIdx.x < 64 && cht < c) {
            double a = 0.0, b2 = 0.0;
#pragma unroll 4
            for (int gg = 0; gg < G; ++gg) {
                const double a0 = acc[((size_t)gg * c + cht) * 2], a1 = acc[((size_t)gg * c + cht) * 2 + 1];
                a += a0;
                // relu != 0 (backward): accum holds sum g*z, not sum g*zhat (GV_ACCUM_RAW_Z; p0f = mean, p1f = inv)
                b2 += relu ? (double)p1f[gg * c + cht] * (a1 - (double)p0f[gg * c + cht] * a0) : a1;
            }
            if (ex.dbeta) ex.dbeta[cht] += (float)a;
            if (ex.dgamma) ex.dgamma[cht] += (float)b2;
        }
    }
    const bool rmask = BWD && !yact && scale;                    // ReLU mask recomputed from z (x here)
    // backward: the folded coefficients of the block's 64 channels — ONE thread per channel reads mean / inv / the sums /
    // gamma and does the fp64 conversion, everybody else picks A, B, C (and the mask constants) up from LDS: the small
    // late layers are latency-bound, and 256 threads each fetching 7 values for each of their 8 channels before the first
    // activation load was most of their prologue
    __shared__ float sBw[5][64];
    if constexpr (BWD) {
        const int cht = blockIdx.x * 64 + threadIdx.x;
        if (threadIdx.x < 64 && cht < c) {
            const int gi = g * c + cht;
            const float iv = p1f[gi], mu = p0f[gi];              // p0f = mean, p1f = inv
            const float rm = 1.f / (float)counts[g];
            const double a0 = acc[(size_t)gi * 2], a1 = acc[(size_t)gi * 2 + 1];
            const float s1 = (float)a0, s2 = (float)(relu ? (double)iv * (a1 - (double)mu * a0) : a1);   // (relu: GV_ACCUM_RAW_Z)
            const float A_ = (gamma ? gamma[cht] : 1.f) * iv;
            sBw[0][threadIdx.x] = A_;
            sBw[1][threadIdx.x] = -A_ * iv * s2 * rm;
            sBw[2][threadIdx.x] = A_ * (mu * iv * s2 - s1) * rm;
            sBw[3][threadIdx.x] = rmask ? scale[gi] : 0.f;
            sBw[4][threadIdx.x] = rmask ? shift[gi] : 0.f;
        }
        __syncthreads();
    }
    if (ch >= c || p0 + pl >= p1) return;
    float A[8], B[8], Cc[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int gi = g * c + ch + e;
        if constexpr (BWD) {
            A[e] = sBw[0][cl * NE + e];
            B[e] = sBw[1][cl * NE + e];
            Cc[e] = sBw[2][cl * NE + e];
            sc[e] = sBw[3][cl * NE + e];
            sh[e] = sBw[4][cl * NE + e];
        } else if (ex.fin_acc) {                                 // finalized above
            sc[e] = sh[e] = 0.f;
            A[e] = sA[cl * NE + e];
            B[e] = sB[cl * NE + e];
        } else {                                                 // p0f = scale, p1f = shift
            sc[e] = sh[e] = 0.f;
            A[e] = p0f[gi];
            B[e] = p1f[gi];
        }
    }
    GroupWalk w;
    w.init(p0 + pl, hw, G, g);
    const int n_it = (p1 - p0 - pl + PL - 1) / PL;
    auto one = [&](int64_t pix, const u32x4 xq, const u32x4 gq, const u32x4 yq, const u32x4 oq) {
        float xv[8], o[8];
        unpackq<E, T>(xq, xv);
        if constexpr (BWD) {
            float gv[8], yv[8];
            unpackq<E, T>(gq, gv);
            if (accumulate) unpackq<E, T>(oq, o);
            else
#pragma unroll
                for (int e = 0; e < NE; ++e) o[e] = 0.f;
            if (yact) unpackq<E, T>(yq, yv);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                float gg = gv[e];
                if (yact && !(yv[e] > 0.f)) gg = 0.f;
                if (rmask && !(fmaf(xv[e], sc[e], sh[e]) > 0.f)) gg = 0.f;
                o[e] += fmaf(A[e], gg, fmaf(B[e], xv[e], Cc[e]));
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                o[e] = fmaf(xv[e], A[e], B[e]);
                if (relu) o[e] = fmaxf(o[e], 0.f);
            }
        }
        *reinterpret_cast<u32x4*>(out + pix * out_ld + ch) = packq<E, T>(o);
    };
    int it = 0;
    for (; it + U <= n_it; it += U) {
        u32x4 xq[U], gq[U], yq[U], oq[U];
        int64_t px[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            px[u] = w.pix;
            xq[u] = *reinterpret_cast<const u32x4*>(x + w.pix * x_ld + ch);
            if constexpr (BWD) {
                gq[u] = *reinterpret_cast<const u32x4*>(dy + w.pix * dy_ld + ch);
                if (yact) yq[u] = *reinterpret_cast<const u32x4*>(yact + w.pix * y_ld + ch);
                if (accumulate) oq[u] = *reinterpret_cast<const u32x4*>(out + w.pix * out_ld + ch);
            }
            w.advance(PL);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(px[u], xq[u], gq[u], yq[u], oq[u]);
    }
    for (; it < n_it; ++it) {
        u32x4 xq, gq = {0u, 0u, 0u, 0u}, yq = gq, oq = gq;
        xq = *reinterpret_cast<const u32x4*>(x + w.pix * x_ld + ch);
        if constexpr (BWD) {
            gq = *reinterpret_cast<const u32x4*>(dy + w.pix * dy_ld + ch);
            if (yact) yq = *reinterpret_cast<const u32x4*>(yact + w.pix * y_ld + ch);
            if (accumulate) oq = *reinterpret_cast<const u32x4*>(out + w.pix * out_ld + ch);
        }
        one(w.pix, xq, gq, yq, oq);
        w.advance(PL);
    }
}

// A, B, C of dz = A*g + B*z + C per (group, channel) from the backward sums (the arithmetic of bn_stream_v8<BWD>), and
// dbeta / dgamma += the sums over the groups: what a kernel other than bn_stream_v8 needs to finish a BatchNorm backward.
__global__ __launch_bounds__(256) void bn_bwd_coeffs(const double* __restrict__ acc, const int* __restrict__ counts,
                                                     const float* __restrict__ mean, const float* __restrict__ inv,
                                                     const float* __restrict__ gamma, int c, int G, int raw_z,
                                                     float* __restrict__ A, float* __restrict__ B, float* __restrict__ Cc,
                                                     float* __restrict__ dbeta, float* __restrict__ dgamma) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    double sa = 0.0, sb = 0.0;
    for (int g = 0; g < G; ++g) {
        const int gi = g * c + ch;
        const float iv = inv[gi], mu = mean[gi];
        const float rm = 1.f / (float)counts[g];
        const double a0 = acc[(size_t)gi * 2], a1 = acc[(size_t)gi * 2 + 1];
        const double s2d = raw_z ? (double)iv * (a1 - (double)mu * a0) : a1;
        const float s1 = (float)a0, s2 = (float)s2d;
        const float A_ = (gamma ? gamma[ch] : 1.f) * iv;
        A[gi] = A_;
        B[gi] = -A_ * iv * s2 * rm;
        Cc[gi] = A_ * (mu * iv * s2 - s1) * rm;
        sa += a0;
        sb += s2d;
    }
    if (dbeta) dbeta[ch] += (float)sa;
    if (dgamma) dgamma[ch] += (float)sb;
}

// Pixel splits of a streaming BN launch: ~32 pixels per thread on the big layers, but at least ~1024 workgroups in
// total while a thread still gets 4 pixels — the small late layers are latency-bound, a longer per-thread loop only
// adds to their time.
inline int stream_splits(int nb, int hw, int G, int c) {
    const int64_t npix = (int64_t)((nb + G - 1) / G) * hw;
    int64_t s = (npix + 1023) / 1024;
    const int64_t want = 1024 / ((int64_t)((c + 63) / 64) * G) + 1, most = (npix + 127) / 128;
    if (s < want) s = want < most ? want : most;
    return (int)(s < 1 ? 1 : (s > 65535 ? 65535 : s));
}

// Pool backward as a GATHER over the input pixels (no atomics: 16-bit storage has none worth using, and the result
// is deterministic): an input pixel visits the windows that contain it.  max: it receives a window's gradient when
// it is that window's first maximum in scan order (tf MaxPoolGrad / torch); avg: dy / #valid taps of every window.
template <typename E, typename T, int VEC>
__global__ __launch_bounds__(256) void pool2d_bwd_lp(const E* __restrict__ x, int x_ld,
                                                     const E* __restrict__ dy, int dy_ld, int nb, int ih,
                                                     int iw, int c, int kh, int kw, int stride, int pad_t, int pad_l,
                                                     int oh, int ow, int mode, int store, E* __restrict__ dx,
                                                     int dx_ld) {
    const int cg = c / VEC;
    const int64_t total = (int64_t)nb * ih * iw * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        const int64_t pix = idx / cg;
        const int ix = (int)(pix % iw);
        const int64_t t = pix / iw;
        const int iy = (int)(t % ih);
        const int n = (int)(t / ih);
        // windows (oy, ox) with oy*stride - pad_t <= iy <= oy*stride - pad_t + kh - 1
        int oy0 = iy + pad_t - kh + 1;
        oy0 = oy0 <= 0 ? 0 : (oy0 + stride - 1) / stride;
        int oy1 = (iy + pad_t) / stride;
        if (oy1 > oh - 1) oy1 = oh - 1;
        int ox0 = ix + pad_l - kw + 1;
        ox0 = ox0 <= 0 ? 0 : (ox0 + stride - 1) / stride;
        int ox1 = (ix + pad_l) / stride;
        if (ox1 > ow - 1) ox1 = ow - 1;
        if (oy0 > oy1 || ox0 > ox1) {                            // no window covers this pixel
            if (store) {
                float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                store_v<T, VEC>(dx + pix * dx_ld + q * VEC, zero);
            }
            continue;
        }
        float sum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = 0.f;
        float mine[8];
        if (mode == GV_POOL_MAX) load_v<T, VEC>(x + pix * x_ld + q * VEC, mine);
        for (int oy = oy0; oy <= oy1; ++oy) {
            for (int ox = ox0; ox <= ox1; ++ox) {
                float g[8];
                load_v<T, VEC>(dy + ((int64_t)(n * oh + oy) * ow + ox) * dy_ld + q * VEC, g);
                if (mode == GV_POOL_MAX) {
                    // this pixel wins when no earlier tap is >= it and no later tap is > it
                    bool win[8];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) win[e] = true;
                    bool before = true;
                    for (int r = 0; r < kh; ++r) {
                        const int yy = oy * stride + r - pad_t;
                        if ((unsigned)yy >= (unsigned)ih) continue;
                        for (int s = 0; s < kw; ++s) {
                            const int xx = ox * stride + s - pad_l;
                            if ((unsigned)xx >= (unsigned)iw) continue;
                            if (yy == iy && xx == ix) { before = false; continue; }
                            float v[8];
                            load_v<T, VEC>(x + ((int64_t)(n * ih + yy) * iw + xx) * x_ld + q * VEC, v);
#pragma unroll
                            for (int e = 0; e < VEC; ++e)
                                win[e] = win[e] && (before ? !(v[e] >= mine[e]) : !(v[e] > mine[e]));
                        }
                    }
#pragma unroll
                    for (int e = 0; e < VEC; ++e) sum[e] += win[e] ? g[e] : 0.f;
                } else {
                    int cnt = 0;
                    for (int r = 0; r < kh; ++r) {
                        if ((unsigned)(oy * stride + r - pad_t) >= (unsigned)ih) continue;
                        for (int s = 0; s < kw; ++s) cnt += (unsigned)(ox * stride + s - pad_l) < (unsigned)iw;
                    }
                    const float rc = 1.f / (float)cnt;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) sum[e] += g[e] * rc;
                }
            }
        }
        float d[8];
        if (store) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = 0.f;
        } else {
            load_v<T, VEC>(dx + pix * dx_ld + q * VEC, d);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] += sum[e];
        store_v<T, VEC>(dx + pix * dx_ld + q * VEC, d);
    }
}

// ---- max pool with a recorded argmax ---------------------------------------------------------------------------------
// Forward: y = max over the window AND, per output element, the row-major tap index of the FIRST maximum (one byte).
// Backward: an input pixel receives dy of every window whose recorded tap is that pixel — no input re-read, no
// nine-tap compare chain: 3x3 / stride 2 costs 2 loads per input pixel (a thread owns a 2x2 block and reads the four
// windows that touch it once) against the 11 of the recomputing kernel above, and the HBM side drops the forward input
// (MaxPool_3a at 32 x 12 views: 1.45 GB -> 0.8 GB).
template <typename E, typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_argmax_fwd(const E* __restrict__ x, int x_ld, int nb, int ih, int iw, int c,
                                                          int kh, int kw, int stride, int pad_t, int pad_l, int oh, int ow,
                                                          E* __restrict__ y, int y_ld, unsigned char* __restrict__ arg) {
    const int cg = c / VEC;
    const int64_t total = (int64_t)nb * oh * ow * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        const int64_t opix = idx / cg;
        const int ox = (int)(opix % ow);
        const int64_t t = opix / ow;
        const int oy = (int)(t % oh);
        const int n = (int)(t / oh);
        float best[8];
        int a[8];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; a[e] = -1; }
        for (int r = 0; r < kh; ++r) {
            const int yy = oy * stride - pad_t + r;
            if ((unsigned)yy >= (unsigned)ih) continue;
            for (int s_ = 0; s_ < kw; ++s_) {
                const int xx = ox * stride - pad_l + s_;
                if ((unsigned)xx >= (unsigned)iw) continue;
                float v[8];
                load_v<T, VEC>(x + ((int64_t)(n * ih + yy) * iw + xx) * x_ld + q * VEC, v);
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (v[e] > best[e] || a[e] < 0) { best[e] = v[e]; a[e] = r * kw + s_; }
            }
        }
        store_v<T, VEC>(y + opix * y_ld + q * VEC, best);
        unsigned char* ap = arg + opix * c + q * VEC;
        if constexpr (VEC == 8) {
            uint2 w;
            w.x = (unsigned)a[0] | ((unsigned)a[1] << 8) | ((unsigned)a[2] << 16) | ((unsigned)a[3] << 24);
            w.y = (unsigned)a[4] | ((unsigned)a[5] << 8) | ((unsigned)a[6] << 16) | ((unsigned)a[7] << 24);
            *reinterpret_cast<uint2*>(ap) = w;
        } else if constexpr (VEC == 4) {
            *reinterpret_cast<unsigned*>(ap) = (unsigned)a[0] | ((unsigned)a[1] << 8) | ((unsigned)a[2] << 16) | ((unsigned)a[3] << 24);
        } else {
            ap[0] = (unsigned char)a[0];
        }
    }
}

template <int VEC>
__device__ __forceinline__ void load_arg(const unsigned char* p, int (&a)[8]) {
    if constexpr (VEC == 8) {
        const uint2 w = *reinterpret_cast<const uint2*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (w.x >> (8 * e)) & 0xff; a[4 + e] = (w.y >> (8 * e)) & 0xff; }
    } else if constexpr (VEC == 4) {
        const unsigned w = *reinterpret_cast<const unsigned*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = (w >> (8 * e)) & 0xff;
    } else {
        a[0] = p[0];
    }
}

// any window: one input pixel x VEC channels per thread, the windows that contain it
template <typename E, typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_argmax_bwd(const unsigned char* __restrict__ arg, const E* __restrict__ dy,
                                                          int dy_ld, int nb, int ih, int iw, int c, int kh, int kw,
                                                          int stride, int pad_t, int pad_l, int oh, int ow, int store,
                                                          E* __restrict__ dx, int dx_ld) {
    const int cg = c / VEC;
    const int64_t total = (int64_t)nb * ih * iw * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        const int64_t pix = idx / cg;
        const int ix = (int)(pix % iw);
        const int64_t t = pix / iw;
        const int iy = (int)(t % ih);
        const int n = (int)(t / ih);
        float sum[8];
#pragma unroll
        for (int e = 0; e < VEC; ++e) sum[e] = 0.f;
        const int py = iy + pad_t, px = ix + pad_l;
        const int oy_hi = min(oh - 1, py / stride), ox_hi = min(ow - 1, px / stride);
        const int oy_lo = py - kh + 1 <= 0 ? 0 : (py - kh + stride) / stride;      // ceil((py - kh + 1) / stride)
        const int ox_lo = px - kw + 1 <= 0 ? 0 : (px - kw + stride) / stride;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int tap = (py - oy * stride) * kw + (px - ox * stride);
                const int64_t opix = (int64_t)(n * oh + oy) * ow + ox;
                int a[8];
                float g[8];
                load_arg<VEC>(arg + opix * c + q * VEC, a);
                load_v<T, VEC>(dy + opix * dy_ld + q * VEC, g);
#pragma unroll
                for (int e = 0; e < VEC; ++e) sum[e] += a[e] == tap ? g[e] : 0.f;
            }
        E* dp = dx + pix * dx_ld + q * VEC;
        float d[8];
        if (store) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[e] = 0.f;
        } else {
            load_v<T, VEC>(dp, d);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] += sum[e];
        store_v<T, VEC>(dp, d);
    }
}

// 3x3 / stride 2 / VALID: the 2x2 input block (2a..2a+1, 2b..2b+1) and the four windows (a-1..a, b-1..b) touching it
// BnTail (z != nullptr): the max pool follows relu(BN_train(z)) and pools z itself (BN + ReLU with a positive scale are
// monotone: max(relu(bn(z))) = relu(bn(max z)), same winner) — the gradient this kernel gathers is then the gradient of
// the BatchNorm's OUTPUT at that pixel, and instead of storing it the kernel finishes the BatchNorm backward pass:
// dz = A*g + B*z + C with g = dy*[z*scale + shift > 0] and the per-(group, channel) coefficients of gv_bn_bwd_coeffs_t.
// The activation y and its gradient are never materialised at the un-pooled size.
struct BnTail {
    const void* z;
    int z_ld, G;
    const float *A, *B, *C, *scale, *shift;                          // [G][c]
};

template <typename E, typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool3s2_argmax_bwd(const unsigned char* __restrict__ arg, const E* __restrict__ dy,
                                                             int dy_ld, int nb, int ih, int iw, int c, int oh, int ow,
                                                             int store, E* __restrict__ dx, int dx_ld, BnTail bn) {
    const int cg = c / VEC, ah = (ih + 1) / 2, aw = (iw + 1) / 2;
    const int64_t total = (int64_t)nb * ah * aw * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const unsigned per_img = (unsigned)(ah * aw * cg);           // (one image's work fits 32 bits: checked by the launcher)
        const int n = (int)(idx / per_img);
        unsigned t = (unsigned)(idx - (int64_t)n * per_img);
        const int q = (int)(t % (unsigned)cg);
        t /= (unsigned)cg;
        const int b = (int)(t % (unsigned)aw);
        const int a = (int)(t / (unsigned)aw);
        float sum[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum[k][e] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = a - 1 + wy, ox = b - 1 + wx;
                if ((unsigned)oy >= (unsigned)oh || (unsigned)ox >= (unsigned)ow) continue;
                const int64_t opix = (int64_t)(n * oh + oy) * ow + ox;
                int am[8];
                float g[8];
                load_arg<VEC>(arg + opix * c + q * VEC, am);
                load_v<T, VEC>(dy + opix * dy_ld + q * VEC, g);
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int tr = py + 2 * (1 - wy), tc = px + 2 * (1 - wx);
                        if (tr > 2 || tc > 2) continue;
                        const int tap = tr * 3 + tc;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) sum[2 * py + px][e] += am[e] == tap ? g[e] : 0.f;
                    }
            }
        }
        float cA[8], cB[8], cC[8], cs[8], ch_[8];
        if (bn.z) {
            const int gi = (n % bn.G) * c + q * VEC;
            auto ld = [&](const float* p, float* o) {                // (c % VEC == 0 and 16-byte aligned tables: vector loads)
                if constexpr (VEC % 4 == 0) {
#pragma unroll
                    for (int e = 0; e < VEC; e += 4) {
                        const float4 v = *reinterpret_cast<const float4*>(p + gi + e);
                        o[e] = v.x; o[e + 1] = v.y; o[e + 2] = v.z; o[e + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o[e] = p[gi + e];
                }
            };
            ld(bn.A, cA); ld(bn.B, cB); ld(bn.C, cC);
            if (bn.scale) { ld(bn.scale, cs); ld(bn.shift, ch_); }
            else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { cs[e] = 0.f; ch_[e] = 1.f; }
            }
        }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int iy = 2 * a + py, ix = 2 * b + px;
                if (iy >= ih || ix >= iw) continue;
                E* dp = dx + ((int64_t)(n * ih + iy) * iw + ix) * dx_ld + q * VEC;
                float d[8];
                if (bn.z) {                                      // dz = A*g + B*z + C, stored
                    float zv[8];
                    load_v<T, VEC>(reinterpret_cast<const E*>(bn.z) + ((int64_t)(n * ih + iy) * iw + ix) * bn.z_ld + q * VEC, zv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float g = fmaf(zv[e], cs[e], ch_[e]) > 0.f ? sum[2 * py + px][e] : 0.f;
                        d[e] = fmaf(cA[e], g, fmaf(cB[e], zv[e], cC[e]));
                    }
                    store_v<T, VEC>(dp, d);
                    continue;
                }
                if (store) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) d[e] = 0.f;
                } else {
                    load_v<T, VEC>(dp, d);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) d[e] += sum[2 * py + px][e];
                store_v<T, VEC>(dp, d);
            }
    }
}

// 3x3 / stride 2 / VALID max pool (every max pool of Inception-v3): a thread owns the 2x2 input pixels (2a..2a+1,
// 2b..2b+1) x 8 channels and visits the four windows (a-1..a, b-1..b) that touch them ONCE each — argmax per window
// from its nine taps, then the gradient goes to whichever of the thread's pixels is that tap: 11 loads per input pixel
// instead of the 23 of the per-pixel gather above.
template <typename E, typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool3s2_bwd_lp(const E* __restrict__ x, int x_ld,
                                                         const E* __restrict__ dy, int dy_ld, int nb,
                                                         int ih, int iw, int c, int oh, int ow, int store,
                                                         E* __restrict__ dx, int dx_ld) {
    const int cg = c / VEC, ah = (ih + 1) / 2, aw = (iw + 1) / 2;
    const int64_t total = (int64_t)nb * ah * aw * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cg);
        int64_t t = idx / cg;
        const int b = (int)(t % aw);
        t /= aw;
        const int a = (int)(t % ah);
        const int n = (int)(t / ah);
        float sum[4][8];                                         // [2*dy + dx of the 2x2 block]
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum[k][e] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = a - 1 + wy, ox = b - 1 + wx;
                if ((unsigned)oy >= (unsigned)oh || (unsigned)ox >= (unsigned)ow) continue;
                float best[8];
                int arg[8];
#pragma unroll
                for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; arg[e] = -1; }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int yy = 2 * oy + tap / 3, xx = 2 * ox + tap % 3;      // in bounds: VALID padding
                    float v[8];
                    load_v<T, VEC>(x + ((int64_t)(n * ih + yy) * iw + xx) * x_ld + q * VEC, v);
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (v[e] > best[e] || arg[e] < 0) { best[e] = v[e]; arg[e] = tap; }
                }
                float g[8];
                load_v<T, VEC>(dy + ((int64_t)(n * oh + oy) * ow + ox) * dy_ld + q * VEC, g);
                // the thread's pixel (2a+py, 2b+px) is tap (2a+py-2oy, 2b+px-2ox) = (py + 2*(1-wy), px + 2*(1-wx))
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int tr = py + 2 * (1 - wy), tc = px + 2 * (1 - wx);
                        if (tr > 2 || tc > 2) continue;
                        const int tap = tr * 3 + tc;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) sum[2 * py + px][e] += arg[e] == tap ? g[e] : 0.f;
                    }
            }
        }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int iy = 2 * a + py, ix = 2 * b + px;
                if (iy >= ih || ix >= iw) continue;
                E* dp = dx + ((int64_t)(n * ih + iy) * iw + ix) * dx_ld + q * VEC;
                float d[8];
                if (store) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) d[e] = 0.f;
                } else {
                    load_v<T, VEC>(dp, d);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) d[e] += sum[2 * py + px][e];
                store_v<T, VEC>(dp, d);
            }
    }
}

// Backward of the fused view pooling + group fusion (view_pool_fuse_bwd_f32 in train.hip) with F and dF in the
// storage type and dS in fp32.
template <typename T>
__global__ __launch_bounds__(256) void view_pool_fuse_bwd_lp(
    const unsigned short* __restrict__ F, const float* __restrict__ dS, int V, int N, int64_t E, int64_t view_stride,
    int64_t shape_stride, const int* __restrict__ scheme, int G, const float* __restrict__ weight, int mode,
    unsigned short* __restrict__ dF, int64_t scheme_stride, int64_t weight_stride) {
    __shared__ unsigned long long s_mask[64];
    __shared__ float s_w[64];
    __shared__ float s_wsum;
    const int n = blockIdx.y;
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
    if (s_wsum == 0.f) return;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const size_t base = (size_t)n * shape_stride + e;
        const float ds = dS[(size_t)n * E + e];
        for (int g = 0; g < G; ++g) {
            const unsigned long long m0 = s_mask[g];
            if (m0 == 0) continue;
            const float coef = s_w[g] / s_wsum * ds;
            if (mode == GV_VIEWPOOL_MEAN) {
                const float each = coef / (float)__popcll(m0);
                for (unsigned long long m = m0; m; m &= m - 1) {
                    const size_t a = base + (size_t)(__ffsll((long long)m) - 1) * view_stride;
                    dF[a] = down<T>(up<T>(dF[a]) + each);
                }
            } else {
                float best = -INFINITY;
                for (unsigned long long m = m0; m; m &= m - 1)
                    best = fmaxf(best, up<T>(F[base + (size_t)(__ffsll((long long)m) - 1) * view_stride]));
                int ties = 0;
                for (unsigned long long m = m0; m; m &= m - 1)
                    ties += up<T>(F[base + (size_t)(__ffsll((long long)m) - 1) * view_stride]) == best;
                const float each = coef / (float)ties;
                for (unsigned long long m = m0; m; m &= m - 1) {
                    const size_t a = base + (size_t)(__ffsll((long long)m) - 1) * view_stride;
                    if (up<T>(F[a]) == best) dF[a] = down<T>(up<T>(dF[a]) + each);
                }
            }
        }
    }
}

// ---- filter gradient on the 16-bit MFMA ----------------------------------------------------------------------------
// dW[tap][ci][co] += sum_m X[shift_tap(m)][ci] * dZ[m][co].  One TN GEMM per filter tap; a workgroup owns a
// (64*TI) x (64*TO) tile of one tap and a slice of the pixels (slices are combined with fp32 atomics), 2x2 waves of
// TI x TO accumulators.  32 pixels per stage, LDS double buffered, the next stage's global loads are issued before
// this stage's MFMAs, one barrier per stage.
//   LDS rows = pixels, 2*B + 64 bytes apart: the four rows a transposed read touches per 16-lane group then start 16
//   dwords apart (conflict-free), and the 16-byte writes of 8 consecutive lanes fill one row's 128 bytes.
//   Operand of a 32-channel tile and 16 pixels k0..k0+15: lane l (channel c0 + l%32) needs pixels k0 + 8*(l/32) + 0..7 =
//   two ds_read_b64_tr_b16; in each, lane 4q+p of a 16-lane group supplies row q, channels 4p..4p+3 of the group's
//   16 channels.  Both operands use the same pixel order, so any consistent order is a valid k order.
template <typename T>
__device__ __forceinline__ f32x16 mfma16(s16x8 a, s16x8 b, f32x16 c) {
    if constexpr (__is_same(T, __bf16))
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ s16x4 lds_read_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

template <typename T, int TI, int TO>
__global__ __launch_bounds__(256) void conv_wgrad_lp(const unsigned short* __restrict__ x, int x_ld,
                                                     const unsigned short* __restrict__ dz, int dz_ld, int nb, int ih,
                                                     int iw, int cin, int kh, int kw, int stride, int pad_t,
                                                     int pad_l, int oh, int ow, int cout, int64_t M,
                                                     int64_t m_per_block, const GvDw dw) {
    constexpr int PT = 32, BI = 64 * TI, BO = 64 * TO;
    constexpr int SX = 2 * BI + 64, SZ = 2 * BO + 64;                  // row strides in bytes
    constexpr int XV = PT * BI / 8 / 256, ZV = PT * BO / 8 / 256;      // 16-byte loads per thread and stage
    __shared__ __attribute__((aligned(16))) unsigned char sX[2][PT * SX];
    __shared__ __attribute__((aligned(16))) unsigned char sZ[2][PT * SZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int ntile_co = (cout + BO - 1) / BO, ntile_ci = (cin + BI - 1) / BI;
    // 1-D grid, XCD-aware: all tiles (taps x ci x co) of one PIXEL SLICE re-read the same X and dZ rows, so they get
    // consecutive logical ids = one XCD's L2 (round-robin dispatch would spread them over all eight)
    const int tiles = ntile_co * ntile_ci * kh * kw;
    const int logical = gv_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int b = logical % tiles;
    const int slice = logical / tiles;
    const int tco = b % ntile_co; b /= ntile_co;
    const int tci = b % ntile_ci; b /= ntile_ci;
    const int tap = b;
    const int fr = tap / kw, fs = tap - fr * kw;
    const int ci0 = tci * BI, co0 = tco * BO;
    const int64_t m0 = (int64_t)slice * m_per_block;
    const int64_t m1 = m0 + m_per_block < M ? m0 + m_per_block : M;
    f32x16 acc[TI][TO];
#pragma unroll
    for (int t = 0; t < TI; ++t)
#pragma unroll
        for (int u = 0; u < TO; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    const int ohow = oh * ow;
    u32x4 xr[XV], zr[ZV];
    auto load = [&](int64_t mt) {
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int idx = tid + j * 256;
            const int p = idx / (BI / 8), c = ci0 + (idx % (BI / 8)) * 8;
            const int64_t m = mt + p;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < m1 && c < cin) {
                const int n = (int)(m / ohow);
                const int rem = (int)(m - (int64_t)n * ohow);
                const int oy = rem / ow, ox = rem - oy * ow;
                const int iy = oy * stride + fr - pad_t, ix = ox * stride + fs - pad_l;
                if ((unsigned)iy < (unsigned)ih && (unsigned)ix < (unsigned)iw)
                    v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * ih + iy) * iw + ix) * x_ld + c);
            }
            xr[j] = v;
        }
#pragma unroll
        for (int j = 0; j < ZV; ++j) {
            const int idx = tid + j * 256;
            const int p = idx / (BO / 8), c = co0 + (idx % (BO / 8)) * 8;
            const int64_t m = mt + p;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < m1 && c < cout) v = *reinterpret_cast<const u32x4*>(dz + (size_t)m * dz_ld + c);
            zr[j] = v;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < XV; ++j) {
            const int idx = tid + j * 256;
            *reinterpret_cast<u32x4*>(&sX[buf][(idx / (BI / 8)) * SX + (idx % (BI / 8)) * 16]) = xr[j];
        }
#pragma unroll
        for (int j = 0; j < ZV; ++j) {
            const int idx = tid + j * 256;
            *reinterpret_cast<u32x4*>(&sZ[buf][(idx / (BO / 8)) * SZ + (idx % (BO / 8)) * 16]) = zr[j];
        }
    };
    // transposed-read address of this lane inside a (16-pixel, 32-channel) operand block
    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int row_l = 8 * (g16 >> 1) + q;                              // + 4*j for the second read
    const int col_l = 16 * (g16 & 1) + 4 * p4;                         // channel within the 32-channel tile
    const int xo = row_l * SX + 2 * ((wi * TI) * 32 + col_l);
    const int zo = row_l * SZ + 2 * ((wj * TO) * 32 + col_l);
    load(m0);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int64_t mt = m0; mt < m1; mt += PT) {
        const bool more = mt + PT < m1;
        if (more) load(mt + PT);
#pragma unroll
        for (int k = 0; k < PT; k += 16) {
            s16x8 av[TI], bv[TO];
#pragma unroll
            for (int t = 0; t < TI; ++t) {
                const unsigned char* pa = &sX[buf][xo + k * SX + t * 64];
                const s16x4 lo = lds_read_tr(pa), hi = lds_read_tr(pa + 4 * SX);
                av[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int u = 0; u < TO; ++u) {
                const unsigned char* pb = &sZ[buf][zo + k * SZ + u * 64];
                const s16x4 lo = lds_read_tr(pb), hi = lds_read_tr(pb + 4 * SZ);
                bv[u] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int t = 0; t < TI; ++t)
#pragma unroll
                for (int u = 0; u < TO; ++u) acc[t][u] = mfma16<T>(av[t], bv[u], acc[t][u]);
        }
        if (more) store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int u = 0; u < TO; ++u) {
        const int col = co0 + (wj * TO + u) * 32 + li;
        if (col >= cout) continue;
#pragma unroll
        for (int t = 0; t < TI; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + (wi * TI + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < cin) gv_dw_put(dw, slice, ((size_t)tap * cin + ci) * cout + col, acc[t][u][r]);
            }
    }
}

// ---- filter gradient, strip form (stride-1 filters with more than one tap) ----------------------------------------
// The tap-per-workgroup kernel above re-reads dZ and X once per filter tap and is bound by L2->LDS traffic at
// 32-64 flop/byte.  Here a workgroup owns a tile (ci x co) for a GROUP OF TAPS (up to 9) and walks over "strips":
// up to 32 output pixels = R consecutive rows x L columns of one image.  Per strip it loads dZ ONCE ([32 slots][co])
// and the input patch with its halo ONCE ([(R+kh-1) x (L+kw-1) pixels][ci]); tap (r,s) of output slot (rho, ox) is
// LDS row (rho+r)*(L+kw-1) + ox+s, and because every lane of ds_read_b64_tr_b16 supplies its own row address the
// k-slot -> LDS-row map is free: the same patch serves every tap at a wave-uniform byte offset.  Accumulators: one
// 32x32 tile per (wave, tap).  Waves = WI x WJ x WT: WI x WJ sub-tiles of 32 channels, WT-way split of the taps
// (few-channel stem layers have only one 32x32 tile per tap).  One LDS buffer; the next strip's global loads are in
// flight (registers) while this strip's MFMAs run.
struct StripGeom {
    int nb, ih, iw, cin, kh, kw, pad_t, pad_l, oh, ow, cout;
    int R, L, Wx, Hx;             // strip rows/cols, patch width/height
    int nrs, ncs;                 // strips per image along y / x
    int tap0, ntaps;              // tap group of this launch slice: taps tap0 .. tap0+ntaps-1 (set per blockIdx.z)
    int stages, stages_per_block;
    int xrows;                    // Hx*Wx
    int x_ld, dz_ld;
};

// KS: 16-pixel k-steps per strip.  2: the strip of up to 32 pixels described above.  16 (the DEEP form, 32-channel inputs:
// Conv2d_2a / 2b): a strip is 8 rows x 32 columns of one image — the tile of the forward halo kernel — so that one fill of
// LDS (16 KB of dZ per 32 columns of co, 22 KB of input patch) and one barrier pair feed 16 x (taps per wave) MFMAs per
// wave instead of 2 x: the 32-pixel strips of a 109-wide map spend their time at barriers (6 MFMAs per wave between two).
template <typename T, int WI, int WJ, int WT, int NTW, int KS = 2>
__global__ __launch_bounds__(256) void conv_wgrad_strip_lp(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ dz, StripGeom gm,
                                                           int taps_per_group, const GvDw dw) {
    static_assert(WI * WJ * WT == 4, "four waves");
    static_assert(KS == 2 || WI == 1, "the deep form: 32 input channels");
    constexpr int SLOTS = 16 * KS;
    constexpr int BI = 32 * WI, BO = 32 * WJ;
    // row strides: the four rows a transposed read touches per 16-lane group must start 16 dwords apart; a 32-channel
    // row IS 16 dwords (no padding: 2*B + 64 = 128 bytes would put rows q and q+2 on the same banks — measured 43 %
    // bank conflicts on the stem layers before this)
    constexpr int SX = BI == 32 ? 64 : 2 * BI + 64, SZ = BO == 32 ? 64 : 2 * BO + 64;
    constexpr int XC = BI / 8, ZC = BO / 8;                      // 16-byte chunks per LDS row
    constexpr int XVMAX = 6, ZV = (SLOTS * ZC + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* sZ = lds;
    unsigned char* sX = lds + SLOTS * SZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave % WT, wj = (wave / WT) % WJ, wi = wave / (WT * WJ);
    const int ntile_co = (gm.cout + BO - 1) / BO, ntile_ci = (gm.cin + BI - 1) / BI;
    const int groups = (gm.kh * gm.kw + taps_per_group - 1) / taps_per_group;
    const int tiles = ntile_co * ntile_ci * groups;
    const int logical = gv_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // tiles of one strip range share an XCD's L2
    int b = logical % tiles;
    const int tco = b % ntile_co; b /= ntile_co;
    const int tci = b % ntile_ci; b /= ntile_ci;
    const int tap0 = b * taps_per_group;
    const int ntaps = min(taps_per_group, gm.kh * gm.kw - tap0);
    const int ci0 = tci * BI, co0 = tco * BO;
    const int s0 = (logical / tiles) * gm.stages_per_block;
    const int s1 = min(s0 + gm.stages_per_block, gm.stages);
    if (s0 >= s1) return;
    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // loader roles (strip-invariant): X chunk j of this thread -> patch pixel (py, px) and channel chunk
    const int xchunks = gm.xrows * XC;
    int xpy[XVMAX], xpx[XVMAX], xoff[XVMAX], xch[XVMAX];
#pragma unroll
    for (int j = 0; j < XVMAX; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / XC;
        xch[j] = ci0 + (idx % XC) * 8;
        xpy[j] = row / gm.Wx;
        xpx[j] = row - xpy[j] * gm.Wx;
        xoff[j] = idx < xchunks ? row * SX + (idx % XC) * 16 : -1;
    }
    int zr[ZV], zc[ZV], zch[ZV], zoff[ZV];
#pragma unroll
    for (int j = 0; j < ZV; ++j) {
        const int idx = tid + j * 256;
        const int p = idx / ZC;                                  // k slot
        zr[j] = p / gm.L;
        zc[j] = p - zr[j] * gm.L;
        zch[j] = co0 + (idx % ZC) * 8;
        zoff[j] = idx < SLOTS * ZC ? p * SZ + (idx % ZC) * 16 : -1;
        if (zr[j] >= gm.R) zr[j] = -1;                           // padding slot: always zero
    }
    u32x4 xq[XVMAX], zq[ZV];
    auto load = [&](int stage) {
        const int cs = stage % gm.ncs;
        const int t = stage / gm.ncs;
        const int rs = t % gm.nrs;
        const int n = t / gm.nrs;
        const int oy0 = rs * gm.R, ox0 = cs * gm.L;
        const int lv = min(gm.L, gm.ow - ox0);                   // valid columns of this strip
#pragma unroll
        for (int j = 0; j < XVMAX; ++j) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (xoff[j] >= 0) {
                const int iy = oy0 - gm.pad_t + xpy[j], ix = ox0 - gm.pad_l + xpx[j];
                if ((unsigned)iy < (unsigned)gm.ih && (unsigned)ix < (unsigned)gm.iw && xch[j] < gm.cin)
                    v = *reinterpret_cast<const u32x4*>(x + (((size_t)n * gm.ih + iy) * gm.iw + ix) * gm.x_ld + xch[j]);
            }
            xq[j] = v;
        }
#pragma unroll
        for (int j = 0; j < ZV; ++j) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (zoff[j] >= 0 && zr[j] >= 0) {
                const int oy = oy0 + zr[j];
                if (oy < gm.oh && zc[j] < lv && zch[j] < gm.cout)
                    v = *reinterpret_cast<const u32x4*>(dz + (((size_t)n * gm.oh + oy) * gm.ow + ox0 + zc[j]) * gm.dz_ld + zch[j]);
            }
            zq[j] = v;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < XVMAX; ++j)
            if (xoff[j] >= 0) *reinterpret_cast<u32x4*>(sX + xoff[j]) = xq[j];
#pragma unroll
        for (int j = 0; j < ZV; ++j)
            if (zoff[j] >= 0) *reinterpret_cast<u32x4*>(sZ + zoff[j]) = zq[j];
    };
    // transposed-read addresses: k slot p = 16*ks + 8*(lane/32) + 4*j + q  ->  dZ row p, X row (p/L)*Wx + p%L (+ tap)
    const int g16 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int col_l = 16 * (g16 & 1) + 4 * p4;
    // (deep form: L = 32, so slot p = 16 ks + c is row ks / 2, column 16 (ks & 1) + c of the strip: one base per j, a
    // compile-time-unrolled offset per k-step)
    int xb[2][2], zb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = 16 * ks + 8 * (g16 >> 1) + 4 * j + q;
            int rr = p / gm.L, cc = p - rr * gm.L;
            if (rr >= gm.R) { rr = 0; cc = 0; }                  // padding slot: dZ is zero there, any row will do
            xb[ks][j] = (rr * gm.Wx + cc) * SX + 2 * (wi * 32 + col_l);
            zb[ks][j] = p * SZ + 2 * (wj * 32 + col_l);
        }
    const int xstep = gm.Wx * SX;                                // deep form: one strip row further down the patch
    int tapoff[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tap = tap0 + wt + WT * i;
        const int fr = tap / gm.kw, fs = tap - fr * gm.kw;
        tapoff[i] = (fr * gm.Wx + fs) * SX;
    }
    load(s0);
    for (int stage = s0; stage < s1; ++stage) {
        store();
        __syncthreads();
        if (stage + 1 < s1) load(stage + 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // (KS = 2: the tables; deep: k-step ks = row ks / 2 of the strip, half ks & 1 of its 32 columns)
            const int zo = KS == 2 ? 0 : (ks >> 1) * 32 * SZ, xo = KS == 2 ? 0 : (ks >> 1) * xstep;
            const s16x4 blo = lds_read_tr(sZ + zb[ks & 1][0] + zo), bhi = lds_read_tr(sZ + zb[ks & 1][1] + zo);
            const s16x8 bv = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if (wt + WT * i < ntaps) {                       // wave-uniform
                    const s16x4 alo = lds_read_tr(sX + xb[ks & 1][0] + tapoff[i] + xo);
                    const s16x4 ahi = lds_read_tr(sX + xb[ks & 1][1] + tapoff[i] + xo);
                    const s16x8 av = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[i] = mfma16<T>(av, bv, acc[i]);
                }
            }
        }
        __syncthreads();
    }
    const int li = lane & 31, lh = lane >> 5;
    const int col = co0 + wj * 32 + li;
    if (col < gm.cout) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            if (wt + WT * i >= ntaps) continue;
            const int tap = tap0 + wt + WT * i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < gm.cin) gv_dw_put(dw, logical / tiles, ((size_t)tap * gm.cin + ci) * gm.cout + col, acc[i][r]);
            }
        }
    }
}

inline bool vec8(const void* p, int ld) { return p == nullptr || (gv_aligned16(p) && (ld % 8) == 0); }
inline bool vec4(const void* p, int ld) { return p == nullptr || (gv_aligned16(p) && (ld % 4) == 0); }

template <typename T>
int sums_t(int mode, const unsigned short* z, int z_ld, const unsigned short* dy, int dy_ld, const unsigned short* y,
           int y_ld, const float* mean, const float* inv, const float* scale, const float* shift, int nb, int hw, int c,
           int G, int splits, double* acc, hipStream_t st) {
    const bool v = (c % 8 == 0) && vec8(z, z_ld) && vec8(dy, dy_ld) && vec8(y, y_ld);
    const dim3 grid((c + 63) / 64, splits, G);
    if (v && (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll) {
#define GV_SUMS8(MODE)                                                                                               \
        hipLaunchKernelGGL((grouped_sums_v8<unsigned short, T, MODE>), grid, dim3(256), 0, st, z, z_ld, dy, dy_ld, y, y_ld, mean, inv, scale, \
                           shift, nb, hw, c, G, acc)
        if (mode == 0) GV_SUMS8(0);
        else if (mode == 1) GV_SUMS8(1);
        else GV_SUMS8(2);
#undef GV_SUMS8
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
#define GV_SUMS(MODE)                                                                                                \
    do {                                                                                                             \
        if (v)                                                                                                       \
            hipLaunchKernelGGL((grouped_sums_lp<T, MODE, 8>), grid, dim3(256), 0, st, z, z_ld, dy, dy_ld, y, y_ld, mean, \
                               inv, scale, shift, nb, hw, c, G, acc);                                                              \
        else                                                                                                         \
            hipLaunchKernelGGL((grouped_sums_lp<T, MODE, 1>), grid, dim3(256), 0, st, z, z_ld, dy, dy_ld, y, y_ld, mean, \
                               inv, scale, shift, nb, hw, c, G, acc);                                                              \
    } while (0)
    if (mode == 0) GV_SUMS(0);
    else if (mode == 1) GV_SUMS(1);
    else GV_SUMS(2);
#undef GV_SUMS
    GV_LAUNCH_CHECK();
    return GV_OK;
}

// Launch geometry: d->tile_cfg = 0 picks the side widths by divisibility and ~2048 workgroups; tile_cfg = 1 + tile +
// 9*split selects tile (TI,TO) in {1,2,3}^2 (64/128/192 channels per side) and a target of 1024 / 2048 / 4096
// workgroups; 28..30 the strip form (TrainGVCNN.autotune measures them per layer).
// strip form: geometry and eligibility
inline bool strip_geom(const gv_conv_desc* d, int dz_ld, int bi, StripGeom* gm, bool deep = false) {
    if (d->stride != 1 || d->kh * d->kw < 2) return false;
    StripGeom& g = *gm;
    g.nb = d->nb; g.ih = d->ih; g.iw = d->iw; g.cin = d->cin; g.kh = d->kh; g.kw = d->kw;
    g.pad_t = d->pad_t; g.pad_l = d->pad_l; g.oh = d->oh; g.ow = d->ow; g.cout = d->cout;
    g.x_ld = d->x_ld; g.dz_ld = dz_ld;
    if (deep) {                                                  // 8 rows x 32 columns per strip (KS = 16)
        if (bi != 32 || d->ow < 24 || d->oh < 8) return false;
        g.ncs = (d->ow + 31) / 32;
        g.L = 32;
        g.R = 8;
    } else if (d->ow >= 32) {
        g.ncs = (d->ow + 31) / 32;
        g.L = (d->ow + g.ncs - 1) / g.ncs;
        g.R = 1;
    } else {
        g.ncs = 1;
        g.L = d->ow;
        g.R = 32 / d->ow;
        if (g.R > d->oh) g.R = d->oh;
    }
    g.nrs = (d->oh + g.R - 1) / g.R;
    g.Wx = g.L + d->kw - 1;
    g.Hx = g.R + d->kh - 1;
    g.xrows = g.Hx * g.Wx;
    const int64_t stages = (int64_t)d->nb * g.nrs * g.ncs;
    if (stages > 0x7fffffff) return false;
    g.stages = (int)stages;
    if (g.xrows * (bi / 8) > 6 * 256) return false;              // XVMAX chunks per thread
    if (!deep && (int64_t)g.xrows * (2 * bi + 64) + 32 * (2 * 128 + 64) > 64 * 1024) return false;
    return true;
}

template <typename T, int WI, int WJ, int WT, int NTW, int KS = 2>
int strip_launch(const gv_conv_desc* d, const unsigned short* x, const unsigned short* dz, int dz_ld, const GvDw& dw,
                 int target_wgs, hipStream_t st) {
    constexpr int BI = 32 * WI, BO = 32 * WJ;
    StripGeom gm;
    if (!strip_geom(d, dz_ld, BI, &gm, KS != 2)) return GV_E_UNSUPPORTED;
    const int taps = d->kh * d->kw, tpg = NTW * WT;              // taps per workgroup
    const int groups = (taps + tpg - 1) / tpg;
    const int tiles = ((d->cin + BI - 1) / BI) * ((d->cout + BO - 1) / BO) * groups;
    int64_t splits = (target_wgs + tiles - 1) / tiles;
    const int64_t max_splits = KS == 2 ? (gm.stages + 7) / 8     // at least 8 strips per workgroup (deep: 3 of 256 pixels)
                                       : (gm.stages + 2) / 3;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const size_t elems = (size_t)taps * d->cin * d->cout;
    splits = gv_dw_clamp(dw, elems, splits);
    gm.stages_per_block = (int)((gm.stages + splits - 1) / splits);
    splits = (gm.stages + gm.stages_per_block - 1) / gm.stages_per_block;
    const size_t lds = (size_t)(16 * KS) * (BO == 32 ? 64 : 2 * BO + 64) + (size_t)gm.xrows * (BI == 32 ? 64 : 2 * BI + 64);
    auto kern = conv_wgrad_strip_lp<T, WI, WJ, WT, NTW, KS>;
    const bool attr = GV_BIG_LDS_OK(kern, KS == 2 ? 64 * 1024 : 96 * 1024);
    if (KS != 2 && !attr) return GV_E_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * splits)), dim3(256), lds, st, x, dz, gm, tpg,
                       gv_dw_sink(dw, elems, splits));
    GV_LAUNCH_CHECK();
    return gv_dw_finish(dw, elems, splits, st);
}

int g_strip_ntw = 5;     // taps per workgroup of the general strip variant: 5 keeps two waves per SIMD (9: one)

template <typename T>
int strip_t(const gv_conv_desc* d, const unsigned short* x, const unsigned short* dz, int dz_ld, const GvDw& dw,
            int target_wgs, hipStream_t st) {
    if (d->cin <= 32 && d->cout <= 32) return strip_launch<T, 1, 1, 4, 3>(d, x, dz, dz_ld, dw, target_wgs, st);
    if (d->cin <= 32) return strip_launch<T, 1, 2, 2, 5>(d, x, dz, dz_ld, dw, target_wgs, st);
    if (d->cout <= 32) return strip_launch<T, 2, 1, 2, 5>(d, x, dz, dz_ld, dw, target_wgs, st);
    if (g_strip_ntw == 5) return strip_launch<T, 2, 2, 1, 5>(d, x, dz, dz_ld, dw, target_wgs, st);
    if (g_strip_ntw == 4) return strip_launch<T, 2, 2, 1, 4>(d, x, dz, dz_ld, dw, target_wgs, st);
    return strip_launch<T, 2, 2, 1, 9>(d, x, dz, dz_ld, dw, target_wgs, st);
}

// the deep form (8 x 32 pixel strips): 32 input channels, 32 or 64 output channels
template <typename T>
int strip_deep_t(const gv_conv_desc* d, const unsigned short* x, const unsigned short* dz, int dz_ld, const GvDw& dw,
                 int target_wgs, hipStream_t st) {
    // (measured on wider layers too — Conv2d_4a 80 -> 192: equal to the LDS-DMA form at best; Mixed_5's 3x3 / 5x5 on 25 x 25
    // maps: 20-40 % slower — so the form stays with the layers it wins on)
    if (d->cin > 32 || d->cout > 64) return GV_E_UNSUPPORTED;
    if (d->cout <= 32) return strip_launch<T, 1, 1, 4, 3, 16>(d, x, dz, dz_ld, dw, target_wgs, st);
    return strip_launch<T, 1, 2, 2, 5, 16>(d, x, dz, dz_ld, dw, target_wgs, st);
}

int g_strip_default = 1;

template <typename T>
int wgrad_t(const gv_conv_desc* d, const unsigned short* x, const unsigned short* dz, int dz_ld, const GvDw& dw,
            hipStream_t st) {
    const int64_t M = (int64_t)d->nb * d->oh * d->ow;
    // side width 64 / 128 / 192 channels (TI, TO = 1..3): the widest that pads no more channels than 64-wide tiles
    // would (a 192-wide side triples the flop/byte of this L2->LDS bound kernel on the many 192-channel layers)
    auto side = [](int c) {
        const int base = (c + 63) / 64 * 64;
        if ((c + 191) / 192 * 192 == base) return 3;
        if ((c + 127) / 128 * 128 == base) return 2;
        return 1;
    };
    int ti = side(d->cin), to = side(d->cout);
    int64_t target = 2048;
    if (d->tile_cfg > 30 + gvlp::wgrad_dma_num_cfgs() + 5) return GV_E_BADARG;
    if (d->tile_cfg > 30 + gvlp::wgrad_dma_num_cfgs()) {         // 92..96: the deep strip form, 512 ... 4096 workgroups
        const int tws[5] = {512, 768, 1024, 2048, 4096};
        const int tw = tws[d->tile_cfg - 31 - gvlp::wgrad_dma_num_cfgs()];
        const int rc = strip_deep_t<T>(d, x, dz, dz_ld, dw, tw, st);
        return rc != GV_E_UNSUPPORTED ? rc : strip_t<T>(d, x, dz, dz_ld, dw, tw, st);   // (other layers: the 32-pixel strips)
    }
    if (d->tile_cfg > 30)                                        // 31..91: LDS-DMA staging (wgrad_dma.hip)
        return gvlp::conv_wgrad_dma_launch(d, x, dz, dz_ld, dw, d->tile_cfg - 31, st);
    if (d->tile_cfg > 27) {                                      // 28..30: strip form, 1024 / 2048 / 4096 workgroups
        return strip_t<T>(d, x, dz, dz_ld, dw, 1024 << (d->tile_cfg - 28), st);
    }
    // heuristic: the strip form for the few-channel stem layers (2.5-3.5x there); its general 64x64x9-tap variant
    // runs at one wave per SIMD and loses to the tap-per-workgroup tiles — autotune may still pick it (cfg 13-15)
    if (d->tile_cfg == 0 && g_strip_default && d->cin <= 32) {
        const int rc = strip_t<T>(d, x, dz, dz_ld, dw, 2048, st);
        if (rc != GV_E_UNSUPPORTED) return rc;
    }
    if (d->tile_cfg == 0) {
        // un-tuned default: the two-stage LDS-DMA form (fastest on 64 of Inception-v3's 67 layers), 128-channel sides
        // where the layer has them, ~2048 workgroups; the tap-per-workgroup tiles below where it does not take the layer
        const int shape = (d->cin >= 128 ? 1 : 0) + (d->cout >= 128 ? 2 : 0);
        const int rc = gvlp::conv_wgrad_dma_launch(d, x, dz, dz_ld, dw, 12 + 4 + shape, st);
        if (rc != GV_E_UNSUPPORTED) return rc;
    }
    if (d->tile_cfg > 0) {                                       // 1..27: tile (TI, TO) in {1,2,3}^2 x workgroup target
        const int k = d->tile_cfg - 1;
        ti = 1 + (k % 9) % 3;
        to = 1 + (k % 9) / 3;
        target = 1024 << (k / 9);
    }
    const int tiles = d->kh * d->kw * ((d->cin + 64 * ti - 1) / (64 * ti)) * ((d->cout + 64 * to - 1) / (64 * to));
    int64_t splits = (target + tiles - 1) / tiles;
    const int64_t max_splits = (M + 511) / 512;                  // at least 512 pixels per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    splits = gv_dw_clamp(dw, elems, splits);
    int64_t per = (M + splits - 1) / splits;
    per = (per + 31) / 32 * 32;
    splits = (M + per - 1) / per;
    const dim3 grid((unsigned)(tiles * splits));
    const GvDw sink = gv_dw_sink(dw, elems, splits);
#define GV_WGRAD_LP(TI, TO)                                                                                          \
    hipLaunchKernelGGL((conv_wgrad_lp<T, TI, TO>), grid, dim3(256), 0, st, x, d->x_ld, dz, dz_ld, d->nb, d->ih, d->iw, \
                       d->cin, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, d->cout, M, per, sink)
    switch (ti * 10 + to) {
        case 11: GV_WGRAD_LP(1, 1); break;
        case 12: GV_WGRAD_LP(1, 2); break;
        case 13: GV_WGRAD_LP(1, 3); break;
        case 21: GV_WGRAD_LP(2, 1); break;
        case 22: GV_WGRAD_LP(2, 2); break;
        case 23: GV_WGRAD_LP(2, 3); break;
        case 31: GV_WGRAD_LP(3, 1); break;
        case 32: GV_WGRAD_LP(3, 2); break;
        default: GV_WGRAD_LP(3, 3); break;
    }
#undef GV_WGRAD_LP
    GV_LAUNCH_CHECK();
    return gv_dw_finish(dw, elems, splits, st);
}


// ---- filter gradient of the 3-channel stems (Conv2d_1a 3x3/2, ResNet conv1 7x7/2) on the 16-bit MFMA -----------------------
// dW[rho][co] = sum over pixels of X[pixel shifted by tap(rho)][ci(rho)] * dZ[pixel][co], rho = tap*cin + ci: 27 (147)
// rows of 32 (64) columns over millions of pixels — a reduction, HBM-bound by construction (Conv2d_1a: 116 MB of
// images + 303 MB of dZ per step).  The direct kernel (train.hip: conv_wgrad_direct_f32) gathers both operands with
// 2-byte loads, one per lane and pixel, and multiplies on the fp32 MFMA (2 pixels per instruction): 0.48 ms at 17 TFLOP/s.
// Here a WAVE owns a run of (image, output row) units.  Per unit it copies the kh input rows and the one dZ row into its
// private LDS area with 16-byte loads (both are contiguous in memory), one unit ahead in registers, and multiplies
// 16 pixels per v_mfma_f32_32x32x16: the dZ operand comes back through the transposing ds_read_b64_tr_b16 (lane =
// channel, 4 consecutive pixels per read), the image operand as 8 two-byte LDS reads per lane at a 2*stride*cin-byte
// pitch (row rho of the operand is one (filter row, column, channel) = one fixed offset into the staged rows).  No
// barrier inside the loop (a wave reads only what it wrote); the four waves' accumulators are summed in LDS and
// leave as one fp32 atomic per (rho, co) and workgroup.
// ZL: 16-byte loads per lane that hold the unit's dZ row (10: rows up to 640 chunks — Conv2d_1a's 111 x 32; 14: ResNet conv1's
// 112 pixels x 64 channels = 896 chunks, 96 KB of LDS per workgroup — one workgroup per CU, whose four waves keep 24 loads per
// lane in flight each: the layer was 2.75 ms on the fp32-MFMA fallback, 9 % of the ResNet training step).
template <typename T, int NRT, int NCT, int ZL = 10>
__global__ __launch_bounds__(256) void conv_wgrad_stem_rows_lp(const unsigned short* __restrict__ x,
                                                               const unsigned short* __restrict__ dz, int dz_ld, int nb,
                                                               int ih, int iw, int cin, int kh, int kw, int stride,
                                                               int pad_t, int pad_l, int oh, int ow, int cout, int xrow_b,
                                                               int zrows, int units_per_wave, const GvDw dw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int ZB = NCT * 64;                                       // bytes of one dZ pixel row in LDS
    const int wave_b = kh * xrow_b + zrows * ZB;                   // this wave's area: kh image rows, then the dZ row
    unsigned char* sXw = smem_w + wave * wave_b;
    unsigned char* sZw = sXw + kh * xrow_b;
    for (int i = lane * 16; i < wave_b; i += 1024) *reinterpret_cast<u32x4*>(sXw + i) = u32x4{0u, 0u, 0u, 0u};
    const int R = kh * kw * cin;
    // image rows sit pad_l pixels (+ xoff bytes, so that their 4-byte pieces stay aligned: ResNet conv1 pads 3 pixels of 6 bytes)
    // into their LDS rows; what is in front stays zero
    const int xoff = (4 - ((pad_l * cin * 2) & 3)) & 3;
    int abase[NRT];                                                // byte offset of operand row rho at output column 0
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        const int rho = min(t * 32 + li, R - 1);                   // (rows past R: any finite data; never stored)
        const int tap = rho / cin, ci = rho - tap * cin;
        const int r = tap / kw, sx = tap - r * kw;
        abase[t] = r * xrow_b + xoff + (sx * cin + ci) * 2;
    }
    const int apitch = stride * cin * 2;                           // bytes between consecutive output columns
    // transposed-read address of this lane inside a (16-pixel, 32-channel) block of the dZ row (see conv_wgrad_lp)
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int zoff = (8 * (g16 >> 1) + q4) * ZB + 2 * (16 * (g16 & 1) + 4 * p4);
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t units = (int64_t)nb * oh;
    const int64_t u0 = wid * units_per_wave;
    const int64_t u1 = u0 + units_per_wave < units ? u0 + units_per_wave : units;
    f32x16 acc[NRT][NCT];
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int u = 0; u < NCT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    constexpr int XL = NRT == 1 ? 6 : 10;                          // 16-byte loads per lane: image rows (capacity; ZL: the dZ row)
    const int xbytes = iw * cin * 2;                               // one image row (contiguous: x_ld == cin)
    const int xchunks = (kh * xbytes + 15) / 16, zchunks = ow * (NCT * 4);
    u32x4 xr[XL], zr[ZL];
    auto fetch = [&](int64_t u) {
        const int n = (int)(u / oh), oy = (int)(u - (int64_t)n * oh);
        const int iy0 = oy * stride - pad_t;
#pragma unroll
        for (int k = 0; k < XL; ++k) {
            const int c = lane + 64 * k;                           // chunk c covers bytes [16c, 16c + 16) of the kh rows
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < xchunks) {
                const int row = (c * 16) / xbytes;                 // (a chunk may straddle two rows: both must be inside)
                const int row2 = (c * 16 + 15) / xbytes;
                const int iyA = iy0 + row, iyB = iy0 + (row2 < kh ? row2 : kh - 1);
                if (iyA >= 0 && iyB < ih)
                    v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(x) +
                                                        ((size_t)n * ih + iy0) * xbytes + (size_t)c * 16);
            }
            xr[k] = v;
        }
#pragma unroll
        for (int k = 0; k < ZL; ++k) {
            const int c = lane + 64 * k;                           // (pixel, 8-channel chunk)
            u32x4 v = {0u, 0u, 0u, 0u};
            if (c < zchunks) {
                const int px = c / (NCT * 4), ch = (c - px * (NCT * 4)) * 8;
                if (ch < cout) v = *reinterpret_cast<const u32x4*>(dz + ((size_t)u * ow + px) * dz_ld + ch);
            }
            zr[k] = v;
        }
    };
    auto stage = [&]() {                                           // registers -> this wave's LDS area
#pragma unroll
        for (int k = 0; k < XL; ++k) {
            const int c = lane + 64 * k;
            if (c < xchunks) {
                const int b0 = c * 16;                             // (rows xrow_b apart)
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {                   // 4-byte pieces: a chunk may straddle two rows
                    const int b = b0 + 4 * w4, row = b / xbytes, off = b - row * xbytes;
                    if (row < kh) *reinterpret_cast<unsigned*>(sXw + row * xrow_b + xoff + pad_l * cin * 2 + off) = xr[k][w4];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < ZL; ++k) {
            const int c = lane + 64 * k;
            if (c < zchunks) *reinterpret_cast<u32x4*>(sZw + (c / (NCT * 4)) * ZB + (c % (NCT * 4)) * 16) = zr[k];
        }
    };
    if (u0 < u1) fetch(u0);
    for (int64_t u = u0; u < u1; ++u) {
        __builtin_amdgcn_wave_barrier();
        stage();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (u + 1 < u1) fetch(u + 1);                              // in flight under this unit's reads and MFMAs
        for (int k0 = 0; k0 < ow; k0 += 16) {
            s16x8 bv[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const unsigned char* pb = sZw + zoff + k0 * ZB + c * 64;
                const s16x4 lo = lds_read_tr(pb), hi = lds_read_tr(pb + 4 * ZB);
                bv[c] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                s16x8 av;
                const unsigned char* pa = sXw + abase[t] + (k0 + 8 * lh) * apitch;
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const short*>(pa + j * apitch);
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[t][c] = mfma16<T>(av, bv[c], acc[t][c]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every read of this unit done before it is overwritten
    }
    // the four waves' tiles -> LDS -> one atomic per element
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_w);                 // [4][NRT*NCT][32 x 32]
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[((wave * NRT + t) * NCT + c) * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[t][c][r];
    __syncthreads();
    for (int i = threadIdx.x; i < NRT * NCT * 1024; i += 256) {
        const int tc = i >> 10, e = i & 1023;
        const int t = tc / NCT, c = tc - t * NCT;
        const int rho = t * 32 + (e >> 5), co = c * 32 + (e & 31);
        if (rho < R && co < cout) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(w * NRT * NCT + tc) * 1024 + e];
            gv_dw_put(dw, blockIdx.x, (size_t)rho * cout + co, v);
        }
    }
}

// the stems this kernel takes: 3 input channels stored densely (x_ld == cin), stride 2, <= 64 output channels, rows that
// fit the per-lane staging registers; 0 on success, GV_E_UNSUPPORTED otherwise
template <typename T>
int wgrad_stem_rows(const gv_conv_desc* d, const unsigned short* x, const unsigned short* dz, int dz_ld, const GvDw& dw,
                    hipStream_t st) {
    const int R = d->kh * d->kw * d->cin;
    const int nrt = (R + 31) / 32, nct = (d->cout + 31) / 32;
    if (d->cin > 4 || d->x_ld != d->cin || d->stride != 2 || (nrt != 1 && nrt != 5) || nct > 2 || d->cout % 8 != 0 ||
        dz_ld % 8 != 0 || !gv_aligned16(x) || !gv_aligned16(dz) || (d->iw * d->cin * 2) % 16 != 0)
        return GV_E_UNSUPPORTED;
    const int xbytes = d->iw * d->cin * 2;
    // what a lane's staging registers hold (XL / ZL of the kernel), and 4-byte aligned rows behind the left padding
    const bool wide = d->ow * nct * 4 > 10 * 64;                   // the dZ row needs the 14-load form (nrt == 5, nct == 2 only)
    if ((d->kh * xbytes + 15) / 16 > (nrt == 1 ? 6 : 10) * 64 || d->ow * nct * 4 > (nrt == 5 && nct == 2 ? 14 : 10) * 64 ||
        xbytes % 4 != 0)
        return GV_E_UNSUPPORTED;
    // LDS row of the image: left padding + iw pixels + the columns the last (padded to 16) output pixels reach
    const int owp = (d->ow + 15) / 16 * 16;
    const int need_px = (owp - 1) * d->stride + d->kw;
    const int row_px = need_px > d->iw + d->pad_l ? need_px : d->iw + d->pad_l;
    const int xrow_b = (row_px * d->cin * 2 + 15) / 16 * 16 + 16;
    const int zrows = owp;
    const size_t wave_b = (size_t)d->kh * xrow_b + (size_t)zrows * nct * 64;
    size_t lds = 4 * wave_b;
    const size_t red_b = (size_t)4 * nrt * nct * 1024 * 4;
    if (red_b > lds) lds = red_b;
    if (lds > (wide ? 160 : 64) * 1024) return GV_E_UNSUPPORTED;
    if (wide && !GV_BIG_LDS_OK((conv_wgrad_stem_rows_lp<T, 5, 2, 14>), 160 * 1024)) return GV_E_UNSUPPORTED;
    const int64_t units = (int64_t)d->nb * d->oh;
    int64_t waves = 256 * (wide ? 1 : 3) * 4;                      // three workgroups per CU (the wide form: one)
    int64_t per = (units + waves - 1) / waves;
    if (per < 4) per = 4;
    int64_t nwg = ((units + per - 1) / per + 3) / 4;
    const size_t elems = (size_t)R * d->cout;
    if (gv_dw_clamp(dw, elems, nwg) < nwg) {                      // fewer workgroups (slices) than wanted: what the workspace holds
        nwg = gv_dw_clamp(dw, elems, nwg);
        per = (units + 4 * nwg - 1) / (4 * nwg);
        nwg = ((units + per - 1) / per + 3) / 4;
    }
    const GvDw sink = gv_dw_sink(dw, elems, nwg);
#define GV_WSTEM(NRT, NCT)                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_stem_rows_lp<T, NRT, NCT>), dim3((unsigned)nwg), dim3(256), lds, st, x, dz, dz_ld,     \
                       d->nb, d->ih, d->iw, d->cin, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, d->cout,   \
                       xrow_b, zrows, (int)per, sink)
    if (wide)
        hipLaunchKernelGGL((conv_wgrad_stem_rows_lp<T, 5, 2, 14>), dim3((unsigned)nwg), dim3(256), lds, st, x, dz, dz_ld, d->nb,
                           d->ih, d->iw, d->cin, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, d->cout, xrow_b, zrows,
                           (int)per, sink);
    else if (nrt == 1 && nct == 1) GV_WSTEM(1, 1);
    else if (nrt == 1) GV_WSTEM(1, 2);
    else if (nct == 1) GV_WSTEM(5, 1);
    else GV_WSTEM(5, 2);
#undef GV_WSTEM
    GV_LAUNCH_CHECK();
    return gv_dw_finish(dw, elems, nwg, st);
}

}  // namespace

extern "C" void gv_conv2d_wgrad_set_strip_taps(int n) { g_strip_ntw = n; }

namespace gvlp {

#define GV_LP_DISPATCH(dtype, CALL)                       \
    do {                                                  \
        if ((dtype) == GV_BF16) { using T = __bf16; CALL; }   \
        if ((dtype) == GV_F16) { using T = _Float16; CALL; }  \
        return GV_E_UNSUPPORTED;                          \
    } while (0)

int accumulate(int dtype, const void* src, int src_ld, void* dst, int dst_ld, int64_t npix, int c, hipStream_t st) {
    const unsigned short* s = (const unsigned short*)src;
    unsigned short* d = (unsigned short*)dst;
    const bool v = (c % 8 == 0) && vec8(s, src_ld) && vec8(d, dst_ld);
    GV_LP_DISPATCH(dtype, {
        if (v)
            hipLaunchKernelGGL((accumulate_lp<T, 8>), dim3(grid_for(npix * (c / 8))), dim3(256), 0, st, s, src_ld, d,
                               dst_ld, npix, c);
        else
            hipLaunchKernelGGL((accumulate_lp<T, 1>), dim3(grid_for(npix * c)), dim3(256), 0, st, s, src_ld, d, dst_ld,
                               npix, c);
        GV_LAUNCH_CHECK();
        return GV_OK;
    });
}

int grouped_sums(int dtype, int mode, const void* z, int z_ld, const void* dy, int dy_ld, const void* y, int y_ld,
                 const float* mean, const float* inv, const float* scale, const float* shift, int nb, int hw, int c,
                 int G, int splits, double* acc, hipStream_t st) {
    if (dtype == GV_F32) {                                       // the same streaming kernels on fp32 storage (4 per quad)
        if (!((c % 4 == 0) && vec4(z, z_ld) && vec4(dy, dy_ld) && vec4(y, y_ld) &&
              (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll))
            return GV_E_UNSUPPORTED;
        const dim3 grid((c + 63) / 64, splits, G);
        const float *zf = (const float*)z, *df = (const float*)dy, *yf = (const float*)y;
#define GV_SUMS32(MODE)                                                                                              \
        hipLaunchKernelGGL((grouped_sums_v8<float, float, MODE>), grid, dim3(256), 0, st, zf, z_ld, df, dy_ld, yf, y_ld,   \
                           mean, inv, scale, shift, nb, hw, c, G, acc)
        if (mode == 0) GV_SUMS32(0);
        else if (mode == 1) GV_SUMS32(1);
        else GV_SUMS32(2);
#undef GV_SUMS32
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    GV_LP_DISPATCH(dtype, return sums_t<T>(mode, (const unsigned short*)z, z_ld, (const unsigned short*)dy, dy_ld,
                                           (const unsigned short*)y, y_ld, mean, inv, scale, shift, nb, hw, c, G, splits,
                                           acc, st));
}

int scale_shift_act_grouped(int dtype, const void* x, int nb, int hw, int c, int x_ld, const float* scale,
                            const float* shift, int G, int relu, void* y, int y_ld, hipStream_t st) {
    if (dtype == GV_F32) {
        if (!((c % 4 == 0) && vec4(x, x_ld) && vec4(y, y_ld) && (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll))
            return GV_E_UNSUPPORTED;
        hipLaunchKernelGGL((bn_stream_v8<float, float, false>), dim3((c + 63) / 64, stream_splits(nb, hw, G, c), G), dim3(256), 0,
                           st, (const float*)x, x_ld, (const float*)nullptr, 0, (const float*)nullptr, 0, scale, shift,
                           (const float*)nullptr, (const double*)nullptr, (const int*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, 0, nb, hw, c, G, relu, (float*)y, y_ld, BnExtra{});
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    const unsigned short* xs = (const unsigned short*)x;
    unsigned short* ys = (unsigned short*)y;
    const bool v = (c % 8 == 0) && vec8(xs, x_ld) && vec8(ys, y_ld) && (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll;
    GV_LP_DISPATCH(dtype, {
        if (v)
            hipLaunchKernelGGL((bn_stream_v8<unsigned short, T, false>), dim3((c + 63) / 64, stream_splits(nb, hw, G, c), G), dim3(256), 0, st,
                               xs, x_ld, (const unsigned short*)nullptr, 0, (const unsigned short*)nullptr, 0, scale, shift,
                               (const float*)nullptr, (const double*)nullptr, (const int*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, 0, nb, hw, c, G, relu, ys, y_ld, BnExtra{});
        else
            hipLaunchKernelGGL((scale_shift_act_grouped_lp<T, 1>), dim3(grid_for((int64_t)nb * hw * c)), dim3(256), 0, st,
                               xs, nb, hw, c, x_ld, scale, shift, G, relu, ys, y_ld);
        GV_LAUNCH_CHECK();
        return GV_OK;
    });
}

// Fused finalize + apply (forward).  Returns GV_E_UNSUPPORTED when the shape needs the separate kernels.
int bn_finalize_apply_grouped(int dtype, const double* acc, const int* counts, const float* gamma, const float* beta,
                              float eps, const void* x, int nb, int hw, int c, int x_ld, int G, int relu, void* y,
                              int y_ld, float* mean, float* var, float* inv, float* scale, float* shift, hipStream_t st) {
    BnExtra ex{};
    ex.fin_acc = acc; ex.beta = beta; ex.eps = eps;
    ex.mean = mean; ex.var = var; ex.inv = inv; ex.scale_out = scale; ex.shift_out = shift;
    if (dtype == GV_F32) {
        if (!((c % 4 == 0) && vec4(x, x_ld) && vec4(y, y_ld) && (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll))
            return GV_E_UNSUPPORTED;
        hipLaunchKernelGGL((bn_stream_v8<float, float, false>), dim3((c + 63) / 64, stream_splits(nb, hw, G, c), G), dim3(256), 0,
                           st, (const float*)x, x_ld, (const float*)nullptr, 0, (const float*)nullptr, 0,
                           (const float*)nullptr, (const float*)nullptr, gamma, (const double*)nullptr, counts,
                           (const float*)nullptr, (const float*)nullptr, 0, nb, hw, c, G, relu, (float*)y, y_ld, ex);
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    const unsigned short* xs = (const unsigned short*)x;
    unsigned short* ys = (unsigned short*)y;
    const bool v = (c % 8 == 0) && vec8(xs, x_ld) && vec8(ys, y_ld) && (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll;
    if (!v) return GV_E_UNSUPPORTED;
    GV_LP_DISPATCH(dtype, {
        hipLaunchKernelGGL((bn_stream_v8<unsigned short, T, false>), dim3((c + 63) / 64, stream_splits(nb, hw, G, c), G), dim3(256), 0, st, xs,
                           x_ld, (const unsigned short*)nullptr, 0, (const unsigned short*)nullptr, 0, (const float*)nullptr,
                           (const float*)nullptr, gamma, (const double*)nullptr, counts, (const float*)nullptr,
                           (const float*)nullptr, 0, nb, hw, c, G, relu, ys, y_ld, ex);
        GV_LAUNCH_CHECK();
        return GV_OK;
    });
}

int bn_bwd_apply_grouped(int dtype, const void* dy, int dy_ld, const void* y, int y_ld, const void* z, int z_ld,
                         const float* mean, const float* inv, const float* gamma, const double* acc, const int* counts,
                         const float* scale, const float* shift, int accumulate, int nb, int hw, int c, int G, void* dz,
                         int dz_ld, float* dbeta, float* dgamma, bool* param_grads_done, hipStream_t st, int raw_z) {
    if (dtype == GV_F32) {
        *param_grads_done = false;
        if (raw_z) return GV_E_UNSUPPORTED;
        if (!((c % 4 == 0) && vec4(dy, dy_ld) && vec4(y, y_ld) && vec4(z, z_ld) && vec4(dz, dz_ld) &&
              (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll))
            return GV_E_UNSUPPORTED;
        BnExtra ex32{};
        ex32.dbeta = dbeta; ex32.dgamma = dgamma;
        *param_grads_done = true;
        hipLaunchKernelGGL((bn_stream_v8<float, float, true>), dim3((c + 63) / 64, stream_splits(nb, hw, G, c), G), dim3(256), 0,
                           st, (const float*)z, z_ld, (const float*)dy, dy_ld, (const float*)y, y_ld, mean, inv, gamma, acc,
                           counts, scale, shift, accumulate, nb, hw, c, G, 0, (float*)dz, dz_ld, ex32);
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    const unsigned short* a = (const unsigned short*)dy;
    const unsigned short* b = (const unsigned short*)y;
    const unsigned short* zz = (const unsigned short*)z;
    unsigned short* o = (unsigned short*)dz;
    const bool v = (c % 8 == 0) && vec8(a, dy_ld) && vec8(b, y_ld) && vec8(zz, z_ld) && vec8(o, dz_ld) &&
                   (int64_t)((nb + G - 1) / G) * hw < 0x7fffffffll;
    BnExtra ex{};
    ex.dbeta = dbeta; ex.dgamma = dgamma;
    *param_grads_done = v;
    if (raw_z && !v) return GV_E_UNSUPPORTED;                    // (sum g*z accumulators: the streaming kernel converts them)
    GV_LP_DISPATCH(dtype, {
        if (v)
            hipLaunchKernelGGL((bn_stream_v8<unsigned short, T, true>), dim3((c + 63) / 64, stream_splits(nb, hw, G, c), G), dim3(256), 0, st,
                               zz, z_ld, a, dy_ld, b, y_ld, mean, inv, gamma, acc, counts, scale, shift, accumulate, nb, hw, c,
                               G, raw_z ? 1 : 0, o, dz_ld, ex);
        else
            hipLaunchKernelGGL((bn_bwd_apply_grouped_lp<T, 1>), dim3(grid_for((int64_t)nb * hw * c)), dim3(256), 0, st, a,
                               dy_ld, b, y_ld, zz, z_ld, mean, inv, gamma, acc, counts, scale, shift, accumulate, nb, hw, c, G,
                               o, dz_ld);
        GV_LAUNCH_CHECK();
        return GV_OK;
    });
}

int pool2d_bwd(const gv_pool_desc* d, const void* x, const void* dy, int dy_ld, void* dx, int dx_ld, hipStream_t st) {
    const int store = (d->mode & GV_POOL_BWD_STORE) ? 1 : 0;
    const int mode = d->mode & ~GV_POOL_BWD_STORE;
    const int64_t npix = (int64_t)d->nb * d->ih * d->iw;
    const bool m3s2g = mode == GV_POOL_MAX && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_t == 0 &&
                       d->pad_l == 0 && d->oh == (d->ih - 3) / 2 + 1 && d->ow == (d->iw - 3) / 2 + 1;
    const int64_t nblk2 = (int64_t)d->nb * ((d->ih + 1) / 2) * ((d->iw + 1) / 2);
#define GV_POOL_BWD(E, T, VEC, XS, G_, O_)                                                                              \
    do {                                                                                                                \
        if (m3s2g && VEC > 1)                                                                                           \
            hipLaunchKernelGGL((maxpool3s2_bwd_lp<E, T, VEC>), dim3(grid_for(nblk2 * (d->c / VEC))), dim3(256), 0, st, XS, \
                               d->x_ld, G_, dy_ld, d->nb, d->ih, d->iw, d->c, d->oh, d->ow, store, O_, dx_ld);          \
        else                                                                                                            \
            hipLaunchKernelGGL((pool2d_bwd_lp<E, T, VEC>), dim3(grid_for(npix * (d->c / VEC))), dim3(256), 0, st, XS,      \
                               d->x_ld, G_, dy_ld, d->nb, d->ih, d->iw, d->c, d->kh, d->kw, d->stride, d->pad_t,        \
                               d->pad_l, d->oh, d->ow, mode, store, O_, dx_ld);                                         \
        GV_LAUNCH_CHECK();                                                                                              \
        return GV_OK;                                                                                                   \
    } while (0)
    if (d->dtype == GV_F32) {                                    // the same gather on fp32 storage (4 channels per thread)
        const float* xs = (const float*)x;
        const float* g = (const float*)dy;
        float* o = (float*)dx;
        const bool v4 = (d->c % 4 == 0) && (xs == nullptr || (gv_aligned16(xs) && d->x_ld % 4 == 0)) && gv_aligned16(g) &&
                        dy_ld % 4 == 0 && gv_aligned16(o) && dx_ld % 4 == 0;
        if (v4) GV_POOL_BWD(float, float, 4, xs, g, o);
        GV_POOL_BWD(float, float, 1, xs, g, o);
    }
    const unsigned short* xs = (const unsigned short*)x;
    const unsigned short* g = (const unsigned short*)dy;
    unsigned short* o = (unsigned short*)dx;
    const bool v = (d->c % 8 == 0) && vec8(xs, d->x_ld) && vec8(g, dy_ld) && vec8(o, dx_ld);
    GV_LP_DISPATCH(d->dtype, {
        if (v) GV_POOL_BWD(unsigned short, T, 8, xs, g, o);
        GV_POOL_BWD(unsigned short, T, 1, xs, g, o);
    });
#undef GV_POOL_BWD
}

// forward max pool + argmax bytes ([nb, oh, ow, c] dense), any storage type
int pool2d_fwd_argmax(const gv_pool_desc* d, const void* x, void* y, unsigned char* arg, hipStream_t st) {
    const int64_t opix = (int64_t)d->nb * d->oh * d->ow;
#define GV_AMAX_F(E, T, VEC, XS, YS)                                                                                        \
    do {                                                                                                                    \
        hipLaunchKernelGGL((maxpool_argmax_fwd<E, T, VEC>), dim3(grid_for(opix * (d->c / VEC))), dim3(256), 0, st, XS, d->x_ld, \
                           d->nb, d->ih, d->iw, d->c, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, YS, d->y_ld, \
                           arg);                                                                                            \
        GV_LAUNCH_CHECK();                                                                                                  \
        return GV_OK;                                                                                                       \
    } while (0)
    if (d->dtype == GV_F32) {
        const float* xs = (const float*)x;
        float* ys = (float*)y;
        const bool v4 = (d->c % 4 == 0) && gv_aligned16(xs) && d->x_ld % 4 == 0 && gv_aligned16(ys) && d->y_ld % 4 == 0 &&
                        (((uintptr_t)arg) & 3) == 0;
        if (v4) GV_AMAX_F(float, float, 4, xs, ys);
        GV_AMAX_F(float, float, 1, xs, ys);
    }
    const unsigned short* xs = (const unsigned short*)x;
    unsigned short* ys = (unsigned short*)y;
    const bool v = (d->c % 8 == 0) && vec8(xs, d->x_ld) && vec8(ys, d->y_ld) && (((uintptr_t)arg) & 7) == 0;
    GV_LP_DISPATCH(d->dtype, {
        if (v) GV_AMAX_F(unsigned short, T, 8, xs, ys);
        GV_AMAX_F(unsigned short, T, 1, xs, ys);
    });
#undef GV_AMAX_F
}

int pool2d_bwd_argmax(const gv_pool_desc* d, const unsigned char* arg, const void* dy, int dy_ld, void* dx, int dx_ld,
                      hipStream_t st, const void* bn_z, int bn_z_ld, int bn_G, const float* bn_A, const float* bn_B,
                      const float* bn_C, const float* bn_scale, const float* bn_shift) {
    BnTail bn{bn_z, bn_z_ld, bn_G, bn_A, bn_B, bn_C, bn_scale, bn_shift};
    const int store = (d->mode & GV_POOL_BWD_STORE) ? 1 : 0;
    const int64_t npix = (int64_t)d->nb * d->ih * d->iw;
    // 3x3 / 2 windows that start at even pixels: VALID, or TF's SAME on an even map (pads (0, 1): the last window is clipped —
    // ResNet-v2's pool1, nets/resnet_v2.py:181; the 2 x 2 block kernel only asks which windows exist)
    const bool same_h = d->ih % 2 == 0 && d->oh == d->ih / 2, same_w = d->iw % 2 == 0 && d->ow == d->iw / 2;
    const bool m3s2 = d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_t == 0 && d->pad_l == 0 &&
                      (d->oh == (d->ih - 3) / 2 + 1 || same_h) && (d->ow == (d->iw - 3) / 2 + 1 || same_w);
    if (bn_z && !m3s2) return GV_E_UNSUPPORTED;                  // the BatchNorm tail exists in the 3x3 / 2 kernel only
    if (bn_z && !(gv_aligned16(bn_A) && gv_aligned16(bn_B) && gv_aligned16(bn_C) && gv_aligned16(bn_scale) &&
                  gv_aligned16(bn_shift)))
        return GV_E_BADARG;                                          // (the coefficient tables are read as float4)
    if (m3s2 && (int64_t)((d->ih + 1) / 2) * ((d->iw + 1) / 2) * d->c > 0x7fffffff) return GV_E_UNSUPPORTED;
    const int64_t nblk2 = (int64_t)d->nb * ((d->ih + 1) / 2) * ((d->iw + 1) / 2);
#define GV_AMAX_B(E, T, VEC, G_, O_)                                                                                        \
    do {                                                                                                                    \
        if (m3s2)                                                                                                           \
            hipLaunchKernelGGL((maxpool3s2_argmax_bwd<E, T, VEC>), dim3(grid_for(nblk2 * (d->c / VEC))), dim3(256), 0, st, arg, \
                               G_, dy_ld, d->nb, d->ih, d->iw, d->c, d->oh, d->ow, store, O_, dx_ld, bn);                   \
        else                                                                                                                \
            hipLaunchKernelGGL((maxpool_argmax_bwd<E, T, VEC>), dim3(grid_for(npix * (d->c / VEC))), dim3(256), 0, st, arg, G_, \
                               dy_ld, d->nb, d->ih, d->iw, d->c, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, d->ow, \
                               store, O_, dx_ld);                                                                           \
        GV_LAUNCH_CHECK();                                                                                                  \
        return GV_OK;                                                                                                       \
    } while (0)
    if (d->dtype == GV_F32) {
        const float* g = (const float*)dy;
        float* o = (float*)dx;
        const bool v4 = (d->c % 4 == 0) && gv_aligned16(g) && dy_ld % 4 == 0 && gv_aligned16(o) && dx_ld % 4 == 0 &&
                        (((uintptr_t)arg) & 3) == 0;
        if (v4) GV_AMAX_B(float, float, 4, g, o);
        GV_AMAX_B(float, float, 1, g, o);
    }
    const unsigned short* g = (const unsigned short*)dy;
    unsigned short* o = (unsigned short*)dx;
    const bool v = (d->c % 8 == 0) && vec8(g, dy_ld) && vec8(o, dx_ld) && (((uintptr_t)arg) & 7) == 0;
    GV_LP_DISPATCH(d->dtype, {
        if (v) GV_AMAX_B(unsigned short, T, 8, g, o);
        GV_AMAX_B(unsigned short, T, 1, g, o);
    });
#undef GV_AMAX_B
}

int bn_bwd_coeffs_launch(const double* acc, const int* counts, const float* mean, const float* inv, const float* gamma,
                         int c, int G, int raw_z, float* A, float* B, float* Cc, float* dbeta, float* dgamma, hipStream_t st) {
    hipLaunchKernelGGL(bn_bwd_coeffs, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, st, acc, counts, mean, inv, gamma, c, G,
                       raw_z, A, B, Cc, dbeta, dgamma);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

int view_pool_fuse_bwd(int dtype, const void* F, const float* dS, int V, int N, int64_t E, int64_t vs, int64_t ss,
                       const int* scheme, int G, const float* weight, int mode, void* dF, hipStream_t st,
                       int64_t scheme_stride, int64_t weight_stride) {
    int64_t bx = (E + 255) / 256;
    if (bx > 1024) bx = 1024;
    const dim3 grid((unsigned)bx, (unsigned)N);
    GV_LP_DISPATCH(dtype, {
        hipLaunchKernelGGL(view_pool_fuse_bwd_lp<T>, grid, dim3(256), 0, st, (const unsigned short*)F, dS, V, N, E, vs, ss,
                           scheme, G, weight, mode, (unsigned short*)dF, scheme_stride, weight_stride);
        GV_LAUNCH_CHECK();
        return GV_OK;
    });
}

bool wgrad_mfma_ok(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld) {
    return d->cin % 8 == 0 && d->cout % 8 == 0 && d->x_ld % 8 == 0 && dz_ld % 8 == 0 && gv_aligned16(x) &&
           gv_aligned16(dz);
}

int conv_wgrad(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld, const GvDw& dw, hipStream_t st) {
    GV_LP_DISPATCH(d->dtype, return wgrad_t<T>(d, (const unsigned short*)x, (const unsigned short*)dz, dz_ld, dw, st));
}

// the 3-channel stems on the 16-bit MFMA (conv_wgrad_stem_rows_lp); GV_E_UNSUPPORTED: not such a layer
int conv_wgrad_stem(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld, const GvDw& dw, hipStream_t st) {
    GV_LP_DISPATCH(d->dtype, return wgrad_stem_rows<T>(d, (const unsigned short*)x, (const unsigned short*)dz, dz_ld, dw, st));
}

}  // namespace gvlp
