// pool.hip — HBM-bound NHWC window ops of the per-view backbone (fp32 path).
//   * max / average pooling (slim.max_pool2d / slim.avg_pool2d: nets/inception_v3.py:112,127,152,
//     219,355; nets/resnet_v2.py:181; `subsample`, nets/resnet_utils.py:64-67)
//   * per-channel scale/shift(+ReLU)  (stand-alone slim.batch_norm, nets/resnet_v2.py:75)
//   * global average pool              (tf.keras GlobalAveragePooling2D, nets/model.py:144,163)
// One thread per (output pixel, 4-channel group): 16-byte loads/stores, consecutive lanes on
// consecutive channels, so every wave instruction moves whole 128-byte lines.  Input and output
// carry an explicit pixel stride so a pool can read/write a channel slice of a concat buffer
// (Mixed_6a / Mixed_7a max-pool branches write straight into the block's output).
#include <math.h>

#include "gv_common.h"
#include "conv_x3_epi.h"
#include "lowp.h"

namespace {

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = f32x4; };
template <>
struct VecT<1> { using type = float; };

// XP3 / YP3 (VEC = 4 only): input / output in the three-plane layout of conv_x3_epi.h (ld in channels, multiples of 16)
template <int VEC, bool XP3 = false, bool YP3 = false>
__global__ __launch_bounds__(256) void pool2d_f32(const float* __restrict__ x, float* __restrict__ y,
                                                  int nb, int ih, int iw, int c, int x_ld, int kh,
                                                  int kw, int stride, int pad_t, int pad_l, int oh,
                                                  int ow, int y_ld, int mode) {
    using V = typename VecT<VEC>::type;
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
        V acc;
        if constexpr (VEC == 4) acc = mode == GV_POOL_MAX ? V{-INFINITY, -INFINITY, -INFINITY, -INFINITY} : V{0.f, 0.f, 0.f, 0.f};
        else acc = mode == GV_POOL_MAX ? -INFINITY : 0.f;
        int cnt = 0;
        for (int r = 0; r < kh; ++r) {
            const int iy = oy * stride + r - pad_t;
            if ((unsigned)iy >= (unsigned)ih) continue;
            for (int s = 0; s < kw; ++s) {
                const int ix = ox * stride + s - pad_l;
                if ((unsigned)ix >= (unsigned)iw) continue;
                V v;
                if constexpr (XP3) v = p3_load4(reinterpret_cast<const char*>(x), (size_t)(n * ih + iy) * iw + ix, x_ld, g * 4);
                else v = *reinterpret_cast<const V*>(x + ((size_t)(n * ih + iy) * iw + ix) * x_ld + g * VEC);
                if (mode == GV_POOL_MAX) {
                    if constexpr (VEC == 4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], v[e]);
                    } else {
                        acc = fmaxf(acc, v);
                    }
                } else {
                    acc += v;
                }
                ++cnt;
            }
        }
        if (mode != GV_POOL_MAX) {
            const float inv = (float)cnt;     // divisor = number of valid taps (TF SAME semantics)
            const bool relu = mode == GV_POOL_AVG_RELU;
            if constexpr (VEC == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = acc[e] / inv;
                    if (relu) acc[e] = fmaxf(acc[e], 0.f);
                }
            } else {
                acc = acc / inv;
                if (relu) acc = fmaxf(acc, 0.f);
            }
        }
        if constexpr (YP3) p3_store4(reinterpret_cast<char*>(y), (size_t)pix, y_ld, g * 4, acc);
        else *reinterpret_cast<V*>(y + (size_t)pix * y_ld + g * VEC) = acc;
    }
}

// 3x3 / stride 2 / VALID max pool (MaxPool_3a / 5a and the pooled branches of Mixed_6a / 7a, nets/inception_v3.py:112,127,
// 214,347) on fp32 storage, four vertically adjacent outputs of a 4-channel group per thread: 9 input rows instead of 12, and
// the rows two outputs share are not fetched again by a workgroup on another XCD (lowp.hip: maxpool3x3s2_rows_lp).
template <int RH>
__global__ __launch_bounds__(256) void maxpool3x3s2_rows_f32(const float* __restrict__ x, float* __restrict__ y, int nb,
                                                             int ih, int iw, int c, int x_ld, int oh, int ow, int y_ld) {
    const int cg = c >> 2;
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
        const int nr = min(RH, oh - oy0);
        const float* xp = x + ((size_t)(n * ih + 2 * oy0) * iw + 2 * ox) * x_ld + g * 4;
        float* yp = y + ((size_t)(n * oh + oy0) * ow + ox) * y_ld + g * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r <= 2 * RH; ++r) {
            if (r > 2 * nr) break;
            const float* rp = xp + (size_t)r * iw * x_ld;
            const f32x4 a = *reinterpret_cast<const f32x4*>(rp), b = *reinterpret_cast<const f32x4*>(rp + x_ld),
                        d_ = *reinterpret_cast<const f32x4*>(rp + 2 * x_ld);
            f32x4 hm;
#pragma unroll
            for (int e = 0; e < 4; ++e) hm[e] = fmaxf(fmaxf(a[e], b[e]), d_[e]);
            if (r == 0) {
                acc = hm;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], hm[e]);
                if ((r & 1) == 0) {
                    *reinterpret_cast<f32x4*>(yp + (size_t)(r / 2 - 1) * ow * y_ld) = acc;
                    acc = hm;
                }
            }
        }
    }
}

// 3x3 / stride 1 / SAME average pool (slim.avg_pool2d under the Inception arg-scope,
// nets/inception_v3.py:152,...): one thread produces 4 horizontally adjacent outputs of a 4-channel
// group from a 3x6 input patch (column sums are shared), i.e. 18 16-byte loads for 4 outputs
// instead of 36 — the generic kernel is L2-request bound on this op (2.4 TB/s).
// XP3: the input is in the three-plane layout of GV_CONV_Y_P3 (conv_x3_epi.h); the sum of the planes is the fp32 value.
template <bool XP3, bool YP3 = false>
__global__ __launch_bounds__(256) void avgpool3x3s1_row4_f32(const float* __restrict__ x,
                                                             float* __restrict__ y, int nb, int ih,
                                                             int iw, int c, int x_ld, int y_ld, int relu) {
    const int cg = c >> 2;
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
        f32x4 col[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) col[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int rows = 0;
#pragma unroll
        for (int r = -1; r <= 1; ++r) {
            const int iy = oy + r;
            if ((unsigned)iy >= (unsigned)ih) continue;
            ++rows;
            const float* rowp = x + ((size_t)(n * ih + iy) * iw) * x_ld + g * 4;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ix = ox0 - 1 + j;
                if ((unsigned)ix < (unsigned)iw) {
                    if constexpr (XP3)
                        col[j] += p3_load4(reinterpret_cast<const char*>(x), (size_t)(n * ih + iy) * iw + ix, x_ld, g * 4);
                    else
                        col[j] += *reinterpret_cast<const f32x4*>(rowp + (size_t)ix * x_ld);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = ox0 + j;
            if (ox >= iw) break;
            const int cols = 1 + (ox > 0 ? 1 : 0) + (ox + 1 < iw ? 1 : 0);
            const float inv = (float)(rows * cols);           // valid taps only (TF SAME semantics)
            f32x4 v = col[j] + col[j + 1] + col[j + 2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = v[e] / inv;
                if (relu) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (YP3) p3_store4(reinterpret_cast<char*>(y), (size_t)(n * ih + oy) * iw + ox, y_ld, g * 4, v);
            else *reinterpret_cast<f32x4*>(y + ((size_t)(n * ih + oy) * iw + ox) * y_ld + g * 4) = v;
        }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void scale_shift_act_f32(const float* __restrict__ x, int64_t npix,
                                                           int c, int x_ld,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           float* __restrict__ y, int y_ld) {
    using V = typename VecT<VEC>::type;
    const int cg = c / VEC;
    const int64_t total = npix * cg;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % cg);
        const int64_t pix = idx / cg;
        V v = *reinterpret_cast<const V*>(x + (size_t)pix * x_ld + g * VEC);
        const V sc = *reinterpret_cast<const V*>(scale + g * VEC);
        const V sh = *reinterpret_cast<const V*>(shift + g * VEC);
        v = v * sc + sh;
        if (relu) {
            if constexpr (VEC == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else {
                v = fmaxf(v, 0.f);
            }
        }
        *reinterpret_cast<V*>(y + (size_t)pix * y_ld + g * VEC) = v;
    }
}

// y[b][c] = mean_p x[b][p][c]; thread per (b, c): lanes walk consecutive channels.
__global__ __launch_bounds__(256) void global_avg_pool_f32(const float* __restrict__ x, int nb, int hw,
                                                           int c, int x_ld, float* __restrict__ y) {
    const int64_t total = (int64_t)nb * c;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % c);
    const int b = (int)(idx / c);
    const float* p = x + (size_t)b * hw * x_ld + ch;
    float s = 0.f;
    for (int i = 0; i < hw; ++i) s += p[(size_t)i * x_ld];
    y[idx] = s / (float)hw;
}

// The same pool on THREE-PLANE input, 8 channels and 2 x 4 outputs per thread: 16-byte plane loads (the form above reads a
// pixel's 4 channels as three 8-byte loads), rows oy and oy + 1 of the pair shared in registers, and workgroups mapped so
// that an XCD owns a contiguous range of them — the rows two outputs share are then in ITS L2 (round-robin dispatch put
// vertical neighbours on different XCDs: the pooled branches of c2 fetched 3.5 x their input from beyond L2, at an L2
// hit rate of 0.28).  Same additions in the same order as the form above: bitwise the same values.
template <bool YP3>
__global__ __launch_bounds__(256) void avgpool3x3s1_p3x8(const char* __restrict__ x, char* __restrict__ y, int nb, int ih,
                                                         int iw, int c, int x_ld, int y_ld, int relu) {
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
    const int oy0 = yg * 2, ox0 = xg * 4, ch = g * 8;
    float col[2][6][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) col[q][j][e] = 0.f;
    const size_t choff = (size_t)(ch & ~15) * 6 + (size_t)(ch & 15) * 2;
#pragma unroll
    for (int r = -1; r <= 2; ++r) {
        const int iy = oy0 + r;
        if ((unsigned)iy >= (unsigned)ih) continue;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int ix = ox0 - 1 + j;
            if ((unsigned)ix >= (unsigned)iw) continue;
            const char* src = x + ((size_t)(n * ih + iy) * iw + ix) * (size_t)x_ld * 6 + choff;
            p3_u32x4 pl[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) pl[p] = *reinterpret_cast<const p3_u32x4*>(src + p * 32);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int sh = (e & 1) * 16;
                const float a0 = __builtin_bit_cast(float, ((pl[0][e >> 1] >> sh) & 0xffffu) << 16);
                const float a1 = __builtin_bit_cast(float, ((pl[1][e >> 1] >> sh) & 0xffffu) << 16);
                const float a2 = __builtin_bit_cast(float, ((pl[2][e >> 1] >> sh) & 0xffffu) << 16);
                const float v = (a2 + a1) + a0;
                if (r <= 1) col[0][j][e] += v;
                if (r >= 0) col[1][j][e] += v;
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
            const size_t pix = (size_t)(n * ih + oy) * iw + ox;
            if constexpr (YP3) {
                p3_store8(y, pix, y_ld, ch, v);
            } else {
                float* yp = reinterpret_cast<float*>(y) + pix * (size_t)y_ld + ch;
                *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(yp + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        }
    }
}

inline unsigned grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    const int64_t cap = 256 * 16;            // 16 blocks per CU, grid-stride the rest
    return (unsigned)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace

// train_data.py:63,81-84,101: legacy bilinear resize + flips + brightness + x/255 - 0.5; thread per output pixel
__global__ __launch_bounds__(256) void preprocess_views_kernel(const unsigned char* __restrict__ src, int nimg, int h0,
                                                               int w0, int H, int W, const int* __restrict__ flip,
                                                               const float* __restrict__ delta,
                                                               float* __restrict__ dst) {
    const int64_t total = (int64_t)nimg * H * W;
    const float sy = (float)h0 / (float)H, sx = (float)w0 / (float)W;        // resize_images: scale = in / out
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int64_t t = idx / W;
        const int y = (int)(t % H);
        const int img = (int)(t / H);
        const int f = flip ? flip[img] : 0;
        const int xr = (f & 1) ? W - 1 - x : x;                               // a flip of the resized image
        const int yr = (f & 2) ? H - 1 - y : y;
        const float fy = __fmul_rn((float)yr, sy), fx = __fmul_rn((float)xr, sx);
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = min(y0 + 1, h0 - 1), x1 = min(x0 + 1, w0 - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const unsigned char* p = src + (size_t)img * h0 * w0 * 3;
        const float d = delta ? delta[img] : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float tl = p[((size_t)y0 * w0 + x0) * 3 + c], tr = p[((size_t)y0 * w0 + x1) * 3 + c];
            const float bl = p[((size_t)y1 * w0 + x0) * 3 + c], br = p[((size_t)y1 * w0 + x1) * 3 + c];
            const float top = __fadd_rn(tl, __fmul_rn(tr - tl, lx));
            const float bot = __fadd_rn(bl, __fmul_rn(br - bl, lx));
            float v = __fadd_rn(top, __fmul_rn(bot - top, ly));
            v = __fadd_rn(v, d);
            dst[idx * 3 + c] = __fadd_rn(__fmul_rn(v, 1.0f / 255.0f), -0.5f);
        }
    }
}

extern "C" int gv_preprocess_views(const uint8_t* src, int32_t nimg, int32_t h0, int32_t w0, int32_t height,
                                   int32_t width, const int32_t* flip, const float* delta, float* dst,
                                   void* stream) {
    if (!src || !dst || nimg <= 0 || h0 <= 0 || w0 <= 0 || height <= 0 || width <= 0) return GV_E_BADARG;
    const int64_t total = (int64_t)nimg * height * width;
    hipLaunchKernelGGL(preprocess_views_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, nimg, h0,
                       w0, height, width, flip, delta, dst);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_pool2d_fwd(const gv_pool_desc* d, const void* x, void* y, void* stream) {
    if (!d || !x || !y) return GV_E_BADARG;
    if (d->nb <= 0 || d->ih <= 0 || d->iw <= 0 || d->c <= 0 || d->kh <= 0 || d->kw <= 0 ||
        d->stride <= 0 || d->oh <= 0 || d->ow <= 0 || d->pad_t < 0 || d->pad_l < 0)
        return GV_E_BADARG;
    if (d->x_ld < d->c || d->y_ld < d->c) return GV_E_BADARG;
    if (!gv_pool_geometry_ok(d)) return GV_E_BADARG;             // (every dtype and storage form: checked before any branch)
    const bool xp3 = (d->mode & GV_POOL_X_P3) != 0, yp3 = (d->mode & GV_POOL_Y_P3) != 0;
    const int mode = d->mode & ~(GV_POOL_X_P3 | GV_POOL_Y_P3);
    if (mode != GV_POOL_MAX && mode != GV_POOL_AVG && mode != GV_POOL_AVG_RELU) return GV_E_BADARG;
    if (xp3 || yp3) {                            // three-plane input and / or output (fp32 values, GV_MATH_BF16X3 plans)
        if (d->dtype != GV_F32 || d->c % 4 != 0 || !gv_aligned16(x) || !gv_aligned16(y) ||
            (xp3 ? d->x_ld % 16 != 0 : d->x_ld % 4 != 0) || (yp3 ? d->y_ld % 16 != 0 : d->y_ld % 4 != 0))
            return GV_E_UNSUPPORTED;
        hipStream_t st = (hipStream_t)stream;
        const float* xf = (const float*)x;
        float* yf = (float*)y;
        if (mode != GV_POOL_MAX && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 && d->pad_l == 1 &&
            d->oh == d->ih && d->ow == d->iw) {
            const int64_t tot4 = (int64_t)d->nb * d->ih * ((d->iw + 3) / 4) * (d->c / 4);
            const int relu = mode == GV_POOL_AVG_RELU ? 1 : 0;
            const int64_t blk8 = ((int64_t)d->nb * ((d->ih + 1) / 2) * ((d->iw + 3) / 4) * (d->c / 8) + 255) / 256;
            if (xp3 && d->c % 8 == 0 && blk8 < 0x7fffffff) {     // three-plane input, whole 8-channel chunks
                if (yp3) hipLaunchKernelGGL((avgpool3x3s1_p3x8<true>), dim3((unsigned)blk8), dim3(256), 0, st, (const char*)x, (char*)y,
                                            d->nb, d->ih, d->iw, d->c, d->x_ld, d->y_ld, relu);
                else hipLaunchKernelGGL((avgpool3x3s1_p3x8<false>), dim3((unsigned)blk8), dim3(256), 0, st, (const char*)x, (char*)y,
                                        d->nb, d->ih, d->iw, d->c, d->x_ld, d->y_ld, relu);
                GV_LAUNCH_CHECK();
                return GV_OK;
            }
#define GV_AVG4(XP, YP) hipLaunchKernelGGL((avgpool3x3s1_row4_f32<XP, YP>), dim3(grid_for(tot4)), dim3(256), 0, st, xf, yf, \
                                           d->nb, d->ih, d->iw, d->c, d->x_ld, d->y_ld, relu)
            if (xp3 && yp3) GV_AVG4(true, true); else if (xp3) GV_AVG4(true, false); else GV_AVG4(false, true);
#undef GV_AVG4
        } else {
            const int64_t total = (int64_t)d->nb * d->oh * d->ow * (d->c / 4);
#define GV_POOL4(XP, YP) hipLaunchKernelGGL((pool2d_f32<4, XP, YP>), dim3(grid_for(total)), dim3(256), 0, st, xf, yf, d->nb, \
                                            d->ih, d->iw, d->c, d->x_ld, d->kh, d->kw, d->stride, d->pad_t, d->pad_l, d->oh, \
                                            d->ow, d->y_ld, mode)
            if (xp3 && yp3) GV_POOL4(true, true); else if (xp3) GV_POOL4(true, false); else GV_POOL4(false, true);
#undef GV_POOL4
        }
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    if (d->dtype != GV_F32) return gvlp::pool2d(d, x, y, (hipStream_t)stream);
    const bool vec = (d->c % 4 == 0) && (d->x_ld % 4 == 0) && (d->y_ld % 4 == 0) && gv_aligned16(x) &&
                     gv_aligned16(y);
    const int64_t total = (int64_t)d->nb * d->oh * d->ow * (vec ? d->c / 4 : d->c);
    hipStream_t st = (hipStream_t)stream;
    if (vec && d->mode != GV_POOL_MAX && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad_t == 1 &&
        d->pad_l == 1 && d->oh == d->ih && d->ow == d->iw) {
        const int64_t tot4 = (int64_t)d->nb * d->ih * ((d->iw + 3) / 4) * (d->c / 4);
        hipLaunchKernelGGL((avgpool3x3s1_row4_f32<false, false>), dim3(grid_for(tot4)), dim3(256), 0, st, (const float*)x,
                           (float*)y, d->nb, d->ih, d->iw, d->c, d->x_ld, d->y_ld, d->mode == GV_POOL_AVG_RELU ? 1 : 0);
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    if (vec && gvlp::pool_rows() && d->mode == GV_POOL_MAX && d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_t == 0 &&
        d->pad_l == 0 && d->oh == (d->ih - 3) / 2 + 1 && d->ow == (d->iw - 3) / 2 + 1) {
        constexpr int RH = 4;
        const int64_t tot = (int64_t)d->nb * ((d->oh + RH - 1) / RH) * d->ow * (d->c / 4);
        hipLaunchKernelGGL(maxpool3x3s2_rows_f32<RH>, dim3(grid_for(tot)), dim3(256), 0, st, (const float*)x, (float*)y,
                           d->nb, d->ih, d->iw, d->c, d->x_ld, d->oh, d->ow, d->y_ld);
        GV_LAUNCH_CHECK();
        return GV_OK;
    }
    if (vec)
        hipLaunchKernelGGL(pool2d_f32<4>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x,
                           (float*)y, d->nb, d->ih, d->iw, d->c, d->x_ld, d->kh, d->kw, d->stride,
                           d->pad_t, d->pad_l, d->oh, d->ow, d->y_ld, d->mode);
    else
        hipLaunchKernelGGL(pool2d_f32<1>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x,
                           (float*)y, d->nb, d->ih, d->iw, d->c, d->x_ld, d->kh, d->kw, d->stride,
                           d->pad_t, d->pad_l, d->oh, d->ow, d->y_ld, d->mode);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_scale_shift_act(const void* x, int64_t npix, int32_t c, int32_t x_ld,
                                  const float* scale, const float* shift, int32_t relu, void* y,
                                  int32_t y_ld, int32_t dtype, void* stream) {
    if (!x || !y || !scale || !shift || npix <= 0 || c <= 0 || x_ld < c || y_ld < c) return GV_E_BADARG;
    if (dtype != GV_F32)
        return gvlp::scale_shift_act(dtype, x, npix, c, x_ld, scale, shift, relu, y, y_ld, (hipStream_t)stream);
    const bool vec = (c % 4 == 0) && (x_ld % 4 == 0) && (y_ld % 4 == 0) && gv_aligned16(x) &&
                     gv_aligned16(y) && gv_aligned16(scale) && gv_aligned16(shift);
    const int64_t total = npix * (vec ? c / 4 : c);
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(scale_shift_act_f32<4>, dim3(grid_for(total)), dim3(256), 0, st,
                           (const float*)x, npix, c, x_ld, scale, shift, relu, (float*)y, y_ld);
    else
        hipLaunchKernelGGL(scale_shift_act_f32<1>, dim3(grid_for(total)), dim3(256), 0, st,
                           (const float*)x, npix, c, x_ld, scale, shift, relu, (float*)y, y_ld);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

extern "C" int gv_global_avg_pool(const void* x, int32_t nb, int32_t hw, int32_t c, int32_t x_ld,
                                  float* y, int32_t dtype, void* stream) {
    if (!x || !y || nb <= 0 || hw <= 0 || c <= 0 || x_ld < c) return GV_E_BADARG;
    if (dtype != GV_F32) return gvlp::global_avg_pool(dtype, x, nb, hw, c, x_ld, y, (hipStream_t)stream);
    const int64_t total = (int64_t)nb * c;
    hipLaunchKernelGGL(global_avg_pool_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)x, nb, hw, c, x_ld, y);
    GV_LAUNCH_CHECK();
    return GV_OK;
}

// ---- PNG row un-filtering (host code: tf.image.decode_png of train_data.py:55 after zlib inflate) -------------------
// raw: h rows of (1 filter byte + rowbytes data bytes); out: h x rowbytes.  Filters 0-4 of the PNG specification
// (None, Sub, Up, Average, Paeth); bpp = bytes per pixel.  Sequential by definition (each byte depends on its left
// neighbour), which is why it lives here and not in a Python loop.
extern "C" int gv_png_unfilter(const uint8_t* raw, int32_t h, int32_t rowbytes, int32_t bpp, uint8_t* out) {
    if (!raw || !out || h <= 0 || rowbytes <= 0 || bpp <= 0) return GV_E_BADARG;
    const uint8_t* prev = nullptr;
    for (int y = 0; y < h; ++y) {
        const uint8_t* line = raw + (size_t)y * (rowbytes + 1);
        uint8_t* cur = out + (size_t)y * rowbytes;
        const int ft = line[0];
        ++line;
        if (ft > 4) return GV_E_BADARG;
        for (int x = 0; x < rowbytes; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0;
            const int b = prev ? prev[x] : 0;
            const int c = (prev && x >= bpp) ? prev[x - bpp] : 0;
            int p = 0;
            if (ft == 1) p = a;
            else if (ft == 2) p = b;
            else if (ft == 3) p = (a + b) >> 1;
            else if (ft == 4) {
                const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                p = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[x] = (uint8_t)(line[x] + p);
        }
        prev = cur;
    }
    return GV_OK;
}
