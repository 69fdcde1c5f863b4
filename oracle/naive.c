/* Oracle (C restatement) — TEST INFRASTRUCTURE ONLY, see oracle/__init__.py.
 *
 * Direct-loop fp32 restatement of the TensorFlow ops the reference's hot path
 * is built from, written independently of oracle/backbone.py so the two CPU
 * implementations can arbitrate each other (TensorFlow itself
 * cannot run here; float parity is UNPINNED).
 *
 * Reference call sites restated (paths relative to /root/reference):
 *   conv      slim.conv2d          nets/inception_v3.py:97..., nets/resnet_v2.py:79-89,
 *                                  nets/resnet_utils.py:94-105 (explicit pad + VALID)
 *   bn        slim.batch_norm      nets/inception_utils.py:52-62, nets/resnet_utils.py:225-231
 *   max pool  slim.max_pool2d      nets/inception_v3.py:112,127,219,355, nets/resnet_v2.py:181
 *   avg pool  slim.avg_pool2d      nets/inception_v3.py:152,... (3x3/1 SAME, valid-tap divisor)
 *   grouping  nets/model.py:16-41 (group_scheme/group_weight),
 *             nets/model.py:44-102 (view_pooling/group_fusion)
 *
 * Layouts: activations NHWC, filters HWIO.  Accumulation in double, rounded
 * to float once.  Build: `make -C oracle` -> oracle/libgvref.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* y[n,oy,ox,co] = bias[co] + sum_{r,s,ci} x[n, oy*sh+r-pad_t, ox*sw+s-pad_l, ci] * w[r,s,ci,co] */
void gvref_conv2d_nhwc(const float* x, int nb, int ih, int iw, int cin,
                       const float* w, int kh, int kw, int cout,
                       int sh, int sw, int pad_t, int pad_l, int oh, int ow,
                       const float* bias, float* y)
{
    double* acc = (double*)malloc(sizeof(double) * (size_t)cout);
    for (int n = 0; n < nb; ++n)
        for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox) {
                for (int co = 0; co < cout; ++co) acc[co] = bias ? (double)bias[co] : 0.0;
                for (int r = 0; r < kh; ++r) {
                    int iy = oy * sh + r - pad_t;
                    if (iy < 0 || iy >= ih) continue;
                    for (int s = 0; s < kw; ++s) {
                        int ix = ox * sw + s - pad_l;
                        if (ix < 0 || ix >= iw) continue;
                        const float* xp = x + (((size_t)n * ih + iy) * iw + ix) * cin;
                        const float* wp = w + ((size_t)(r * kw + s) * cin) * cout;
                        for (int ci = 0; ci < cin; ++ci) {
                            double xv = xp[ci];
                            const float* wr = wp + (size_t)ci * cout;
                            for (int co = 0; co < cout; ++co) acc[co] += xv * (double)wr[co];
                        }
                    }
                }
                float* yp = y + (((size_t)n * oh + oy) * ow + ox) * cout;
                for (int co = 0; co < cout; ++co) yp[co] = (float)acc[co];
            }
    free(acc);
}

/* in place: x = (x-mean)*rsqrt(var+eps)*gamma + beta ; optional ReLU */
void gvref_bn_inference(float* x, int64_t npix, int c, const float* mean, const float* var,
                        const float* beta, const float* gamma, float eps, int relu)
{
    for (int64_t p = 0; p < npix; ++p)
        for (int ch = 0; ch < c; ++ch) {
            double inv = 1.0 / sqrt((double)var[ch] + (double)eps);
            if (gamma) inv *= (double)gamma[ch];
            double v = ((double)x[p * c + ch] - (double)mean[ch]) * inv + (double)beta[ch];
            if (relu && v < 0.0) v = 0.0;
            x[p * c + ch] = (float)v;
        }
}

/* mode 0: max (pad = -inf) ; mode 1: average over VALID taps only */
void gvref_pool2d_nhwc(const float* x, int nb, int ih, int iw, int c, int kh, int kw,
                       int sh, int sw, int pad_t, int pad_l, int oh, int ow, int mode, float* y)
{
    for (int n = 0; n < nb; ++n)
        for (int oy = 0; oy < oh; ++oy)
            for (int ox = 0; ox < ow; ++ox)
                for (int ch = 0; ch < c; ++ch) {
                    double m = -INFINITY, sum = 0.0;
                    int cnt = 0;
                    for (int r = 0; r < kh; ++r) {
                        int iy = oy * sh + r - pad_t;
                        if (iy < 0 || iy >= ih) continue;
                        for (int s = 0; s < kw; ++s) {
                            int ix = ox * sw + s - pad_l;
                            if (ix < 0 || ix >= iw) continue;
                            double v = x[(((size_t)n * ih + iy) * iw + ix) * c + ch];
                            if (v > m) m = v;
                            sum += v;
                            ++cnt;
                        }
                    }
                    y[(((size_t)n * oh + oy) * ow + ox) * c + ch] =
                        (float)(mode == 0 ? m : sum / (double)cnt);
                }
}

/* nets/model.py:16-41.  bin = (int)(score * (float)num_bins) with an fp32 product,
 * truncated toward zero.  Returns 0, or 1 when some bin >= G (the reference raises
 * IndexError there, model.py:23); scheme is [G][V] one-hot, weight[g] = 1 + count. */
int gvref_group_assign(const float* scores, int V, int G, int num_bins,
                       int32_t* gidx, int32_t* scheme, float* weight)
{
    int bad = 0;
    memset(scheme, 0, sizeof(int32_t) * (size_t)G * V);
    for (int v = 0; v < V; ++v) {
        volatile float prod = scores[v] * (float)num_bins;
        int b = (int)prod;
        gidx[v] = b;
        if (b >= G || b < 0) { bad = 1; continue; }
        scheme[b * V + v] = 1;
    }
    for (int g = 0; g < G; ++g) {
        int s = 1;
        for (int v = 0; v < V; ++v) if (scheme[g * V + v] == 1) s += 1;
        weight[g] = (float)s;
    }
    return bad;
}

/* nets/model.py:44-102.  F is [V][N][E] (E = h*w*C), scheme [G][V].
 * mode 0: max / mode 1: mean over the group's views; empty group -> `fill`.
 * D (nullable) receives the G group descriptors [G][N][E]; S (nullable) the
 * fused shape descriptor sum_g w_g D_g / sum_g w_g. */
void gvref_view_pool_fuse(const float* F, int V, int N, int64_t E, const int32_t* scheme, int G,
                          const float* weight, int mode, float fill, float* D, float* S)
{
    float wsum = 0.f;
    for (int g = 0; g < G; ++g) wsum += weight[g];
    for (int n = 0; n < N; ++n)
        for (int64_t e = 0; e < E; ++e) {
            float acc = 0.f;
            for (int g = 0; g < G; ++g) {
                float d = 0.f; int cnt = 0; float m = -INFINITY; float sum = 0.f;
                for (int v = 0; v < V; ++v) {
                    if (!scheme[g * V + v]) continue;
                    float x = F[((size_t)v * N + n) * E + e];
                    if (x > m) m = x;
                    sum += x; ++cnt;
                }
                d = cnt == 0 ? fill : (mode == 0 ? m : sum / (float)cnt);
                if (D) D[((size_t)g * N + n) * E + e] = d;
                acc += weight[g] * d;
            }
            if (S) S[(size_t)n * E + e] = acc / wsum;
        }
}
