// Internal entry points of lowp.hip: the 16-bit storage variants of the HBM-bound ops.  The extern "C"
// functions of pool.hip / grouping.hip validate their arguments and dispatch here on dtype.
#pragma once
#include "gv_common.h"

namespace gvlp {

int pool2d(const gv_pool_desc* d, const void* x, void* y, hipStream_t st);
int pool_rows();                                                  // A/B switch of the multi-row max pool (gv_pool2d_set_rows)
int scale_shift_act(int dtype, const void* x, int64_t npix, int c, int x_ld, const float* scale, const float* shift,
                    int relu, void* y, int y_ld, hipStream_t st);
int global_avg_pool(int dtype, const void* x, int nb, int hw, int c, int x_ld, float* y, hipStream_t st);
int view_score_partial(int dtype, const void* raw, int nb, int hw, int cr, int raw_ld, const float* kernel,
                       const float* bias, int num_views, int order, float* r_img, hipStream_t st);
int view_pool_fuse(int dtype, const void* F, int V, int N, int64_t E, int64_t vs, int64_t ss, const int* scheme,
                   int G, const float* weight, int mode, float fill, void* D, void* S, hipStream_t st,
                   int64_t scheme_stride = 0, int64_t weight_stride = 0);


// train_lp.hip: the training step on 16-bit storage
int accumulate(int dtype, const void* src, int src_ld, void* dst, int dst_ld, int64_t npix, int c, hipStream_t st);
int grouped_sums(int dtype, int mode, const void* z, int z_ld, const void* dy, int dy_ld, const void* y, int y_ld,
                 const float* mean, const float* inv, const float* scale, const float* shift, int nb, int hw, int c,
                 int G, int splits, double* acc, hipStream_t st);
int scale_shift_act_grouped(int dtype, const void* x, int nb, int hw, int c, int x_ld, const float* scale,
                            const float* shift, int G, int relu, void* y, int y_ld, hipStream_t st);
int bn_bwd_apply_grouped(int dtype, const void* dy, int dy_ld, const void* y, int y_ld, const void* z, int z_ld,
                         const float* mean, const float* inv, const float* gamma, const double* acc, const int* counts,
                         const float* scale, const float* shift, int accumulate, int nb, int hw, int c, int G, void* dz,
                         int dz_ld, float* dbeta, float* dgamma, bool* param_grads_done, hipStream_t st, int raw_z = 0);
int bn_finalize_apply_grouped(int dtype, const double* acc, const int* counts, const float* gamma, const float* beta,
                              float eps, const void* x, int nb, int hw, int c, int x_ld, int G, int relu, void* y,
                              int y_ld, float* mean, float* var, float* inv, float* scale, float* shift, hipStream_t st);
int pool2d_bwd(const gv_pool_desc* d, const void* x, const void* dy, int dy_ld, void* dx, int dx_ld, hipStream_t st);
int pool2d_fwd_argmax(const gv_pool_desc* d, const void* x, void* y, unsigned char* arg, hipStream_t st);
int pool2d_bwd_argmax(const gv_pool_desc* d, const unsigned char* arg, const void* dy, int dy_ld, void* dx, int dx_ld,
                      hipStream_t st, const void* bn_z = nullptr, int bn_z_ld = 0, int bn_G = 1, const float* bn_A = nullptr,
                      const float* bn_B = nullptr, const float* bn_C = nullptr, const float* bn_scale = nullptr,
                      const float* bn_shift = nullptr);
int bn_bwd_coeffs_launch(const double* acc, const int* counts, const float* mean, const float* inv, const float* gamma,
                         int c, int G, int raw_z, float* A, float* B, float* Cc, float* dbeta, float* dgamma, hipStream_t st);
int view_pool_fuse_bwd(int dtype, const void* F, const float* dS, int V, int N, int64_t E, int64_t vs, int64_t ss,
                       const int* scheme, int G, const float* weight, int mode, void* dF, hipStream_t st,
                       int64_t scheme_stride, int64_t weight_stride);
bool wgrad_mfma_ok(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld);
int conv_wgrad(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld, const GvDw& dw, hipStream_t st);
int conv_wgrad_stem(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld, const GvDw& dw, hipStream_t st);
// wgrad_dma.hip: the same filter gradient with LDS-DMA operand staging (tile_cfg 31 .. 30 + wgrad_dma_num_cfgs())
int wgrad_dma_num_cfgs();
int conv_wgrad_dma_launch(const gv_conv_desc* d, const void* x, const void* dz, int dz_ld, const GvDw& dw, int k, hipStream_t st);

}  // namespace gvlp
