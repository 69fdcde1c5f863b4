// Internal entry points of lowp.hip: the 16-bit storage variants of the HBM-bound ops.  The extern "C"
// functions of pool.hip / grouping.hip validate their arguments and dispatch here on dtype.
#pragma once
#include "gv_common.h"

namespace gvlp {

int pool2d(const gv_pool_desc* d, const void* x, void* y, hipStream_t st);
int scale_shift_act(int dtype, const void* x, int64_t npix, int c, int x_ld, const float* scale, const float* shift,
                    int relu, void* y, int y_ld, hipStream_t st);
int global_avg_pool(int dtype, const void* x, int nb, int hw, int c, int x_ld, float* y, hipStream_t st);
int view_score_partial(int dtype, const void* raw, int nb, int hw, int cr, int raw_ld, const float* kernel,
                       const float* bias, int num_views, int order, float* r_img, hipStream_t st);
int view_pool_fuse(int dtype, const void* F, int V, int N, int64_t E, int64_t vs, int64_t ss, const int* scheme,
                   int G, const float* weight, int mode, float fill, void* D, void* S, hipStream_t st,
                   int64_t scheme_stride = 0, int64_t weight_stride = 0);

}  // namespace gvlp
