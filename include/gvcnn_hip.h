/* gvcnn_hip.h — C ABI of the MI355X-native GVCNN hot path (libgvcnn_hip.so).
 *
 * Drop-in boundary for the path BASELINE.json:north_star names: the per-view
 * backbone + grouping module of ace19-dev/gvcnn-tf.  The reference has no
 * FFI/plugin layer (it is Python on TensorFlow 1.x), so each entry point cites
 * the reference CALL SITE whose arithmetic it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer would
 * add on the reference side.
 *
 * Conventions
 *  - every function returns 0 on success, a positive hipError_t value on a HIP
 *    failure, or a negative GV_E_* code; nothing throws;
 *  - every pointer is CALLER-OWNED DEVICE memory unless the name ends in
 *    `_host`; no hidden allocation, no hidden synchronisation;
 *  - `stream` is a hipStream_t passed as void*; launches are asynchronous and
 *    re-entrant across streams (safe to capture into a hipGraph);
 *  - activations are NHWC with an explicit pixel stride `ld` (in elements), so
 *    a tensor can be a channel slice of a wider (concat) buffer;
 *  - written for gfx950 only.
 */
#ifndef GVCNN_HIP_H
#define GVCNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GV_ABI_VERSION 1

/* error codes */
#define GV_OK 0
#define GV_E_BADARG (-1)      /* null pointer / non-positive size / inconsistent descriptor */
#define GV_E_UNSUPPORTED (-2) /* valid request the library does not implement (dtype, size cap) */
#define GV_E_ALIGN (-3)       /* pointer / stride not aligned as the vector path requires */
#define GV_E_PLAN (-4)        /* plan misuse (bad slot index, destroyed plan) */

/* element types (arithmetic type of the path; accumulation is always fp32) */
#define GV_F32 0
#define GV_BF16 1
#define GV_F16 2

/* gv_conv_desc.flags */
#define GV_CONV_RELU 1        /* ReLU after scale/shift(/residual)   */
#define GV_CONV_RELU2 2       /* ReLU on the second output           */
#define GV_CONV_SPLIT 4       /* y2 is not a second activation but the destination of output
                                 columns >= split_col (sibling convs fused into one GEMM) */
#define GV_CONV_X_F32 8       /* 16-bit dtype only: x is fp32 (the network input, nets/model.py:121) and is
                                 rounded to `dtype` by the loader — no separate cast pass over the images */
#define GV_CONV_X_P3 16       /* GV_F32 + GV_MATH_BF16X3 only: x holds every fp32 value as its three bf16 planes
                                 a = a0 + a1 + a2, laid out [pixel][channel/16][plane][16] (6 bytes per value; x_ld and
                                 the channel offset of a slice are multiples of 16 channels).  The producer split the
                                 value once; the LDS-DMA loader (csrc/conv_dma.hip) then only moves bytes */
#define GV_CONV_Y_P3 32       /* same dtype / math mode: y is written in that three-plane layout (y_ld, the slice offset
                                 and cout — with GV_CONV_SPLIT split_col — multiples of 16 channels); residual stays fp32 */
#define GV_CONV_Y2_P3 64      /* GV_CONV_SPLIT only: the second destination (columns >= split_col) is three-plane */
#define GV_CONV_MAXPOOL3S2 128 /* the convolution is followed by the reference's max_pool2d 3x3 / stride 2 / VALID
                                 (nets/inception_v3.py:113, Conv2d_2b_3x3 -> MaxPool_3a_3x3) and only the POOLED tensor is
                                 written: y is [nb, (oh-3)/2+1, (ow-3)/2+1, cout] with pixel stride y_ld; oh / ow stay the
                                 convolution's own output size.  Bit-identical to gv_conv2d_fwd + gv_pool2d.  Served for
                                 the halo-kernel class only (16-bit storage, 3x3 / stride 1, cin 32, cout 64, GV_CONV_RELU
                                 on every column, no residual / second output / BatchNorm sums, oh, ow >= 3):
                                 GV_E_UNSUPPORTED otherwise, and the caller issues the two launches.  Also served for the
                                 strip kernel of the 3-channel stems (GV_CONV_X_F32, 3x3 or 7x7 / stride 2, cout 64, any
                                 activation), and on GV_F32 storage under GV_MATH_BF16X3 for the same halo class (fp32
                                 input and output) on the maps where the kernel's 30-pixel strip form wastes fewer columns
                                 than its 16-pixel one: ceil(ow/16)*80 > ceil(ow/30)*128 — ow 17..30, 49..60, 65..90,
                                 97..120, and every ow >= 129 (csrc/conv_bf16s.hip, bf16s_halo_pool_ok) */
#define GV_CONV_MAXPOOL3S2_SAME 256 /* the same with TF's SAME padding on an even map (pads (0, 1): the last window is
                                 clipped; resnet_v2.py:181, conv1 -> pool1): y is [nb, oh/2, ow/2, cout].  Stem strip
                                 kernel only; odd oh / ow: GV_E_UNSUPPORTED */

#define GV_CONV_POOL_ACT2 512  /* with GV_CONV_MAXPOOL3S2[_SAME], y2 == NULL: the POOLED tensor is stored as
                                 act2(pool * scale2[c] + shift2[c]) (GV_CONV_RELU2: ReLU), computed from the pooled value as it
                                 would have been stored — ResNet-v2's first `preact` batch_norm + ReLU (nets/resnet_v2.py:75),
                                 whose only input is pool1 (:181): bit-identical to the fused launch + gv_scale_shift_act.  16-bit
                                 storage, the 3-channel stem kernel's class; anything else GV_E_UNSUPPORTED */

/* gv_conv_desc.math_mode: how an fp32 convolution is evaluated on the matrix cores */
#define GV_MATH_F32 0         /* v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain                         */
#define GV_MATH_BF16X3 1      /* operands split into 3 bf16 planes, 6 bf16 MFMAs per product block:
                                 fp32-level accuracy (dropped terms <= 2^-24 relative), ~2.7x the rate */
#define GV_MATH_BF16X2 2      /* 2 planes, 3 MFMAs: ~2^-16 relative                                     */
#define GV_MATH_BF16X1 3      /* plain bf16 products, fp32 accumulate                                   */

/* pooling modes */
#define GV_POOL_MAX 0         /* padding value -inf (slim.max_pool2d) */
#define GV_POOL_AVG 1         /* divisor = number of VALID taps (slim.avg_pool2d, SAME) */
#define GV_POOL_BWD_STORE 0x100 /* gv_pool2d_bwd only, OR-ed into mode, 16-bit dtypes: dx = ... instead of dx += ... */
#define GV_POOL_X_P3 0x200    /* OR-ed into mode, GV_F32 only: x is in the three-plane layout of GV_CONV_Y_P3 (x_ld and the
                                 slice offset multiples of 16 channels, c a multiple of 4) */
#define GV_POOL_Y_P3 0x400    /* OR-ed into mode, GV_F32 only: y is written in that layout */
#define GV_POOL_AVG_RELU 2    /* GV_POOL_AVG followed by ReLU.  avg_pool -> 1x1 conv -> BN -> ReLU
                                 (nets/inception_v3.py:152-154,...) is evaluated as relu(avgpool(BN(conv1x1(x)))):
                                 the 1x1 conv and the BN affine commute with the average (its weights sum to 1),
                                 so the pool runs on cout instead of cin channels and the 1x1 joins the block's
                                 other 1x1 convs in one GEMM */

/* view-pooling modes */
#define GV_VIEWPOOL_MAX 0     /* tf.reduce_max  — nets/model.py:72   */
#define GV_VIEWPOOL_MEAN 1    /* tf.reduce_mean — unit_test.py:30    */

/* image order of a flattened view batch */
#define GV_ORDER_SHAPE_MAJOR 0 /* image b = n*V + v  (the [N,V,H,W,3] input as stored, model.py:121) */
#define GV_ORDER_VIEW_MAJOR 1  /* image b = v*N + n  (after the transpose of model.py:128)            */

/* ------------------------------------------------------------------------
 * Convolution descriptor (implicit GEMM: M = nb*oh*ow, N = cout, K = kh*kw*cin)
 * ---------------------------------------------------------------------- */
typedef struct gv_conv_desc {
    int32_t nb, ih, iw, cin;   /* input  [nb, ih, iw, cin], pixel stride x_ld */
    int32_t x_ld;
    int32_t kh, kw;            /* filter window */
    int32_t stride;            /* same in h and w (all call sites) */
    int32_t pad_t, pad_l;      /* zero padding before; after-padding is implied by oh/ow */
    int32_t oh, ow, cout;      /* output [nb, oh, ow, cout], pixel stride y_ld */
    int32_t y_ld;
    int32_t res_ld;            /* pixel stride of the residual input (ignored when residual == NULL) */
    int32_t y2_ld;             /* pixel stride of the optional second output */
    int32_t flags;             /* GV_CONV_* */
    int32_t dtype;             /* GV_F32 | GV_BF16 | GV_F16: type of x, w_packed, residual, y, y2 (16-bit types:
                                  fp32 accumulation and epilogue, one rounding on store; math_mode ignored) */
    int32_t split_col;         /* GV_CONV_SPLIT: columns [0,split_col) -> y, [split_col,cout) -> y2 at
                                  column (c - split_col), pixel stride y2_ld; same scale/shift/act */
    int32_t tile_cfg;          /* 0 = library heuristic; k >= 1 = tile configuration k-1 (a speed choice only: the same
                                  products, summed in the same order within a kernel family — bitwise the same result —
                                  and in another k order across families (register-staged, LDS-DMA chunk-major, the
                                  wave-specialised kernels): fp32-rounding-level differences, <= 6e-6 of the tensor's
                                  largest value in the tests) */
    int32_t math_mode;         /* GV_MATH_*; w_packed must have been packed for the same mode */
    int32_t in_dilation;       /* 0/1: plain convolution.  2: the input tensor is read as if zero-dilated by 2
                                  (only even tap positions exist, at index/2) — the data gradient of a stride-2
                                  convolution is a stride-1 convolution of the dilated dZ (GV_MATH_BF16X* only) */
    int32_t relu_cols;         /* 0: GV_CONV_RELU applies to every output column.  n > 0: only to columns < n — the
                                  trailing columns leave the GEMM as BN(conv) without ReLU (an Inception pooled branch
                                  computed as relu(avgpool(BN(conv1x1(x)))), see GV_POOL_AVG_RELU) */
    int32_t y_step;            /* 0: output pixel (n, oy, ox) is pixel (n*oh + oy)*ow + ox of y.  2: it is pixel
                                  (2*oy + y_py, 2*ox + y_px) of an image of y_ih x y_iw pixels — ONE PARITY CLASS of the
                                  data gradient of a stride-2 convolution (16-bit storage): dX at rows of parity y_py
                                  and columns of parity y_px only receives the taps r = (y_py + pad_t) mod 2, +2, ...
                                  (likewise s), so each class is a small stride-1 convolution over the UN-dilated dZ
                                  with those taps; four launches cover dX with 1/4 of the multiply-adds the zero-dilated
                                  form (in_dilation = 2) spends.  The residual, if any, is read at the same pixels. */
    int32_t y_py, y_px, y_ih, y_iw;
} gv_conv_desc;

typedef struct gv_pool_desc {
    int32_t nb, ih, iw, c;
    int32_t x_ld;
    int32_t kh, kw, stride;
    int32_t pad_t, pad_l;
    int32_t oh, ow;
    int32_t y_ld;
    int32_t mode;              /* GV_POOL_* */
    int32_t dtype;
} gv_pool_desc;

int gv_abi_version(void);
/* Human-readable text for a code returned by any function below. */
const char* gv_error_string(int code);

/* ---- filters ------------------------------------------------------------
 * Packed filter layout consumed by gv_conv2d_fwd, GV_MATH_F32: [cout][Kpad], k = (r*kw+s)*cin + c,
 * Kpad = K rounded up to 32, zero filled; GV_MATH_BF16X*: [cout][K/16][plane][16 bf16] (the filter is
 * split into its bf16 planes once, here); dtype GV_BF16 / GV_F16: [cout][Kpad] 16-bit.  Source is TensorFlow's HWIO
 * [kh,kw,cin,cout] fp32 variable (slim `.../weights`). */
int64_t gv_packed_filter_bytes(int32_t kh, int32_t kw, int32_t cin, int32_t cout, int32_t dtype,
                               int32_t math_mode);
int gv_pack_filter_hwio(const float* w_hwio, int32_t kh, int32_t kw, int32_t cin, int32_t cout,
                        void* w_packed, int32_t dtype, int32_t math_mode, void* stream);
/* Many filters in ONE launch (16-bit dtypes; the training step re-packs every filter after each update).
 * jobs (device array): source HWIO variable, destination, geometry; flipped = 1 packs the DATA-GRADIENT filter
 * W'[r',s',co,ci] = W[kh-1-r', kw-1-s', ci, co] (gv_packed_filter_bytes(kh,kw,cout,cin) bytes) straight from the
 * forward variable (the forward image of a fused filter needs no extra field: member rows are contiguous, `out`
 * points at the member's first row).  Job j owns blocks [first_block, first_block + ceil(rows*Kpad/256)); block_job (device int32
 * [num_blocks]) maps a block to its job. */
typedef struct gv_pack_job {
    const float* w;
    void* out;
    int32_t kh, kw, cin, cout;
    int32_t flipped;
    int32_t first_block;
    int32_t k_off, k_total;     /* flipped only, k_total > 0: this filter is columns [k_off, k_off + cout) of a FUSED
                                   filter of k_total output channels (several 1x1 convolutions of one input run as one
                                   GEMM; its data-gradient image has rows of kh*kw*k_total, padding zeroed by the caller) */
    int32_t w_ld;               /* row stride of `w` in elements (0 = cout): a member filter stored as a column slice of
                                   the fused [kh,kw,cin,k_total] variable block */
    int32_t sub_step;           /* flipped only; 0 / 1: every tap.  2: ONE PARITY CLASS of a stride-2 data gradient
                                   (gv_conv_desc.y_step): kh x kw are the class's tap counts and tap (u, v) of the packed
                                   image is W[sub_r0 + 2*(kh-1-u), sub_s0 + 2*(kw-1-v)] of the src_kw-wide variable */
    int32_t sub_r0, sub_s0, src_kw, reserved2;
} gv_pack_job;
int gv_pack_filters_batched(const gv_pack_job* jobs_dev, int32_t num_jobs, const int32_t* block_job_dev,
                            int32_t num_blocks, int32_t dtype, void* stream);

/* ---- convolution ---------------------------------------------------------
 * y  = act( conv(x, w) * scale[c] + shift[c] (+ residual) )
 * y2 = act2( (that pre-activation value) * scale2[c] + shift2[c] )       (optional)
 * or, with GV_CONV_SPLIT, y receives columns [0, split_col) and y2 columns [split_col, cout): the
 * 1x1 convs of one Inception block that read the same input (nets/inception_v3.py:140-146,...)
 * run as one GEMM over their concatenated filters and the input is read once.
 *
 * Replaces slim.conv2d (+ slim.batch_norm + ReLU, folded into scale/shift) at
 * nets/inception_v3.py:97-405 (94 call sites), nets/resnet_v2.py:79-89,
 * nets/resnet_utils.py:94-105 (explicit pad + VALID), and `shortcut + residual`
 * at nets/resnet_v2.py:91 (residual), and the next unit's `preact`
 * slim.batch_norm at nets/resnet_v2.py:75 (second output).
 * scale/shift: fp32 [cout] (BN fold: scale = gamma*rsqrt(var+eps), shift = beta - mean*scale;
 * bias-only conv: scale = 1, shift = bias).  residual/y2/scale2/shift2 may be NULL. */
int gv_conv2d_fwd(const gv_conv_desc* d, const void* x, const void* w_packed,
                  const float* scale, const float* shift, const void* residual,
                  void* y, void* y2, const float* scale2, const float* shift2, void* stream);
/* The same convolution over x' = relu(x * xscale[ci] + xshift[ci]) (xscale/xshift: fp32 [cin], BN fold as above),
 * x' computed by the kernel's loader in fp32 and rounded once to the storage type: the `preact` slim.batch_norm of a
 * ResNet-v2 unit (nets/resnet_v2.py:75) applied by the convolution that consumes it (conv1, nets/resnet_v2.py:83), so
 * that the unit before it writes `shortcut + residual` (nets/resnet_v2.py:91) once and no pre-activation tensor exists.
 * 16-bit storage (GV_BF16 / GV_F16), 1x1 window, no padding, cin a multiple of 8 and <= 2048, 16-byte aligned pixels;
 * tile_cfg 0 or one of the register-staged tiles {1, 2, 7, 8, 9}; or gv_conv2d_special_tile_cfg(-1) + 1: the streaming form
 * (csrc/conv_chain.hip, TAIL) for cin = 4 * cout, cout 64 / 128 / 256, BatchNorm + ReLU on every column, one destination, no
 * residual — the conv1 of a bottleneck identity unit — bit for bit the register-staged result; anything else:
 * GV_E_UNSUPPORTED. */
int gv_conv2d_fwd_xpre(const gv_conv_desc* d, const void* x, const float* xscale, const float* xshift,
                       const void* w_packed, const float* scale, const float* shift, const void* residual,
                       void* y, void* y2, const float* scale2, const float* shift2, void* stream);

/* ---- bottleneck chain ----------------------------------------------------
 * The TAIL of one ResNet-v2 bottleneck unit and the HEAD of the next as ONE launch (csrc/conv_chain.hip):
 *   y = conv1x1(x, w3) * scale3[c] + shift3[c] + shortcut          nets/resnet_v2.py:87-91 of unit u (d -> 4d channels,
 *                                                                  normalizer_fn=None: scale3 = 1, shift3 = biases)
 *   z = relu( conv1x1( relu(y * pre_scale + pre_shift), w1 ) * scale1 + shift1 )
 *                                                                  nets/resnet_v2.py:75 (`preact`) and :83-84 (conv1 +
 *                                                                  BatchNorm + ReLU) of unit u+1 (4d -> d channels)
 * for two consecutive units of one block, where the next unit's shortcut is the identity (resnet_v2.py:76-77) and conv1 is
 * the pre-activation's only reader.  y is written once and never read back by conv1; the pre-activation is applied to the
 * ROUNDED y, as gv_conv2d_fwd_xpre's loader does — the results are those of gv_conv2d_fwd (y) followed by
 * gv_conv2d_fwd_xpre (z), bit for bit.  x [m, d], shortcut / y [m, 4d], z [m, d] with pixel strides *_ld (multiples of 8
 * elements, 16-byte aligned bases); w3_packed [4d][d], w1_packed [d][4d] as gv_pack_filter_hwio writes them for the dtype.
 * 16-bit storage, d = 64 or 128 (the HBM-bound blocks 1 and 2 of ResNet-v2-50): anything else GV_E_UNSUPPORTED and the
 * caller issues the two launches.
 * GV_CHAIN_PROJ (flags; d = 64): the unit's depth CHANGES and its shortcut is the 1x1 projection of its pre-activation
 * (nets/resnet_v2.py:79-81) — x is [m, 2d] = [the unit's conv2 output | the unit's pre-activation], w3_packed the
 * K-concatenated [4d][2d] filter [conv3 ; shortcut] (gv_pack_filter_hwio of the two HWIO filters concatenated along cin),
 * shift3 = conv3's biases + the shortcut's, `shortcut` NULL:  y = (conv3(x2) + shortcut(x0)) * scale3 + shift3  is ONE fp32
 * accumulation with one rounding, and no shortcut tensor is written or read.  Against the separate launches (which round
 * the shortcut to the storage type first) y differs by at most that one rounding. */
#define GV_CHAIN_PROJ 4096
typedef struct gv_chain_desc {
    int32_t m;                 /* rows = nb * oh * ow (all three tensors share the pixel grid) */
    int32_t d;                 /* bottleneck depth */
    int32_t x_ld, res_ld, y_ld, z_ld;
    int32_t dtype;             /* GV_BF16 | GV_F16 */
    int32_t flags;             /* GV_CONV_RELU2: ReLU on z; GV_CHAIN_PROJ */
    int32_t tile_cfg;          /* 0 (reserved) */
} gv_chain_desc;
int gv_bottleneck_chain_fwd(const gv_chain_desc* d, const void* x, const void* w3_packed, const float* scale3,
                            const float* shift3, const void* shortcut, void* y, const float* pre_scale,
                            const float* pre_shift, const void* w1_packed, const float* scale1, const float* shift1,
                            void* z, void* stream);

/* The same launch with the unit's conv2 in front (csrc/conv_chain.hip, FRONT): x is the unit's conv1 output [nb, ih, iw, d] and
 *   c2 = relu( conv3x3(x, w2, stride 1, SAME) * scale2 + shift2 )        nets/resnet_v2.py:85-86 (conv2 + BatchNorm + ReLU)
 * feeds gv_bottleneck_chain_fwd's y / z without ever being stored: one launch per bottleneck unit (from its conv2 to the
 * next unit's conv1), reading [m, d] + [m, 4d] and writing [m, 4d] + [m, d].  w2_packed [d][9*d] as gv_pack_filter_hwio
 * writes a 3x3 filter for the dtype.  conv2's k order here is tap-major; against the separate launches the results agree to
 * fp32 summation order (the other kernel families of gv_conv2d_fwd sum k chunk-major), everything behind c2's rounding is
 * the chain's arithmetic.  Same classes: 16-bit storage, d = 64 or 128, stride 1; else GV_E_UNSUPPORTED. */
typedef struct gv_unit_desc {
    int32_t nb, ih, iw;        /* the pixel grid of all four tensors */
    int32_t d;                 /* bottleneck depth */
    int32_t x_ld, res_ld, y_ld, z_ld;
    int32_t dtype;             /* GV_BF16 | GV_F16 */
    int32_t flags;             /* GV_CONV_RELU2: ReLU on z */
    int32_t tile_cfg;          /* 0 (reserved) */
} gv_unit_desc;
int gv_bottleneck_unit_fwd(const gv_unit_desc* d, const void* x, const void* w2_packed, const float* scale2,
                           const float* shift2, const void* w3_packed, const float* scale3, const float* shift3,
                           const void* shortcut, void* y, const float* pre_scale, const float* pre_shift,
                           const void* w1_packed, const float* scale1, const float* shift1, void* z, void* stream);

/* ---- pooling -------------------------------------------------------------
 * slim.max_pool2d / slim.avg_pool2d: nets/inception_v3.py:112,127,152,219,355,...;
 * nets/resnet_v2.py:181 (3x3/2 SAME, pad (0,1)); nets/resnet_utils.py:64-67
 * (`subsample` = 1x1 max-pool with stride). */
int gv_pool2d_fwd(const gv_pool_desc* d, const void* x, void* y, void* stream);

/* y[p,c] = act(x[p,c]*scale[c] + shift[c]) over npix pixels — slim.batch_norm (+ReLU)
 * used stand-alone: the `preact` of the first unit, nets/resnet_v2.py:75. */
int gv_scale_shift_act(const void* x, int64_t npix, int32_t c, int32_t x_ld, const float* scale,
                       const float* shift, int32_t relu, void* y, int32_t y_ld, int32_t dtype,
                       void* stream);

/* Global average pool over h*w: x [nb, hw, c] (pixel stride x_ld) -> y fp32 [nb, c].
 * tf.keras.layers.GlobalAveragePooling2D at nets/model.py:144,163. */
int gv_global_avg_pool(const void* x, int32_t nb, int32_t hw, int32_t c, int32_t x_ld, float* y,
                       int32_t dtype, void* stream);

/* ---- grouping module ------------------------------------------------------
 * Scorer, nets/model.py:144-145: r_img[b] = GAP(raw[b]) . kernel[v(b)] + bias[v(b)],
 * one Dense(1) per view.  raw [nb, hw, cr] (pixel stride raw_ld), kernel fp32 [V, cr],
 * bias fp32 [V]; v(b) from `order` (GV_ORDER_*), nb = N*V.  r_img fp32 [nb]. */
int gv_view_score_partial(const void* raw, int32_t nb, int32_t hw, int32_t cr, int32_t raw_ld,
                          const float* kernel, const float* bias, int32_t num_views,
                          int32_t order, float* r_img, int32_t dtype, void* stream);

/* nets/model.py:146-147: score[v] = sigmoid(log(|mean_n r_img[b(n,v)]|)), fixed summation
 * order (n ascending) so every rank of a multi-GPU job gets identical scores. */
int gv_view_score_finalize(const float* r_img, int32_t num_shapes, int32_t num_views,
                           int32_t order, float* scores, void* stream);

/* nets/model.py:16-41 (`group_scheme` + `group_weight`) on device:
 * gidx[v] = (int)(scores[v] * (float)num_bins)  — fp32 product, truncation (model.py:23 uses 10);
 * scheme [G,V] one-hot int32; weight[g] = 1 + #views in g (fp32).
 * status (device int32): 0 ok; 1 some gidx >= G or < 0 (the reference raises IndexError
 * there); 2 a score is NaN.  Outputs for in-range views are still written. */
int gv_group_assign(const float* scores, int32_t num_views, int32_t num_groups, int32_t num_bins,
                    int32_t* gidx, int32_t* scheme, float* weight, int32_t* status, void* stream);

/* nets/model.py:28-41 (`group_weight`) alone, for an arbitrary 0/1 scheme fed by the caller:
 * weight[g] = 1 + #{v : scheme[g,v] == 1}. */
int gv_group_weight(const int32_t* scheme, int32_t num_groups, int32_t num_views, float* weight,
                    void* stream);

/* nets/model.py:44-102 (`view_pooling` + `group_fusion`) fused, one pass over the views:
 *   D[g] = pool_{v: scheme[g,v] != 0} F[v]      (empty group -> `empty_fill`: 1 in model.py:63)
 *   S    = sum_g weight[g]*D[g] / sum_g weight[g]
 * F element (v, n, e) at F + v*view_stride + n*shape_stride + e, e < E = h*w*C.
 * D (nullable) [G, N, E], S (nullable) [N, E].  scheme int32 [G,V] and weight fp32 [G] are
 * device arrays (they never visit the host on the fused path). V <= 64, G <= 64. */
int gv_view_pool_fuse_fwd(const void* F, int32_t num_views, int32_t num_shapes, int64_t E,
                          int64_t view_stride, int64_t shape_stride, const int32_t* scheme,
                          int32_t num_groups, const float* weight, int32_t mode, float empty_fill,
                          void* D, void* S, int32_t dtype, void* stream);

/* ---- per-shape grouping (the paper's grouping module; NOT what nets/model.py computes) -------
 * nets/model.py:146 averages the scorer response over the batch, so one scheme serves all N shapes.  The
 * per-shape form scores, bins and fuses every shape on its own: nothing couples the shapes of a batch, hence a
 * shape-sharded multi-GPU job needs no exchange at all. */
#define GV_WEIGHT_COUNT 0       /* weight[g] = 1 + #members                      (nets/model.py:28-41)          */
#define GV_WEIGHT_MEAN_SCORE 1  /* weight[g] = mean score of the members, 0 for an empty group (score-derived,
                                   as in the paper's figure assets/grouping module.png)                          */
/* scores[b] = sigmoid(log(|r_img[b]|)) for every image b (no batch mean). */
int gv_view_score_per_shape(const float* r_img, int32_t nb, float* scores, void* stream);
/* scores [N,V] (image b = n*V + v) -> gidx [N,V], scheme [N,G,V] one-hot int32, weight [N,G]; binning exactly as
 * gv_group_assign; status is OR-ed over the shapes. */
int gv_group_assign_per_shape(const float* scores, int32_t num_shapes, int32_t num_views, int32_t num_groups,
                              int32_t num_bins, int32_t weight_mode, int32_t* gidx, int32_t* scheme,
                              float* weight, int32_t* status, void* stream);
/* gv_view_pool_fuse_fwd with one scheme [G,V] and one weight [G] PER SHAPE (scheme [N,G,V], weight [N,G]).
 * A shape whose weights sum to 0 gets S = 0. */
int gv_view_pool_fuse_fwd_per_shape(const void* F, int32_t num_views, int32_t num_shapes, int64_t E,
                                    int64_t view_stride, int64_t shape_stride, const int32_t* scheme,
                                    int32_t num_groups, const float* weight, int32_t mode, float empty_fill,
                                    void* D, void* S, int32_t dtype, void* stream);

/* tf.keras.layers.Dense at nets/model.py:164: y[n,:] = x[n,:] @ kernel[F,C] + bias (fp32). */
int gv_dense_fwd(const float* x, int32_t n, int32_t f, const float* kernel, const float* bias,
                 int32_t c, float* y, void* stream);

/* ---- input preprocessing (train_data.py:63,81-84,101; eval_data.py:72,109) -----------------------
 * One launch turns a batch of decoded 8-bit RGB views [nimg, h0, w0, 3] into the network input [nimg, H, W, 3] fp32:
 *   tf.image.resize (TF1 resize_images: bilinear, align_corners=False, src = dst * in/out — no half-pixel shift),
 *   random_flip_left_right / random_flip_up_down (bit 0 / bit 1 of flip[img]; NULL = none),
 *   random_brightness (delta[img] added on the 0..255 scale; NULL = none),
 *   x * (1/255) - 0.5.
 * The random decisions are made by the caller (host RNG), the arithmetic happens here. */
int gv_preprocess_views(const uint8_t* src, int32_t nimg, int32_t h0, int32_t w0, int32_t height, int32_t width,
                        const int32_t* flip, const float* delta, float* dst, void* stream);

/* PNG row un-filtering on the HOST (the sequential half of tf.image.decode_png, train_data.py:55 / eval_data.py:64,
 * after zlib inflate): raw = h rows of (filter byte + rowbytes bytes), out = h x rowbytes; filter types 0-4 of the PNG
 * specification, bpp bytes per pixel.  Both pointers are host memory; no stream. */
int gv_png_unfilter(const uint8_t* raw, int32_t h, int32_t rowbytes, int32_t bpp, uint8_t* out);

/* ---- evaluation metrics (eval.py:94-99) ---------------------------------------------------------
 * prediction[n] = argmax_c logits[n,c] (first maximum, like tf.argmax); confusion[label, prediction] += 1
 * (tf.math.confusion_matrix, int32 [C,C], ACCUMULATED so a whole evaluation run needs one buffer);
 * *correct += #{n : prediction[n] == labels[n]}.  labels int64 [N]; a label outside [0,C) is counted in neither. */
int gv_eval_metrics(const float* logits, const int64_t* labels, int32_t n, int32_t c, int64_t* prediction,
                    int32_t* confusion, int32_t* correct, void* stream);

/* ---- training step (train.py:145,166-187, utils/train_utils.py:217-259) -----------
 * fp32.  Gradient outputs ACCUMULATE (+=) into caller-zeroed buffers, because a tensor that feeds several
 * consumers (an Inception block input, a ResNet shortcut) sums their gradients.
 *
 * Data gradient of a convolution = gv_conv2d_fwd on dZ with the filter flipped and transposed
 * (W'[r',s',co,ci] = W[kh-1-r',kw-1-s',ci,co]), pad' = k-1-pad, stride 1, in_dilation = forward stride,
 * residual = dX (accumulate).  Filter gradient: gv_conv2d_wgrad. */

/* Batch statistics per (group, channel), group of image b = b % num_groups (one group per view: the
 * reference normalises each view's graph copy over its own N*h*w values).  counts[g] = pixels of group g.
 * Writes mean/var(biased)/inv = rsqrt(var+eps) and the folded scale = inv*gamma, shift = beta - mean*scale,
 * all [num_groups, c].  accum: workspace double [num_groups, c, 2].  gamma may be NULL (Inception). */
int gv_bn_stats_grouped(const float* z, int32_t nb, int32_t hw, int32_t c, int32_t z_ld, int32_t num_groups,
                        const int32_t* counts, const float* gamma, const float* beta, float eps,
                        double* accum, float* mean, float* var, float* inv, float* scale, float* shift,
                        void* stream);
/* The two halves of gv_bn_stats_grouped, for data-parallel jobs that cut the batch on SHAPE boundaries: a view's
 * images then live on several ranks, so the per-(view, channel) sums (accum: double [num_groups, c, 2] = sum, sum of
 * squares) are all-reduced between the halves and `counts` holds the GLOBAL pixel counts. */
int gv_bn_sums_grouped(const float* z, int32_t nb, int32_t hw, int32_t c, int32_t z_ld, int32_t num_groups,
                       double* accum, void* stream);
int gv_bn_finalize_grouped(const double* accum, int32_t c, int32_t num_groups, const int32_t* counts,
                           const float* gamma, const float* beta, float eps, float* mean, float* var, float* inv,
                           float* scale, float* shift, void* stream);
/* Moving-average update of slim.batch_norm in training (the UPDATE_OPS train.py:178-186 groups with the
 * optimizer step), one update per view graph copy, applied in view order:
 *   moving_mean = moving_mean*decay + mean[g]*(1-decay)
 *   moving_var  = moving_var *decay + var[g]*n/(n-1)*(1-decay)      (fused batch norm: unbiased estimate)
 * for g = 0..num_groups-1, n = counts[g].  decay: inception_utils.py:32 (0.9997), resnet_utils.py:199 (0.997). */
int gv_bn_update_moving(const float* mean, const float* var, const int32_t* counts, int32_t num_groups,
                        int32_t c, float decay, float* moving_mean, float* moving_var, void* stream);
/* gv_bn_update_moving for every BatchNorm layer of a plan in ONE launch: job j = one layer (device pointers as in
 * gv_bn_update_moving), owning blocks [first_block, first_block + ceil(c / 256)); block_job (device int32) maps a
 * block to its job. */
typedef struct gv_bn_moving_job {
    const float* mean;
    const float* var;
    const int32_t* counts;
    float* moving_mean;
    float* moving_var;
    int32_t c;
    int32_t first_block;
    int32_t ld;                 /* row stride of mean / var in elements (0 = c): statistics gathered from all ranks sit
                                   side by side in one [views, sum of 2c] message */
    int32_t reserved;
} gv_bn_moving_job;
int gv_bn_update_moving_batched(const gv_bn_moving_job* jobs_dev, int32_t num_jobs, const int32_t* block_job_dev,
                                int32_t num_blocks, int32_t num_groups, float decay, void* stream);
/* y = act(x*scale[g][c] + shift[g][c]), g = image % num_groups. */
int gv_scale_shift_act_grouped(const float* x, int32_t nb, int32_t hw, int32_t c, int32_t x_ld,
                               const float* scale, const float* shift, int32_t num_groups, int32_t relu,
                               float* y, int32_t y_ld, void* stream);
/* Backward of y = relu(BN_train(z)) (y == NULL: no ReLU): dz += gamma*inv*(g - mean_g(g) - zhat*mean_g(g*zhat)),
 * g = dy*[y>0]; dbeta[c] += sum g, dgamma[c] += sum g*zhat (either may be NULL). */
int gv_bn_relu_bwd_grouped(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld, const float* z,
                           int32_t z_ld, const float* mean, const float* inv, const float* gamma,
                           const int32_t* counts, int32_t nb, int32_t hw, int32_t c, int32_t num_groups,
                           double* accum, float* dz, int32_t dz_ld, float* dbeta, float* dgamma, void* stream);
/* The two halves of gv_bn_relu_bwd_grouped (sums of g and g*zhat -> all-reduce over the ranks -> apply with the
 * global counts).  After the all-reduce dbeta / dgamma are already the GLOBAL gradients on every rank. */
int gv_bn_relu_bwd_sums_grouped(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld, const float* z,
                                int32_t z_ld, const float* mean, const float* inv, int32_t nb, int32_t hw, int32_t c,
                                int32_t num_groups, double* accum, void* stream);
int gv_bn_relu_bwd_apply_grouped(const float* dy, int32_t dy_ld, const float* y, int32_t y_ld, const float* z,
                                 int32_t z_ld, const float* mean, const float* inv, const float* gamma,
                                 const int32_t* counts, int32_t nb, int32_t hw, int32_t c, int32_t num_groups,
                                 const double* accum, float* dz, int32_t dz_ld, float* dbeta, float* dgamma,
                                 void* stream);
/* x *= s (the 1/world factor of a loss averaged over a sharded batch). */
int gv_scale(float* x, int64_t n, float s, void* stream);
/* dst[p][c] += src[p][c]: gradient fan-in of `shortcut + residual` (nets/resnet_v2.py:91). */
int gv_accumulate(const float* src, int32_t src_ld, float* dst, int32_t dst_ld, int64_t npix, int32_t c,
                  void* stream);
/* dbias[c] += sum over pixels of dz (bias-only convolutions, nets/resnet_v2.py:79-89,178-180). */
int gv_bias_grad(const float* dz, int32_t dz_ld, int64_t npix, int32_t c, double* accum, float* dbias,
                 void* stream);
/* dW_hwio[r,s,ci,co] += sum_pixels x[shifted pixel, ci] * dz[pixel, co]  (descriptor of the FORWARD conv).
 * x and dz in d->dtype (GV_F32: exact fp32 MFMA; GV_BF16 / GV_F16: 16-bit MFMA, fp32 accumulation); dW is fp32. */
int gv_conv2d_wgrad(const gv_conv_desc* d, const void* x, const void* dz, int32_t dz_ld, float* dw_hwio,
                    void* stream);
/* The same gradient, bitwise reproducible (the reference's CPU gradients are: utils/train_utils.py:217-259; the plain
 * form combines its pixel slices with fp32 atomics, whose arrival order decides the last bits).  Every pixel slice
 * STORES its partial image of dW into `workspace` (caller-owned device memory, 16-byte aligned, workspace_bytes long;
 * contents are scratch) and one more launch adds the slices into dw_hwio in slice order.  The number of slices is
 * limited to the images the workspace holds (workspace_bytes / (4 * kh*kw*cin*cout)); with room for fewer than two the
 * launch runs un-split.  Same descriptor, same tile_cfg values, same result up to fp32 summation order. */
int gv_conv2d_wgrad_ws(const gv_conv_desc* d, const void* x, const void* dz, int32_t dz_ld, float* dw_hwio,
                       void* workspace, int64_t workspace_bytes, void* stream);
/* Pool backward (descriptor of the forward pool): max -> first maximum of each window (tf MaxPoolGrad),
 * avg -> dy / #valid taps.  x is the forward input (max only).  x, dy, dx in d->dtype. */
int gv_pool2d_bwd(const gv_pool_desc* d, const void* x, const void* dy, int32_t dy_ld, void* dx,
                  int32_t dx_ld, void* stream);
/* Max pool for the training step (slim.max_pool2d, nets/inception_v3.py:112,127 / nets/resnet_v2.py:181, under a
 * gradient tape): the forward also records, per output element, the row-major window tap of its FIRST maximum
 * (argmax: uint8 [nb, oh, ow, c], dense), and the backward routes dy by that record — same result as gv_pool2d_bwd
 * (tf MaxPoolGrad: first maximum) without re-reading the forward input.  d: the forward descriptor, mode GV_POOL_MAX
 * (| GV_POOL_BWD_STORE in the backward: dx = instead of dx +=), kh*kw <= 255; x, y, dy, dx in d->dtype. */
int gv_pool2d_fwd_argmax(const gv_pool_desc* d, const void* x, void* y, uint8_t* argmax, void* stream);
int gv_pool2d_bwd_argmax(const gv_pool_desc* d, const uint8_t* argmax, const void* dy, int32_t dy_ld, void* dx,
                         int32_t dx_ld, void* stream);
/* A max pool that directly follows relu(BatchNorm_train(z)) with a positive BatchNorm scale (slim's Inception arg scope:
 * no gamma — nets/inception_utils.py:36; Conv2d_2b -> MaxPool_3a, Conv2d_4a -> MaxPool_5a: nets/inception_v3.py:107-128)
 * may pool z ITSELF: BN + ReLU are monotone there, so max(relu(bn(z))) = relu(bn(max z)) with the same winner, bit for
 * bit.  Forward: gv_pool2d_fwd_argmax on z, then gv_bn_finalize_apply_grouped_t on the POOLED tensor with the sums and
 * counts of the whole z (a quarter of the elements to normalise, and the activation at the un-pooled size never exists).
 * Backward: the BatchNorm sums are sums over the pooled elements (gv_bn_relu_bwd_sums_grouped_t on (d pooled, pooled z):
 * only a window's winner carries a gradient), gv_bn_bwd_coeffs_t turns them into the per-(group, channel) coefficients of
 * dz = A*g + B*z + C (and adds dbeta / dgamma), and gv_pool2d_bwd_argmax_bn gathers every input pixel's gradient g from
 * the windows that elected it (as gv_pool2d_bwd_argmax), masks it with [z*scale + shift > 0] and stores dz — the gradient
 * of the activation is never materialised either.  3x3 / stride 2 / VALID windows, 16-bit storage. */
int gv_bn_bwd_coeffs_t(const double* accum, const int32_t* counts, const float* mean, const float* inv, const float* gamma,
                       int32_t c, int32_t num_groups, int32_t raw_z, float* coef_a, float* coef_b, float* coef_c,
                       float* dbeta, float* dgamma, void* stream);
int gv_pool2d_bwd_argmax_bn(const gv_pool_desc* d, const uint8_t* argmax, const void* dy, int32_t dy_ld, const void* z,
                            int32_t z_ld, int32_t num_groups, const float* coef_a, const float* coef_b, const float* coef_c,
                            const float* scale, const float* shift, void* dz, int32_t dz_ld, void* stream);
/* Backward of gv_view_pool_fuse_fwd: dF += ... (tf.reduce_max splits equally among ties). */
int gv_view_pool_fuse_bwd(const float* F, const float* dS, int32_t num_views, int32_t num_shapes, int64_t E,
                          int64_t view_stride, int64_t shape_stride, const int32_t* scheme,
                          int32_t num_groups, const float* weight, int32_t mode, float* dF, void* stream);
/* The same with one scheme [G,V] / weight [G] per shape (gv_view_pool_fuse_fwd_per_shape); the scores that produced
 * scheme and weight receive no gradient (they are constants of the backward pass, as in the batch form). */
int gv_view_pool_fuse_bwd_per_shape(const float* F, const float* dS, int32_t num_views, int32_t num_shapes,
                                    int64_t E, int64_t view_stride, int64_t shape_stride, const int32_t* scheme,
                                    int32_t num_groups, const float* weight, int32_t mode, float* dF, void* stream);
int gv_global_avg_pool_bwd(const float* dgap, int32_t nb, int32_t hw, int32_t c, float* dx, int32_t dx_ld,
                           void* stream);
/* Mean sparse-softmax cross-entropy (train.py:145): loss (device scalar) and dlogits = (softmax-onehot)/n. */
int gv_softmax_ce(const float* logits, const int64_t* labels, int32_t n, int32_t c, float* loss,
                  float* dlogits, void* stream);
/* Dense backward: dx = dy W^T (overwritten); dkernel += x^T dy; dbias += sum dy. */
int gv_dense_bwd(const float* x, const float* dy, const float* kernel, int32_t n, int32_t f, int32_t c,
                 float* dx, float* dkernel, float* dbias, void* stream);
/* tf.train.MomentumOptimizer(lr, mu) with the slim L2 term: m = mu*m + (g + wd*w); w -= lr*m. */
int gv_sgd_momentum(float* w, const float* g, float* m, int64_t n, float lr, float mu, float wd,
                    void* stream);

/* ---- the training step on 16-bit storage (configs[2]: bf16 forward + backward) --------------------------
 * Storage-typed forms of the functions above: activations and activation gradients (z, y, dy, dz, F, dF, src, dst)
 * are `dtype` elements (GV_BF16 / GV_F16; GV_F32 forwards to the fp32 function), arithmetic is fp32 with one
 * rounding per stored element, batch statistics accumulate in fp64, parameter gradients (dbeta, dgamma, dbias, dW),
 * dS and the optimizer state stay fp32.  gv_conv2d_fwd (forward and data gradient), gv_conv2d_wgrad,
 * gv_pool2d_fwd/_bwd, gv_global_avg_pool, gv_view_score_partial and gv_view_pool_fuse_fwd take the storage type
 * from their descriptor / dtype argument. */
/* OR-ed into the `dtype` of the two *_sums_grouped_t calls (16-bit dtypes): `accum` is already zero (a training engine
 * keeps one accumulator per layer and zeroes them all with one fill per step instead of one memset per call). */
#define GV_ACCUM_ZEROED 0x100
int gv_bn_sums_grouped_t(const void* z, int32_t nb, int32_t hw, int32_t c, int32_t z_ld, int32_t num_groups,
                         double* accum, int32_t dtype, void* stream);
int gv_scale_shift_act_grouped_t(const void* x, int32_t nb, int32_t hw, int32_t c, int32_t x_ld, const float* scale,
                                 const float* shift, int32_t num_groups, int32_t relu, void* y, int32_t y_ld,
                                 int32_t dtype, void* stream);
/* gv_bn_finalize_grouped + gv_scale_shift_act_grouped_t as one call (one launch on 16-bit storage: every thread
 * derives its channels' scale/shift from the fp64 sums with gv_bn_finalize_grouped's arithmetic, one thread per
 * (group, channel) stores mean/var/inv/scale/shift for the backward pass and the moving averages). */
int gv_bn_finalize_apply_grouped_t(const double* accum, const int32_t* counts, const float* gamma, const float* beta,
                                   float eps, const void* x, int32_t nb, int32_t hw, int32_t c, int32_t x_ld,
                                   int32_t num_groups, int32_t relu, void* y, int32_t y_ld, float* mean, float* var,
                                   float* inv, float* scale, float* shift, int32_t dtype, void* stream);
/* Backward of y = relu(BN_train(z)).  The ReLU mask [y > 0] is read from y, or — y == NULL with scale/shift given (the
 * folded forward coefficients of gv_bn_finalize_grouped) — recomputed as [z*scale + shift > 0], which is the same
 * mask (the forward pass rounded that very value) without reading y at all; y == NULL and scale == NULL: no ReLU.
 * accumulate = 0 stores dz instead of adding to it (z has one consumer: no zero-fill, no read of dz). */
int gv_bn_relu_bwd_sums_grouped_t(const void* dy, int32_t dy_ld, const void* y, int32_t y_ld, const void* z,
                                  int32_t z_ld, const float* mean, const float* inv, int32_t nb, int32_t hw,
                                  int32_t c, int32_t num_groups, double* accum, const float* scale,
                                  const float* shift, int32_t dtype, void* stream);
int gv_bn_relu_bwd_apply_grouped_t(const void* dy, int32_t dy_ld, const void* y, int32_t y_ld, const void* z,
                                   int32_t z_ld, const float* mean, const float* inv, const float* gamma,
                                   const int32_t* counts, int32_t nb, int32_t hw, int32_t c, int32_t num_groups,
                                   const double* accum, void* dz, int32_t dz_ld, float* dbeta, float* dgamma,
                                   const float* scale, const float* shift, int32_t accumulate, int32_t dtype,
                                   void* stream);
/* OR-ed into the `dtype` of gv_bn_relu_bwd_apply_grouped_t: accum[g][c][1] holds sum g*z (the raw BatchNorm input, as
 * gv_conv2d_fwd_bnstats / GV_BN_STATS_BWD produces it) instead of sum g*zhat; the call converts it first:
 * sum g*zhat = inv * (sum g*z - mean * sum g). */
#define GV_ACCUM_RAW_Z 0x200
/* ---- train-mode BatchNorm sums folded into the launch that produces the tensor -------------------------------------
 * slim.batch_norm(is_training=True) (nets/inception_utils.py:52-62, nets/resnet_utils.py:230) normalises every view's
 * graph copy over its own values (nets/model.py:129-141): statistics per (view, channel), view of image b = b % groups.
 * gv_conv2d_fwd_bnstats is gv_conv2d_fwd (16-bit storage, one destination, no activation) whose epilogue also adds the
 * sums of the values it stores, exactly as stored, into fp64 accumulators [groups][c1-c0][2]:
 *   GV_BN_STATS_FWD  a forward convolution writing z:           { sum z, sum z^2 }   = what gv_bn_sums_grouped_t adds
 *   GV_BN_STATS_BWD  a data-gradient launch writing the FINAL dy of a tensor (its last or only contributor):
 *                    { sum g, sum g*z }, g = dy * [z*scale + shift > 0]; z = the BatchNorm's input, scale / shift
 *                    [groups][c1-c0] = its folded forward coefficients (NULL: no ReLU) — what
 *                    gv_bn_relu_bwd_sums_grouped_t adds, with z in place of zhat (GV_ACCUM_RAW_Z).
 * The output's channels may belong to several BatchNorm layers (members of a fused sibling GEMM; the concat a block's
 * data gradient writes): one segment each; acc == NULL: no sums for those columns.  Accumulators must be zero on entry
 * of a pass.  The sums are bitwise reproducible run to run (fixed-point partials, exact fp64 additions).
 * GV_E_UNSUPPORTED: this tile configuration / geometry cannot fold them (the caller runs the plain convolution and the
 * separate sums pass instead); nothing has been written in that case. */
#define GV_BN_STATS_FWD 1
#define GV_BN_STATS_BWD 2
#define GV_BN_STATS_MAX_SEG 8
typedef struct gv_bn_stats_seg {
    int32_t c0, c1;            /* output columns [c0, c1), multiples of 8 */
    int32_t z_ld, reserved;    /* BWD: pixel stride of z */
    const void* z;             /* BWD: BatchNorm input (storage type), channel c0 of the segment at z[pixel * z_ld] */
    const float* scale;        /* BWD: [groups][c1-c0] */
    const float* shift;
    double* acc;               /* [groups][c1-c0][2] */
} gv_bn_stats_seg;
typedef struct gv_bn_stats {
    int32_t mode, groups, nseg, reserved;
    gv_bn_stats_seg seg[GV_BN_STATS_MAX_SEG];
} gv_bn_stats;
int gv_conv2d_fwd_bnstats(const gv_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                          const float* shift, const void* residual, void* y, const gv_bn_stats* stats, void* stream);
int gv_accumulate_t(const void* src, int32_t src_ld, void* dst, int32_t dst_ld, int64_t npix, int32_t c,
                    int32_t dtype, void* stream);
int gv_bias_grad_t(const void* dz, int32_t dz_ld, int64_t npix, int32_t c, double* accum, float* dbias,
                   int32_t dtype, void* stream);
/* gv_view_pool_fuse_bwd (per_shape = 0) / gv_view_pool_fuse_bwd_per_shape (per_shape = 1) with F, dF in `dtype`. */
int gv_view_pool_fuse_bwd_t(const void* F, const float* dS, int32_t num_views, int32_t num_shapes, int64_t E,
                            int64_t view_stride, int64_t shape_stride, const int32_t* scheme, int32_t num_groups,
                            const float* weight, int32_t mode, void* dF, int32_t per_shape, int32_t dtype,
                            void* stream);

/* ---- plan: the per-view backbone as one native launch sequence -------------
 * A plan is an ordered list of the ops above whose operands are (slot, element offset)
 * pairs into a caller-supplied table of device base pointers, so that one plan runs
 * on any set of buffers of the right size.  Building is host-only work; gv_plan_run
 * only launches (capturable).  Replaces the V unrolled copies of the backbone graph
 * that nets/model.py:129-141 builds, folded to one batch of N*V images. */
typedef struct gv_plan gv_plan;
int gv_plan_create(gv_plan** out);
void gv_plan_destroy(gv_plan* p);
int gv_plan_num_ops(const gv_plan* p);
int gv_plan_add_conv(gv_plan* p, const gv_conv_desc* d,
                     int32_t x_slot, int64_t x_off, int32_t w_slot, int64_t w_off,
                     int32_t ss_slot, int64_t scale_off, int64_t shift_off,
                     int32_t res_slot, int64_t res_off, int32_t y_slot, int64_t y_off,
                     int32_t y2_slot, int64_t y2_off, int64_t scale2_off, int64_t shift2_off);
/* Set gv_conv_desc.tile_cfg of conv op `op_index` (plan-time autotuning; speed only). */
int gv_plan_set_conv_tile(gv_plan* p, int32_t op_index, int32_t tile_cfg);
/* Conv op `op_index` runs as gv_conv2d_fwd_xpre: xscale / xshift at these fp32 offsets of the op's scale/shift slot. */
int gv_plan_set_conv_xpre(gv_plan* p, int32_t op_index, int64_t xscale_off, int64_t xshift_off);
/* gv_bottleneck_chain_fwd as a plan op.  Offsets into the weight slot are in `dtype` elements, into the scale/shift slot in
 * fp32 elements: conv3's filter / scale / shift, the pre-activation's scale / shift, conv1's filter / scale / shift.
 * GV_CHAIN_PROJ: res_slot = -1 (no shortcut operand), anything else GV_E_BADARG. */
int gv_plan_add_chain(gv_plan* p, const gv_chain_desc* d, int32_t x_slot, int64_t x_off, int32_t w_slot, int64_t w3_off,
                      int64_t w1_off, int32_t ss_slot, int64_t scale3_off, int64_t shift3_off, int64_t pre_scale_off,
                      int64_t pre_shift_off, int64_t scale1_off, int64_t shift1_off, int32_t res_slot, int64_t res_off,
                      int32_t y_slot, int64_t y_off, int32_t z_slot, int64_t z_off);
/* gv_bottleneck_unit_fwd as a plan op: gv_plan_add_chain's operands plus conv2's filter / scale / shift. */
int gv_plan_add_unit(gv_plan* p, const gv_unit_desc* d, int32_t x_slot, int64_t x_off, int32_t w_slot, int64_t w2_off,
                     int64_t w3_off, int64_t w1_off, int32_t ss_slot, int64_t scale2_off, int64_t shift2_off,
                     int64_t scale3_off, int64_t shift3_off, int64_t pre_scale_off, int64_t pre_shift_off,
                     int64_t scale1_off, int64_t shift1_off, int32_t res_slot, int64_t res_off, int32_t y_slot, int64_t y_off,
                     int32_t z_slot, int64_t z_off);
/* Branch-level concurrency: put op `op_index` on launch lane `lane` (0 = the caller's stream, 1..7 =
 * plan-owned streams) and name the EARLIER ops it must wait for (producers of its inputs, and ops
 * still using a buffer it overwrites).  A whole-plan run then forks the lanes off `stream` and joins
 * them back, so the independent branches of an Inception block (nets/inception_v3.py:139-155) overlap
 * and one kernel's tail is filled by another's workgroups.  Partial runs stay single-lane.  A plan
 * with lanes owns its streams/events: do not run it from two host threads at once. */
int gv_plan_set_schedule(gv_plan* p, int32_t op_index, int32_t lane, const int32_t* deps,
                         int32_t num_deps);
int gv_plan_add_pool(gv_plan* p, const gv_pool_desc* d, int32_t x_slot, int64_t x_off,
                     int32_t y_slot, int64_t y_off);
int gv_plan_add_scale_shift_act(gv_plan* p, int64_t npix, int32_t c, int32_t x_ld, int32_t y_ld,
                                int32_t relu, int32_t dtype, int32_t x_slot, int64_t x_off,
                                int32_t ss_slot, int64_t scale_off, int64_t shift_off,
                                int32_t y_slot, int64_t y_off);
/* buffers_host: host array of `num_slots` device base pointers. */
int gv_plan_run(const gv_plan* p, void* const* buffers_host, int32_t num_slots, void* stream);
/* Run ops [first, first+count) only (per-layer timing / profiling). */
int gv_plan_run_range(const gv_plan* p, int32_t first, int32_t count, void* const* buffers_host,
                      int32_t num_slots, void* stream);

/* ---- hipGraph capture ------------------------------------------------------
 * Small view batches are launch-bound (~90 launches for ~2 ms of work at 12 views): record everything enqueued on
 * `stream` between begin and end — gv_plan_run with its lane fork/join included, the grouping kernels, anything else
 * the caller launches there — into one executable graph and replay it with one call.  `stream` must not be the
 * legacy default stream.  The recorded launches keep the POINTERS they were issued with: replay on the same buffers. */
typedef struct gv_graph gv_graph;
int gv_capture_begin(void* stream);
int gv_capture_end(void* stream, gv_graph** out);
int gv_graph_launch(const gv_graph* g, void* stream);
void gv_graph_destroy(gv_graph* g);

/* ---- timing helper ---------------------------------------------------------
 * Average duration (ms) of `iters` back-to-back gv_conv2d_fwd launches measured with
 * hipEvents on `stream` (the stream the kernel is launched on); used by bench.py for the
 * roofline line.  Synchronises the stream. */
int gv_conv2d_time(const gv_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                   const float* shift, void* y, int32_t iters, float* ms_avg_host, void* stream);
/* Same for a whole plan (or a range of it). */
int gv_plan_time(const gv_plan* p, int32_t first, int32_t count, void* const* buffers_host,
                 int32_t num_slots, int32_t iters, float* ms_avg_host, void* stream);
/* Every op timed IN SEQUENCE (the number bench.py's `roofline` object is built from): `iters` whole passes of the plan
 * on `stream` in plan order (single launch lane), an event behind every op; ms_per_op_host [gv_plan_num_ops()] = average
 * duration of each op where it sits in the step, behind its real predecessor (a launch timed as a warm repeat of itself
 * finds its filters and part of its input in L2). */
int gv_plan_time_each(const gv_plan* p, void* const* buffers_host, int32_t num_slots, int32_t iters,
                      float* ms_per_op_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GVCNN_HIP_H */
