// Shared pieces of the two implicit-GEMM convolution kernels (conv_igemm.hip: fp32 MFMA;
// conv_bf16s.hip: fp32 data split into bf16 planes on the bf16 MFMA).
#pragma once
#include "gv_common.h"
#include "conv_stats.h"

namespace gvconv {

struct ConvArgs {
    const float* x;
    const void* w;              // packed filter (layout depends on the math mode)
    const float* scale;
    const float* shift;
    const float* res;
    float* y;
    float* y2;
    const float* scale2;
    const float* shift2;
    int nb, ih, iw, cin, x_ld;
    int kh, kw, stride, pad_t, pad_l;
    int oh, ow, cout, y_ld, res_ld, y2_ld;
    int M, K, Kpad, ktiles;
    int relu, relu2, split;     // split > 0: columns >= split go to y2 (same scale/shift/relu)
    int relu_limit;             // ReLU only for columns < relu_limit
    int tiles_n;
    int dil_shift;              // 0: plain; 1: input read as zero-dilated by 2 (stride-2 data gradient)
#ifdef GV_PHASE_TIMES
    unsigned long long* phase_buf = nullptr;   // profiling build: per-wave phase timestamps (conv_dma.hip)
#endif
    int dbg;                    // ablation bit (timing experiments only): 4 = no epilogue stores
    const void* zeros;          // conv_dma.hip: a zero page for the padding taps of the gather
    int y_p3, y2_p3;            // destination format: 0 = fp32, 1 = three bf16 planes (conv_x3_epi.h)
    int korder;                 // conv_dma.hip: 0 = k-tiles tap-major, 1 = channel-chunk-major (filter taps innermost)
    const float* xscale;        // conv_lp.hip (gv_conv2d_fwd_xpre): the input is read as relu(x*xscale[c] + xshift[c]);
    const float* xshift;        // nullptr = the input as stored
    ConvStats st;               // train-mode BatchNorm sums folded into the epilogue (conv_stats.h); mode 0 = off
    int y_step = 0;             // 2: output pixel (n, oy, ox) lands on pixel (2*oy + y_py, 2*ox + y_px) of a y_ih x y_iw
    int y_py = 0, y_px = 0;     //    image (one parity class of a stride-2 data gradient; 16-bit staged epilogue only)
    int y_ih = 0, y_iw = 0;
    GvFastDiv y_div_img = {0, -1, 1}, y_div_row = {0, -1, 1};   //    exact m / (oh*ow) and rem / ow
    int pool = 0;               // GV_CONV_MAXPOOL3S2 (1) / _SAME (2; conv3x3_halo_lp: 1 only): y is the 3x3 / 2 max pool of the
    int ph = 0, pw = 0;         //    output, ph x pw pixels per image
};

// May this launch take the lean 16-bit staged epilogue (conv_stats.h STAT_LEAN)?  dbg bit 2: always the full one (A/B).
inline bool lp_epilogue_lean_ok(const ConvArgs& a) {
    const bool one = a.y2 == nullptr && a.split == 0;
    // ... or the two destinations of a fused sibling GEMM, split on a chunk boundary (no second ACTIVATION: that is the
    // full epilogue's dual output)
    const bool two = a.y2 != nullptr && a.split > 0 && a.split % 8 == 0 && a.y2_ld % 8 == 0 && (((uintptr_t)a.y2) & 15) == 0 &&
                     !a.y2_p3;
    return !(a.dbg & 2) && (one || two) && a.cout % 8 == 0 && a.y_ld % 8 == 0 &&
           (((uintptr_t)a.y) & 15) == 0 && (a.res == nullptr || (a.res_ld % 8 == 0 && (((uintptr_t)a.res) & 15) == 0)) &&
           (!a.relu || a.relu_limit >= a.cout || a.relu_limit % 8 == 0);
}

// Epilogue straight from 32x32 MFMA accumulators (C/D layout is dtype independent on gfx950:
// col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)):
//   y = act(acc*scale[c] + shift[c] (+ residual)); y2 = second activation or split destination.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x16 (&acc)[TM][TN], int m0,
                                              int n0, int wm, int wn, int lane) {
    if (a.dbg & 4) {            // keep the accumulators live without storing the tile
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 1.2345e-30f) a.y[0] = t;
        return;
    }
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + col_l;
        if (col >= a.cout) continue;
        const float sc = a.scale[col], sh = a.shift[col];
        const bool to_second = a.split > 0 && col >= a.split;
        const bool dual = a.y2 != nullptr && a.split == 0;
        const bool relu = a.relu && col < a.relu_limit;
        float sc2 = 0.f, sh2 = 0.f;
        if (dual) { sc2 = a.scale2[col]; sh2 = a.shift2[col]; }
        float* ybase = to_second ? a.y2 + (col - a.split) : a.y + col;
        const int yld = to_second ? a.y2_ld : a.y_ld;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + (wm * TM + i) * 32 + row_h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + (r & 3) + 8 * (r >> 2);
                if (m >= a.M) continue;
                float v = acc[i][j][r] * sc + sh;
                if (a.res) v += a.res[(size_t)m * a.res_ld + col];
                if (dual) {
                    float v2 = v * sc2 + sh2;
                    if (a.relu2) v2 = fmaxf(v2, 0.f);
                    a.y2[(size_t)m * a.y2_ld + col] = v2;
                }
                if (relu) v = fmaxf(v, 0.f);
                ybase[(size_t)m * yld] = v;
            }
        }
    }
}

// conv_bf16s.hip
int bf16s_num_cfgs();
int bf16s_special_cfg();        // the halo-tiled / strip stem kernels
int bf16s_pick_tile(int planes, int M, int N, int K);
bool bf16s_halo_ok(int planes, const ConvArgs& a, bool generic);
bool bf16s_halo_pool_ok(int planes, const ConvArgs& a, bool generic);   // GV_CONV_MAXPOOL3S2 on fp32 storage
bool bf16s_stem_ok(int planes, const ConvArgs& a);
int bf16s_launch(int planes, int cfg, const ConvArgs& a, bool generic, hipStream_t st);
int bf16s_pack_filter(const float* w_hwio, int kh, int kw, int cin, int cout, int planes, void* out,
                      hipStream_t st);
int64_t bf16s_packed_bytes(int kh, int kw, int cin, int cout, int planes);


// conv_lp.hip (16-bit storage)
int lp_num_cfgs();
int lp_pick_tile(int M, int N, int K);
bool lp_xpre_cfg_ok(int cfg);   // register-staged tiles instantiated with the pre-activation-on-load loader
int lp_xpre_pick(int M, int N);
bool lp_halo_ok(const ConvArgs& a, bool generic);
bool lp_halo_pool_ok(const ConvArgs& a, bool generic);   // GV_CONV_MAXPOOL3S2
bool lp_stem_pool_ok(const ConvArgs& a, bool xf32);       // GV_CONV_MAXPOOL3S2 / GV_CONV_MAXPOOL3S2_SAME
bool lp_stem_ok(const ConvArgs& a, bool xf32);
int lp_launch(int dtype, int cfg, const ConvArgs& a, bool generic, bool xf32, hipStream_t st);
int lp_pack_filter(const float* w_hwio, int kh, int kw, int cin, int cout, int dtype, void* out, hipStream_t st);
int64_t lp_packed_bytes(int kh, int kw, int cin, int cout);
int lp_pack_filters_batched(const gv_pack_job* jobs_dev, const int* block_job_dev, int nblocks, int dtype, hipStream_t st);
int lp_special_cfg();           // the strip / halo kernels of the stem layers

// conv_dma.hip (LDS-DMA loader; extra tile configurations of the 16-bit storage path)
int dma_lp_num_cfgs();
bool dma_lp_ok(const ConvArgs& a, bool generic, bool xf32);
int dma_lp_launch(int dtype, int cfg, const ConvArgs& a, hipStream_t st);
const void* dma_zero_page();   // one zero page per device for padding taps / rows past the end
// conv_chain.hip: the conv1-behind-a-pre-activation class of gv_conv2d_fwd_xpre (1x1, cin = 4 * cout) as a streaming launch
bool chain_tail_ok(const ConvArgs& a);
int chain_tail_launch(int dtype, const ConvArgs& a, hipStream_t st);
// conv_ws.hip (wave-specialised kernel: loader waves + MFMA consumer waves; configurations follow the LDS-DMA tiles)
int ws_lp_num_cfgs();
int ws_lp_launch(int dtype, int cfg, const ConvArgs& a, hipStream_t st);
// fp32 values stored as three bf16 planes ("P3": [pixel][channel/16][plane][16]), GV_MATH_BF16X3
int dma_x3_num_cfgs();
bool dma_x3_ok(const ConvArgs& a);
int dma_x3_launch(int cfg, const ConvArgs& a, hipStream_t st);
// conv_ws_x3.hip (the wave-specialised strip kernel on three-plane input; configurations follow the LDS-DMA tiles)
int ws_x3_num_cfgs();
int ws_x3_launch(int cfg, const ConvArgs& a, hipStream_t st);
int wsg_x3_num_cfgs();          // its GEMM mode on plain fp32 input: configurations behind conv_bf16s.hip's (after the special one)
int wsg_x3_launch(int cfg, const ConvArgs& a, hipStream_t st);

}  // namespace gvconv
