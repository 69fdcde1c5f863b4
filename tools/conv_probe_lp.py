#!/usr/bin/env python3
"""Layer shapes of the 16-bit Inception step x EVERY tile configuration of the 16-bit kernels -> TFLOP/s, with the
ablation / A-B bits of csrc (gv_conv2d_set_debug), all on the same box in one run:
    0      the product kernels
    2      the FULL staged epilogue where the lean one would run                  (A/B of the round-3 lean epilogue)
    128    tap-major k-tile order (the general per-k-tile locate())               (A/B of the chunk-major fast loader)
    4      no epilogue at all (accumulators kept live; results not stored)        (upper bound of any epilogue work)
Columns: the register-staged tiles (conv_igemm_lp), the strip / halo configuration (0 where the layer is not its class),
then the LDS-DMA tiles (conv_dma) in the order of launch_dma_lp.
    python tools/conv_probe_lp.py [bf16|f16]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
ty = sys.argv[1] if len(sys.argv) > 1 else "bf16"
code, td = {"bf16": (_lib.GV_BF16, torch.bfloat16), "f16": (_lib.GV_F16, torch.float16)}[ty]


def probe(name, nb, h, w, cin, cout, kh, kw, dbgs=(0, 2, 128, 4), iters=20):
    x = torch.randn(nb, h, w, cin, device=dev).to(td)
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, code, 0) // 4
    wf = torch.randn(kh, kw, cin, cout, device=dev) * 0.05
    wp = torch.empty(n, device=dev)
    lib.gv_pack_filter_hwio(wf.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), code, 0, st)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, h, w, cout, device=dev, dtype=td)
    d = _lib.ConvDesc(nb, h, w, cin, cin, kh, kw, 1, kh // 2, kw // 2, h, w, cout, cout, 0, 0, 1, code, 0, 0, 0, 0)
    fl = 2.0 * nb * h * w * cout * kh * kw * cin
    ncfg = lib.gv_conv2d_num_tile_cfgs(-1)
    print("%s: M=%d N=%d K=%d (%dx%d, cin %d)" % (name, nb * h * w, cout, kh * kw * cin, kh, kw, cin))
    for dbg in dbgs:
        lib.gv_conv2d_set_debug(dbg)
        res = []
        for t in range(ncfg):
            lib.gv_conv2d_set_tile_override(t)
            ms = C.c_float(0)
            rc = lib.gv_conv2d_time(C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), iters,
                                    C.byref(ms), st)
            res.append(fl / ms.value / 1e9 if rc == 0 else 0.0)
        lib.gv_conv2d_set_tile_override(-1)
        lib.gv_conv2d_set_debug(0)
        k = lib.gv_conv2d_special_tile_cfg(-1)
        print("  dbg %3d: best %4.0f | staged: %s | strip: %3.0f | LDS-DMA: %s"
              % (dbg, max(res), " ".join("%3.0f" % r for r in res[:k]), res[k], " ".join("%3.0f" % r for r in res[k + 1:])), flush=True)


if __name__ == "__main__":
    print("LDS-DMA tiles: 128x128 256x128/8w 128x256/8w 256x256/8w 128x192 256x192/8w 128x64 64x128 128x96 256x128/4w 128x256/4w "
          "192x128 | two stages: 192x128 128x128 128x192 256x128/4w | 64-deep k-tiles: 128x192 256x192/8w 128x128 128x128/3st 256x128/8w 256x256/8w | 256x96/4w 256x64/4w 256x96/4w/k64")
    probe("Mixed_6c 7x1", 384, 12, 12, 160, 160, 7, 1)
    probe("Mixed_6e 1x7", 384, 12, 12, 192, 192, 1, 7)
    probe("Mixed_6 fused 1x1 siblings", 384, 12, 12, 768, 704, 1, 1)
    probe("Mixed_5 3x3", 384, 25, 25, 64, 96, 3, 3)
    probe("Mixed_5 3x3 96", 384, 25, 25, 96, 96, 3, 3)
    probe("Mixed_5 5x5", 384, 25, 25, 48, 64, 5, 5)
    probe("Conv2d_4a-shaped 3x3", 384, 52, 52, 80, 192, 3, 3)
