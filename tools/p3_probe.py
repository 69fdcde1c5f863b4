#!/usr/bin/env python3
"""fp32-accurate convolution (GV_MATH_BF16X3): the register-staged kernel on fp32 input (split in the loader) against the
LDS-DMA kernel on three-plane (P3) input, same shapes; checks the two outputs against each other.
    python tools/p3_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib, p3  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
X3 = _lib.GV_MATH_BF16X3


def probe(nb, hw, cin, cout, kh, kw, iters=10):
    x = torch.randn(nb, hw, hw, cin, device=dev)
    xp = p3.to_p3(x)
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, _lib.GV_F32, X3) // 4
    wf = torch.randn(kh, kw, cin, cout, device=dev) * (1.0 / (kh * kw * cin) ** 0.5)
    w = torch.empty(n, device=dev)
    _lib.check(lib.gv_pack_filter_hwio(wf.data_ptr(), kh, kw, cin, cout, w.data_ptr(), _lib.GV_F32, X3, st), "pack")
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y0 = torch.empty(nb, hw, hw, cout, device=dev)
    y1 = torch.empty(nb, hw, hw, cout, device=dev)
    M = nb * hw * hw
    fl = 2.0 * M * cout * kh * kw * cin
    out = []
    for flags, xin, y, ncfg in ((1, x, y0, lib.gv_conv2d_special_tile_cfg(X3)), (1 | _lib.GV_CONV_X_P3, xp, y1, 8)):
        d = _lib.ConvDesc(nb, hw, hw, cin, cin, kh, kw, 1, kh // 2, kw // 2, hw, hw, cout, cout, 0, 0, flags, _lib.GV_F32,
                          0, 0, X3, 0, 0)
        best = (1e9, -1)
        times = []
        for t in range(ncfg):
            lib.gv_conv2d_set_tile_override(t)
            ms = C.c_float(0)
            rc = lib.gv_conv2d_time(C.byref(d), xin.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(),
                                    iters, C.byref(ms), st)
            times.append(ms.value if rc == 0 else float("nan"))
            if rc == 0 and ms.value < best[0]:
                best = (ms.value, t)
        lib.gv_conv2d_set_tile_override(best[1])
        _lib.check(lib.gv_conv2d_fwd(C.byref(d), xin.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None,
                                     y.data_ptr(), None, None, None, st), "fwd")
        lib.gv_conv2d_set_tile_override(-1)
        out.append((best, times))
    torch.cuda.synchronize()
    err = float((y0 - y1).abs().max() / y0.abs().max())
    (b0, t0), (b1, t1) = out
    print("M=%7d N=%4d K=%5d (%dx%d cin %d): staged %.4f ms (t%d, %.0f TF/s) | dma-p3 %.4f ms (t%d, %.0f TF/s) x%.2f  rel diff %.1e"
          % (M, cout, kh * kw * cin, kh, kw, cin, b0[0], b0[1], fl / b0[0] / 1e9, b1[0], b1[1], fl / b1[0] / 1e9,
             b0[0] / b1[0], err), flush=True)
    print("      dma tiles ms:", " ".join("%.4f" % t for t in t1))


if __name__ == "__main__":
    probe(384, 52, 80, 192, 3, 3)          # Conv2d_4a
    probe(384, 12, 192, 192, 1, 7)         # Mixed_6e 1x7
    probe(384, 12, 160, 160, 7, 1)         # Mixed_6c 7x1
    probe(384, 25, 48, 64, 5, 5)           # Mixed_5 5x5
    probe(384, 25, 96, 96, 3, 3)           # Mixed_5 3x3
    probe(384, 5, 384, 384, 3, 1)          # Mixed_7 3x1
    probe(384, 5, 448, 384, 3, 3)          # Mixed_7 3x3
    probe(96, 109, 32, 64, 3, 3)           # Conv2d_2b (quarter batch)
