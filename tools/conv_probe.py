#!/usr/bin/env python3
"""Synthetic conv shapes x tile configs -> TFLOP/s table (separates steady-state loop efficiency
from tile-boundary / tail losses).  python tools/conv_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream


def probe(nb, hw, cin, cout, k, iters=5, tiles=None, dbg=0, math=0):
    x = torch.randn(nb, hw, hw, cin, device=dev)
    K = k * k * cin
    n = lib.gv_packed_filter_bytes(k, k, cin, cout, 0, math) // 4
    wf = torch.randn(k, k, cin, cout, device=dev) * 0.05
    w = torch.empty(n, device=dev)
    lib.gv_pack_filter_hwio(wf.data_ptr(), k, k, cin, cout, w.data_ptr(), 0, math, st)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, hw, hw, cout, device=dev)
    d = _lib.ConvDesc(nb, hw, hw, cin, cin, k, k, 1, k // 2, k // 2, hw, hw, cout, cout, 0, 0, 1, 0, 0, 0, math)
    M = nb * hw * hw
    fl = 2.0 * M * cout * K
    res = []
    lib.gv_conv2d_set_debug(dbg)
    for t in (tiles if tiles is not None else range(lib.gv_conv2d_num_tile_cfgs(math))):
        lib.gv_conv2d_set_tile_override(t)
        ms = C.c_float(0)
        rc = lib.gv_conv2d_time(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                y.data_ptr(), iters, C.byref(ms), st)
        res.append(fl / ms.value / 1e9 if rc == 0 else 0.0)
    lib.gv_conv2d_set_tile_override(-1)
    lib.gv_conv2d_set_debug(0)
    print("math%d M=%8d N=%4d K=%5d (k%d cin%d) dbg%d: %s" % (math, M, cout, K, k, cin, dbg, " ".join("%6.1f" % r for r in res)), flush=True)


if __name__ == "__main__":
    for math in (1, 2, 3, 0):
        for (nb, hw, cin, k) in [(54, 32, 128, 3), (256, 32, 32, 3), (256, 32, 128, 3), (256, 32, 512, 3)]:
            for cout in (192, 128):
                probe(nb, hw, cin, cout, k, math=math)
