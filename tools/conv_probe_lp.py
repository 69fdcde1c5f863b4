#!/usr/bin/env python3
"""Synthetic conv shapes x tile configs -> TFLOP/s of the 16-bit storage kernel (steady-state loop efficiency).
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


def probe(nb, hw, cin, cout, k, iters=5, tiles=None, dbg=0):
    x = torch.randn(nb, hw, hw, cin, device=dev).to(td)
    K = k * k * cin
    n = lib.gv_packed_filter_bytes(k, k, cin, cout, code, 0) // 4
    wf = torch.randn(k, k, cin, cout, device=dev) * 0.05
    w = torch.empty(n, device=dev)
    lib.gv_pack_filter_hwio(wf.data_ptr(), k, k, cin, cout, w.data_ptr(), code, 0, st)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, hw, hw, cout, device=dev, dtype=td)
    d = _lib.ConvDesc(nb, hw, hw, cin, cin, k, k, 1, k // 2, k // 2, hw, hw, cout, cout, 0, 0, 1, code, 0, 0, 0, 0)
    M = nb * hw * hw
    fl = 2.0 * M * cout * K
    res = []
    lib.gv_conv2d_set_debug(dbg)
    for t in (tiles if tiles is not None else range(11)):
        lib.gv_conv2d_set_tile_override(t)
        ms = C.c_float(0)
        rc = lib.gv_conv2d_time(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                y.data_ptr(), iters, C.byref(ms), st)
        res.append(fl / ms.value / 1e9 if rc == 0 else 0.0)
    lib.gv_conv2d_set_tile_override(-1)
    lib.gv_conv2d_set_debug(0)
    print("%s M=%8d N=%4d K=%5d (k%d cin%d) dbg%d: %s" % (ty, M, cout, K, k, cin, dbg, " ".join("%6.1f" % r for r in res)), flush=True)


if __name__ == "__main__":
    print("tiles: 128x128 128x64 64x64 128x96 64x128 128x32 256x128 128x256 256x64 128x192 64x192")
    for dbg in (0, 4):
        for (nb, hw, cin, k) in [(256, 32, 128, 3), (256, 32, 512, 1), (256, 32, 2048, 1), (54, 32, 192, 3)]:
            for cout in (256, 192, 128):
                probe(nb, hw, cin, cout, k, dbg=dbg)
