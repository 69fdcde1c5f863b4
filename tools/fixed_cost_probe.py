#!/usr/bin/env python3
"""Launch time vs K at fixed M, N (16-bit storage): separates a conv launch's fixed cost (launch, prologue, epilogue,
tail) from its marginal k-loop rate.   python tools/fixed_cost_probe.py [tiles...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
code, td = _lib.GV_BF16, torch.bfloat16
tiles = [int(t) for t in sys.argv[1:]] or [11, 18, 14, 22]


def probe(nb, hw, cin, cout, dbg=0, iters=20):
    x = torch.randn(nb, hw, hw, cin, device=dev).to(td)
    n = lib.gv_packed_filter_bytes(1, 1, cin, cout, code, 0) // 4
    wf = torch.randn(1, 1, cin, cout, device=dev) * 0.05
    w = torch.empty(n, device=dev)
    lib.gv_pack_filter_hwio(wf.data_ptr(), 1, 1, cin, cout, w.data_ptr(), code, 0, st)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, hw, hw, cout, device=dev, dtype=td)
    d = _lib.ConvDesc(nb, hw, hw, cin, cin, 1, 1, 1, 0, 0, hw, hw, cout, cout, 0, 0, 1, code, 0, 0, 0, 0)
    res = []
    lib.gv_conv2d_set_debug(dbg)
    for t in tiles:
        lib.gv_conv2d_set_tile_override(t)
        ms = C.c_float(0)
        rc = lib.gv_conv2d_time(C.byref(d), x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), iters,
                                C.byref(ms), st)
        res.append(ms.value * 1e3 if rc == 0 else float("nan"))
    lib.gv_conv2d_set_tile_override(-1)
    lib.gv_conv2d_set_debug(0)
    return res


for (nb, hw, cout) in [(384, 12, 192), (384, 25, 96), (384, 5, 384)]:
    for dbg in (0, 4):
        print("M=%d N=%d dbg=%d   us per launch for tiles %s" % (nb * hw * hw, cout, dbg, tiles))
        prev = None
        for cin in (32, 64, 192, 448, 896, 1344, 2688):
            r = probe(nb, hw, cin, cout, dbg)
            extra = ""
            if prev is not None:
                dk = cin - prev[0]
                fl = 2.0 * nb * hw * hw * cout * dk
                extra = "  marginal TF/s: " + " ".join("%6.0f" % (fl / max(a - b, 1e-3) / 1e6) for a, b in zip(r, prev[1]))
            print("  K=%5d: %s%s" % (cin, " ".join("%7.1f" % v for v in r), extra), flush=True)
            prev = (cin, r)
