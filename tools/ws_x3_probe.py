#!/usr/bin/env python3
"""Wave-specialised strip kernel on three-plane input (csrc/conv_ws_x3.hip) against every LDS-DMA tile of the same layer
shape (c2: 384 views, fp32 values as three bf16 planes), warm repeats on one box in one process; checks the two outputs
against each other.  Debug bits as an optional list: 0 product, 4 no epilogue, 16384 consumers alone, 32768 barriers but no
loads, 65536 loads but no barriers.
    python tools/ws_x3_probe.py [dbg ...]"""
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
NWS = 5
PEAK = 2500.0 / 6


def probe(name, nb, h, w, cin, cout, kh, kw, dbgs=(0,), iters=20):
    x = torch.randn(nb, h, w, cin, device=dev)
    xp = p3.to_p3(x)
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, _lib.GV_F32, X3) // 4
    wf = torch.randn(kh, kw, cin, cout, device=dev) * (1.0 / (kh * kw * cin) ** 0.5)
    wp = torch.empty(n, device=dev)
    _lib.check(lib.gv_pack_filter_hwio(wf.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), _lib.GV_F32, X3, st), "pack")
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, h, w, cout, device=dev)
    d = _lib.ConvDesc(nb, h, w, cin, cin, kh, kw, 1, kh // 2, kw // 2, h, w, cout, cout, 0, 0, 1 | _lib.GV_CONV_X_P3, _lib.GV_F32,
                      0, 0, X3, 0, 0)
    fl = 2.0 * nb * h * w * cout * kh * kw * cin
    ncfg = lib.gv_conv2d_num_tile_cfgs(-3)
    print("%-26s M=%7d N=%4d K=%5d (%dx%d, cin %d)" % (name, nb * h * w, cout, kh * kw * cin, kh, kw, cin))
    outs = {}
    for dbg in dbgs:
        lib.gv_conv2d_set_debug(dbg)
        res = []
        for t in range(ncfg):
            lib.gv_conv2d_set_tile_override(t)
            best = 0.0
            for _ in range(2):
                ms = C.c_float(0)
                rc = lib.gv_conv2d_time(C.byref(d), xp.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), iters,
                                        C.byref(ms), st)
                if rc == 0:
                    best = max(best, fl / ms.value / 1e9)
            res.append(best)
            if dbg == 0 and best > 0:
                torch.cuda.synchronize()
                outs[t] = y.clone()
        lib.gv_conv2d_set_tile_override(-1)
        lib.gv_conv2d_set_debug(0)
        old, ws = res[:ncfg - NWS], res[ncfg - NWS:]
        bo = max(range(len(old)), key=lambda i: old[i])
        line = "   dbg %5d: best dma %4.0f TF/s = %.3f (cfg %2d, %.1f us) | ws: %s | ws/dma %.2f" % (
            dbg, old[bo], old[bo] / PEAK, bo, fl / old[bo] / 1e6 if old[bo] else 0,
            " ".join("%d:%.0f=%.3f" % (i, r, r / PEAK) for i, r in enumerate(ws) if r > 0), max(ws) / old[bo] if old[bo] else 0)
        if dbg == 0 and outs:
            ref = outs[bo]
            errs = [float((outs[t] - ref).abs().max() / ref.abs().max()) for t in outs if t >= ncfg - NWS]
            line += " | max rel diff vs dma %.1e" % (max(errs) if errs else 0.0)
        print(line, flush=True)


if __name__ == "__main__":
    dbgs = tuple(int(v) for v in sys.argv[1:]) or (0,)
    nb = 384
    probe("Mixed_6b 1x7 128", nb, 17, 17, 128, 128, 1, 7, dbgs)
    probe("Mixed_6b 7x1 128->192", nb, 17, 17, 128, 192, 7, 1, dbgs)
    probe("Mixed_6c 7x1 160", nb, 17, 17, 160, 160, 7, 1, dbgs)
    probe("Mixed_6c 1x7 160->192", nb, 17, 17, 160, 192, 1, 7, dbgs)
    probe("Mixed_6e 1x7 192", nb, 17, 17, 192, 192, 1, 7, dbgs)
    probe("Mixed_6e 7x1 192", nb, 17, 17, 192, 192, 7, 1, dbgs)
    probe("Mixed_5 3x3 64->96", nb, 35, 35, 64, 96, 3, 3, dbgs)
    probe("Mixed_5 3x3 96->96", nb, 35, 35, 96, 96, 3, 3, dbgs)
    probe("Mixed_5 5x5 48->64", nb, 35, 35, 48, 64, 5, 5, dbgs)
    probe("Mixed_7 1x3 384", nb, 8, 8, 384, 384, 1, 3, dbgs)
    probe("Mixed_7 3x3 448->384", nb, 8, 8, 448, 384, 3, 3, dbgs)
