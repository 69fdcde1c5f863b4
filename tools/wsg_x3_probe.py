#!/usr/bin/env python3
"""GEMM mode of the wave-specialised three-plane kernel (csrc/conv_ws_x3.hip: 1x1 convolutions on plain fp32 input, the
loader waves split into planes) against every register-staged tile (csrc/conv_bf16s.hip) on the fused sibling GEMMs of c2
(384 views), warm repeats, one box, one process.   python tools/wsg_x3_probe.py [dbg ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
X3 = _lib.GV_MATH_BF16X3
PEAK = 2500.0 / 6


def probe(name, nb, hw, cin, cout, dbgs=(0,), iters=20):
    x = torch.randn(nb, hw, hw, cin, device=dev)
    n = lib.gv_packed_filter_bytes(1, 1, cin, cout, _lib.GV_F32, X3) // 4
    wf = torch.randn(1, 1, cin, cout, device=dev) * (1.0 / cin ** 0.5)
    wp = torch.empty(n, device=dev)
    _lib.check(lib.gv_pack_filter_hwio(wf.data_ptr(), 1, 1, cin, cout, wp.data_ptr(), _lib.GV_F32, X3, st), "pack")
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, hw, hw, cout, device=dev)
    d = _lib.ConvDesc(nb, hw, hw, cin, cin, 1, 1, 1, 0, 0, hw, hw, cout, cout, 0, 0, 1, _lib.GV_F32, 0, 0, X3, 0, 0)
    fl = 2.0 * nb * hw * hw * cout * cin
    ncfg = lib.gv_conv2d_num_tile_cfgs(X3)
    sp = lib.gv_conv2d_special_tile_cfg(X3)
    print("%-26s M=%7d N=%4d K=%5d" % (name, nb * hw * hw, cout, cin))
    for dbg in dbgs:
        res, outs = [], {}
        for t in range(ncfg):
            lib.gv_conv2d_set_debug(dbg if t > sp or dbg == 4 else 0)     # (the ablation bits are this kernel's own)
            lib.gv_conv2d_set_tile_override(t)
            best = 0.0
            for _ in range(2):
                ms = C.c_float(0)
                rc = lib.gv_conv2d_time(C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), iters,
                                        C.byref(ms), st)
                if rc == 0:
                    best = max(best, fl / ms.value / 1e9)
            res.append(best)
            if dbg == 0 and best > 0:
                torch.cuda.synchronize()
                outs[t] = y.clone()
        lib.gv_conv2d_set_tile_override(-1)
        lib.gv_conv2d_set_debug(0)
        old, ws = res[:sp], res[sp + 1:]
        bo = max(range(len(old)), key=lambda i: old[i])
        line = "   dbg %5d: best staged %4.0f TF/s = %.3f (cfg %2d, %.1f us) | ws gemm: %s | ws/staged %.2f" % (
            dbg, old[bo], old[bo] / PEAK, bo, fl / old[bo] / 1e6, " ".join("%d:%.0f=%.3f" % (i, r, r / PEAK) for i, r in enumerate(ws) if r > 0),
            max(ws) / old[bo])
        if dbg == 0:
            errs = [float((outs[t] - outs[bo]).abs().max() / outs[bo].abs().max()) for t in outs if t > sp]
            line += " | max rel diff %.1e" % (max(errs) if errs else -1.0)
        print(line, flush=True)


if __name__ == "__main__":
    dbgs = tuple(int(v) for v in sys.argv[1:]) or (0,)
    nb = 384
    probe("Mixed_5b siblings", nb, 25, 192, 208, dbgs)
    probe("Mixed_5c siblings", nb, 25, 256, 240, dbgs)
    probe("Mixed_5d siblings", nb, 25, 288, 240, dbgs)
    probe("Mixed_6b siblings", nb, 12, 768, 640, dbgs)
    probe("Mixed_6c siblings", nb, 12, 768, 704, dbgs)
    probe("Mixed_6e siblings", nb, 12, 768, 768, dbgs)
    probe("Mixed_7b siblings", nb, 5, 1280, 1344, dbgs)
    probe("Mixed_7c siblings", nb, 5, 2048, 1344, dbgs)
