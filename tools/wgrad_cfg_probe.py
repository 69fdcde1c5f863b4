#!/usr/bin/env python3
"""Per-layer time of the 16-bit filter gradient: best tile-per-tap cfg (1..27), best strip cfg (28..30), best LDS-DMA cfg
(31..42 four stages, 43..54 two stages).    python tools/wgrad_cfg_probe.py [--shapes 32]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gvcnn_tf_amd import _lib
from gvcnn_tf_amd.training import TrainGVCNN
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, default=32)
ap.add_argument("--backbone", default="inception_v3")
ap.add_argument("--storage", default="bf16")
a = ap.parse_args()
dev = torch.device("cuda:0")
eng = TrainGVCNN(a.backbone, a.shapes, 12, 224, 224, 40, 7, device=dev, num_bins=7, storage=a.storage)
lib = _lib.load()
x = (torch.rand(a.shapes, 12, 224, 224, 3, device=dev) - 0.5)
eng.forward(x, torch.zeros(a.shapes, dtype=torch.int64), check=False)
eng.backward()
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
n = lib.gv_conv2d_wgrad_num_cfgs(eng.dt)
tot = [0.0, 0.0, 0.0, 0.0, 0.0]
for op in eng.plan.ops:
    if op["kind"] != "conv":
        continue
    xx, y = op["x"], op["y"]
    dw = torch.empty_like(eng._dw(op))
    best = {}
    for cfg in range(0, n + 1):
        op["tile_w"] = cfg
        d = eng._conv_desc(op, wgrad=True)
        args = (C.byref(d), eng._ptr(xx), eng._ptr(y, True), y.ld, dw.data_ptr(), st)
        if lib.gv_conv2d_wgrad(*args) != 0:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.gv_conv2d_wgrad(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        fam = 0 if cfg == 0 else (1 if cfg <= 27 else (2 if cfg <= 30 else (3 if cfg <= 42 else 4)))
        if fam not in best or ms < best[fam][0]:
            best[fam] = (ms, cfg)
    op["tile_w"] = 0
    flops = 2.0 * y.nb * y.h * y.w * op["kh"] * op["kw"] * xx.c * y.c
    row = " ".join("%s %.4f(c%d)" % (nm, best[f][0], best[f][1]) if f in best else "%s --" % nm
                   for f, nm in enumerate(("default", "tile", "strip", "dma", "dma2")))
    print("%-52s cin=%4d cout=%4d k=%dx%d s%d M=%8d  %s  best %.0f TF/s" % (
        op["name"][-52:], xx.c, y.c, op["kh"], op["kw"], op["stride"], y.nb * y.h * y.w, row,
        flops / min(v[0] for v in best.values()) / 1e9))
    for f in range(5):
        tot[f] += best[f][0] if f in best else min(v[0] for v in best.values())
print("sum: default %.3f  tile %.3f  strip %.3f  dma %.3f  dma2 %.3f ms" % tuple(tot))
