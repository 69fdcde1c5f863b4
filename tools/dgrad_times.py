#!/usr/bin/env python3
"""Per-layer time of the forward and data-gradient convolution launches of the training plan (autotuned tiles).
    python tools/dgrad_times.py [--shapes 32]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402
from gvcnn_tf_amd.training import TrainGVCNN  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, default=32)
ap.add_argument("--backbone", default="inception_v3")
ap.add_argument("--storage", default="f32")
a = ap.parse_args()
dev = torch.device("cuda:0")
eng = TrainGVCNN(a.backbone, a.shapes, 12, 224, 224, 40, 7, device=dev, num_bins=7, storage=a.storage)
lib = _lib.load()
x = (torch.rand(a.shapes, 12, 224, 224, 3, device=dev) - 0.5)
eng.forward(x, torch.zeros(a.shapes, dtype=torch.int64), check=False)
eng.backward()
eng.autotune()
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
ms = C.c_float(0)
tf = td = 0.0
rows = []
for op in eng.plan.ops:
    if op["kind"] != "conv":
        continue
    xx, y = op["x"], op["y"]
    d = eng._conv_desc(op)
    d.res_ld = 0
    lib.gv_conv2d_time(C.byref(d), eng._ptr(xx), op["w_fwd"].data_ptr(), eng.ones.data_ptr(), eng.zeros.data_ptr(),
                       eng._ptr(y), 3, C.byref(ms), st)
    f = ms.value
    g = 0.0
    if xx.vbuf >= 0:
        dd = eng._conv_desc(op, dgrad=True)
        dd.res_ld = 0
        lib.gv_conv2d_time(C.byref(dd), eng._ptr(y, True), op["w_dgrad"].data_ptr(), eng.ones.data_ptr(),
                           eng.zeros.data_ptr(), eng._ptr(xx, True), 3, C.byref(ms), st)
        g = ms.value
    tf += f
    td += g
    rows.append((g, f, op))
for g, f, op in sorted(rows, key=lambda r: -r[0])[:25]:
    xx, y = op["x"], op["y"]
    print("%-52s M=%8d cin=%4d cout=%4d k=%dx%d s%d  fwd %7.3f ms %6.1f TF   dgrad %7.3f ms %6.1f TF (tile %d)" % (
        op["name"][-52:], y.npix, xx.c, y.c, op["kh"], op["kw"], op["stride"], f, op["flops"] / f / 1e9, g,
        op["flops"] / g / 1e9 if g else 0.0, op.get("tile_d", 0) - 1))
print("total forward %.2f ms, data gradient %.2f ms" % (tf, td))
