#!/usr/bin/env python3
"""Per-layer time of the filter-gradient kernel (v1 vs v2) on the Inception training geometry.
    python tools/wgrad_times.py [--shapes 32]"""
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
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
tot = [0.0, 0.0]
for op in eng.plan.ops:
    if op["kind"] != "conv":
        continue
    xx, y = op["x"], op["y"]
    d = eng._conv_desc(op)
    dw = eng._dw(op)
    res = []
    for v1 in (1, 0):
        if a.storage == "f32":
            lib.gv_conv2d_wgrad_set_v1(v1)
        else:                                    # 16-bit storage: column 1 = fp32 MFMA with typed loads, column 2 = 16-bit MFMA
            lib.gv_conv2d_wgrad_set_lp_f32(v1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.gv_conv2d_wgrad(C.byref(d), eng._ptr(xx), eng._ptr(y, True), y.ld, dw.data_ptr(), st)
        e0.record()
        for _ in range(3):
            lib.gv_conv2d_wgrad(C.byref(d), eng._ptr(xx), eng._ptr(y, True), y.ld, dw.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 3)
    lib.gv_conv2d_wgrad_set_v1(0)
    lib.gv_conv2d_wgrad_set_lp_f32(0)
    tot[0] += res[0]
    tot[1] += res[1]
    fl = op["flops"]
    print("%-55s M=%8d cin=%4d cout=%4d k=%dx%d  v1 %7.3f ms %6.1f TF   v2 %7.3f ms %6.1f TF" % (
        op["name"][-55:], y.npix, xx.c, y.c, op["kh"], op["kw"], res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9))
print("total v1 %.2f ms, v2 %.2f ms" % tuple(tot))
