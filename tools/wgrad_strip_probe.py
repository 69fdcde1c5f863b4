#!/usr/bin/env python3
"""Filter gradient of the k>1 stride-1 layers: best tap-per-workgroup configuration vs the strip form with 9 / 5 / 4
taps per workgroup (tuning hook gv_conv2d_wgrad_set_strip_taps)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gvcnn_tf_amd import _lib
from gvcnn_tf_amd.training import TrainGVCNN
dev = torch.device("cuda:0")
eng = TrainGVCNN("inception_v3", 32, 12, 224, 224, 40, 7, device=dev, num_bins=7, storage="bf16")
lib = _lib.load()
x = torch.rand(32, 12, 224, 224, 3, device=dev) - 0.5
eng.forward(x, torch.zeros(32, dtype=torch.int64, device=dev), check=False); eng.backward(); torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
def t(op, cfg):
    op["tile_w"] = cfg
    d = eng._conv_desc(op, wgrad=True)
    xx, y = op["x"], op["y"]
    dw = eng._dw(op)
    args = (C.byref(d), eng._ptr(xx), eng._ptr(y, True), y.ld, dw.data_ptr(), st)
    if lib.gv_conv2d_wgrad(*args) != 0: return float("inf")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): lib.gv_conv2d_wgrad(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
tot = [0.0] * 4
seen = set()
for op in eng.plan.ops:
    if op["kind"] != "conv" or op["kh"] * op["kw"] < 2 or op["stride"] != 1 or op["x"].c <= 32: continue
    key = (op["kh"], op["kw"], op["x"].c, op["y"].c, op["y"].h)
    best = min(t(op, c) for c in range(1, 28))
    res = [best]
    for ntw in (9, 5, 4):
        lib.gv_conv2d_wgrad_set_strip_taps(ntw)
        res.append(min(t(op, c) for c in (28, 29, 30)))
    lib.gv_conv2d_wgrad_set_strip_taps(5)
    for i, r in enumerate(res): tot[i] += r
    if key not in seen:
        seen.add(key)
        print("%-45s %dx%d %4d->%4d @%3d  per-tap %.3f  strip9 %.3f  strip5 %.3f  strip4 %.3f" % ((op["name"][-45:],) + key[:5] + tuple(res)))
print("totals: per-tap %.2f  strip9 %.2f  strip5 %.2f  strip4 %.2f ms" % tuple(tot))
