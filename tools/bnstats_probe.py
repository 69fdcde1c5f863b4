#!/usr/bin/env python3
"""Time one training convolution with and without the folded BatchNorm sums (gv_conv2d_fwd_bnstats), per debug bit.
    python tools/bnstats_probe.py --layer Conv2d_4a_3x3 [--bwd] [--tiles 0 7 12 18]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gvcnn_tf_amd import _lib
from gvcnn_tf_amd.training import TrainGVCNN

ap = argparse.ArgumentParser()
ap.add_argument("--layer", nargs="+", default=["Conv2d_4a_3x3"])
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--tiles", type=int, nargs="+", default=[0])
ap.add_argument("--dbg", type=int, nargs="+", default=[0, 2048, 2048 + 4096, 2048 + 4096 + 8192])
ap.add_argument("--shapes", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")
eng = TrainGVCNN("inception_v3", a.shapes, 12, 224, 224, 40, 7, device=dev, num_bins=7, storage="bf16")
x = (torch.rand(a.shapes, 12, 224, 224, 3, device=dev) - 0.5)
eng.forward(x, torch.zeros(a.shapes, dtype=torch.int64), check=False)
eng.backward()
torch.cuda.synchronize()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rc = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, rc


for op in eng.plan.ops:
    if op["kind"] != "conv" or not any(op["name"].endswith(l) for l in a.layer):
        continue
    key = "st_b" if a.bwd else "st_f"
    if not op.get(key):
        print(op["name"], "has no", key)
        continue
    for t in a.tiles:
        op["tile_d" if a.bwd else "tile_f"] = t
        d = eng._conv_desc(op, dgrad=a.bwd)
        if a.bwd:
            d.res_ld = 0
            src, w, dst = eng._ptr(op["y"], True), op["w_dgrad"], eng._ptr(op["x"], True)
        else:
            src = eng._x32.data_ptr() if op["x"].vbuf < 0 else eng._ptr(op["x"])
            w, dst = op["w_fwd"], eng._ptr(op["y"])
        stt = eng._bn_stats(op, key)
        plain = lambda: lib.gv_conv2d_fwd(C.byref(d), src, w.data_ptr(), eng.ones.data_ptr(), eng.zeros.data_ptr(), None, dst, None, None, None, st)
        fused = lambda: lib.gv_conv2d_fwd_bnstats(C.byref(d), src, w.data_ptr(), eng.ones.data_ptr(), eng.zeros.data_ptr(), None, dst, C.byref(stt), st)
        lib.gv_conv2d_set_debug(0)
        tp, rc0 = timeit(plain)
        row = "%-50s tile %2d plain %.3f ms (rc %d) |" % (op["name"][-50:], t, tp, rc0)
        for dbg in a.dbg:
            lib.gv_conv2d_set_debug(dbg)
            tf_, rc = timeit(fused)
            row += " dbg %5d: %.3f (rc %d)" % (dbg, tf_, rc)
        lib.gv_conv2d_set_debug(0)
        print(row)
