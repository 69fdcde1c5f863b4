#!/usr/bin/env python3
"""Every launch of an inference plan timed two ways on the same box: as a WARM REPEAT of itself (gv_plan_time: what the
autotuner's first pass and tools/conv_probe*.py see) and IN SEQUENCE (gv_plan_time_each: an event pair behind every op
over whole passes: what the step pays).  The difference is what a launch loses to its cold input / filter, its launch
boundary and its ramp and tail — not to its steady state.
    python tools/seq_vs_warm.py [--preset c3] [--shapes 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gvcnn_tf_amd as gv  # noqa: E402
from gvcnn_tf_amd import backbones  # noqa: E402

PRESETS = {"c2": ("inception_v3", 12, 224, "f32", "bf16x3"), "c3": ("inception_v3", 12, 224, "bf16", "f32"),
           "c4": ("resnet_v2_50", 12, 224, "bf16", "f32"), "c5": ("inception_v3", 20, 299, "f16", "f32")}
ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="c3")
ap.add_argument("--shapes", type=int, default=32)
a = ap.parse_args()
backbone, V, size, storage, math = PRESETS[a.preset]
dev = torch.device("cuda:0")
nb = a.shapes * V
plan = backbones.make_plan(backbone, nb, size, size, dev, math=math, dtype=storage, lanes=False)
plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
x = (torch.rand(nb, size, size, 3) - 0.5).to(dev)
plan.autotune(x)
warm = [min(plan.time_range(x, i, 1, 10), plan.time_range(x, i, 1, 10)) for i in range(len(plan.ops))]
seq = [min(p, q) for p, q in zip(plan.time_each(x, 10), plan.time_each(x, 10))]
tw = ts = 0.0
print("%-58s %9s %5s %5s | %4s | %8s %8s %6s | TF/s warm / in sequence | GB/s (algorithmic, in sequence)" % ("op", "M", "N", "K", "tile", "warm ms", "seq ms", "+us"))
for i, op in enumerate(plan.ops):
    if op["kind"] != "conv":
        continue
    xx, y = op["x"], op["y"]
    tw += warm[i]
    ts += seq[i]
    print("%-58s %9d %5d %5d | %4d | %8.4f %8.4f %6.1f | %4.0f / %4.0f | %5.0f" % (op["name"][-58:], y.npix, y.c, op["kh"] * op["kw"] * xx.c,
          int(op.get("tile", 0)) - 1, warm[i], seq[i], (seq[i] - warm[i]) * 1e3, op["flops"] / warm[i] / 1e9, op["flops"] / seq[i] / 1e9,
          op["bytes"] / seq[i] / 1e6))
fl = sum(op["flops"] for op in plan.ops if op["kind"] == "conv")
nconv = sum(1 for op in plan.ops if op["kind"] == "conv")
other_w = sum(w for w, op in zip(warm, plan.ops) if op["kind"] != "conv")
other_s = sum(s for s, op in zip(seq, plan.ops) if op["kind"] != "conv")
print("conv launches: %d; warm repeats %.3f ms = %.0f TF/s; in sequence %.3f ms = %.0f TF/s; +%.1f us per launch"
      % (nconv, tw, fl / tw / 1e9, ts, fl / ts / 1e9, (ts - tw) / nconv * 1e3))
print("other ops (pools ...): warm %.3f ms, in sequence %.3f ms" % (other_w, other_s))
