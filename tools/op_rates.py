#!/usr/bin/env python3
"""Every op of an inference plan in sequence: time, TFLOP/s and algorithmic bytes / time (TB/s) — which bound a launch sits at.
    python tools/op_rates.py c4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gvcnn_tf_amd as gv  # noqa: E402
from gvcnn_tf_amd import backbones  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "c4"
PRESETS = {"c2": ("inception_v3", 12, 224, "f32", "bf16x3"), "c3": ("inception_v3", 12, 224, "bf16", "f32"),
           "c4": ("resnet_v2_50", 12, 224, "bf16", "f32"), "c5": ("inception_v3", 20, 299, "f16", "f32")}
backbone, V, size, storage, math = PRESETS[preset]
dev = torch.device("cuda:0")
nb = 32 * V
plan = backbones.make_plan(backbone, nb, size, size, dev, math=math, dtype=storage, lanes=False)
plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
x = (torch.rand(nb, size, size, 3) - 0.5).to(dev)
plan.autotune(x)
seq = [min(p, q) for p, q in zip(plan.time_each(x, 10), plan.time_each(x, 10))]
tot = 0.0
for i, op in enumerate(plan.ops):
    xx, y = op["x"], op["y"]
    tot += seq[i]
    print("%-52s %-5s %4dx%-4d c %4d -> %4d | tile %3d | %.4f ms | %6.0f TF/s | %5.2f TB/s" % (
        op["name"][-52:], op["kind"], xx.h, xx.w, xx.c, y.c, int(op.get("tile", 0)) - 1, seq[i], op["flops"] / seq[i] / 1e9,
        op["bytes"] / seq[i] / 1e9))
print("sum %.3f ms" % tot)
