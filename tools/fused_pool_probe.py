#!/usr/bin/env python3
"""Conv2d_2b_3x3 -> MaxPool_3a_3x3 (Inception) / conv1 -> pool1 (ResNet) as one launch (GV_CONV_MAXPOOL3S2[_SAME],
`make_plan(fuse_maxpool=True)`) against the two launches:
both plans autotuned and timed in sequence on one box.
    python tools/fused_pool_probe.py [--preset c3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gvcnn_tf_amd as gv  # noqa: E402
from gvcnn_tf_amd import backbones  # noqa: E402

PRESETS = {"c3": ("inception_v3", 12, 224, "bf16", "f32"), "c4": ("resnet_v2_50", 12, 224, "bf16", "f32"),
           "c5": ("inception_v3", 20, 299, "f16", "f32")}
ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="c3")
ap.add_argument("--shapes", type=int, default=32)
a = ap.parse_args()
backbone, V, size, storage, math = PRESETS[a.preset]
dev = torch.device("cuda:0")
nb = a.shapes * V
x = (torch.rand(nb, size, size, 3) - 0.5).to(dev)
for fuse in (False, True, False, True):
    plan = backbones.make_plan(backbone, nb, size, size, dev, math=math, dtype=storage, lanes=False, fuse_maxpool=fuse)
    plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
    plan.autotune(x)
    seq = [min(p, q) for p, q in zip(plan.time_each(x, 10), plan.time_each(x, 10))]
    part = sum(s for s, op in zip(seq, plan.ops) if op["name"].endswith(("Conv2d_2b_3x3", "MaxPool_3a_3x3", "/conv1", "/pool1"))
               and "block" not in op["name"])
    whole = min(plan.time_range(x, 0, len(plan.ops), 10), plan.time_range(x, 0, len(plan.ops), 10))
    print("%s fuse_maxpool %-5s: conv (+ max pool) %.4f ms; whole pass %.3f ms = %.0f views/s"
          % (a.preset, fuse, part, whole, nb / whole * 1e3), flush=True)
    del plan
