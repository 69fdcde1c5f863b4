#!/usr/bin/env python3
"""Run the inference plan op by op twice on the same input and report every op whose output differs between the two
passes (a race inside a kernel shows up at the first such op).   python tools/determinism_probe.py [--preset c2] [--shapes 8] [--reps 3]"""
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
ap.add_argument("--preset", default="c2")
ap.add_argument("--shapes", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
backbone, V, size, storage, math = PRESETS[a.preset]
dev = torch.device("cuda:0")
nb = a.shapes * V
plan = backbones.make_plan(backbone, nb, size, size, dev, math=math, dtype=storage, lanes=False)
plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
x = (torch.rand(nb, size, size, 3) - 0.5).to(dev)
plan.autotune(x)
ref = None
bad = {}
for rep in range(a.reps + 1):
    outs = []
    for i, op in enumerate(plan.ops):
        plan.run_range(x, i, 1)
        torch.cuda.synchronize()
        o = [plan.view(op["y"]).clone()]
        if op.get("y2") is not None:
            o.append(plan.view(op["y2"]).clone())
        outs.append(o)
    if ref is None:
        ref = outs
        continue
    for i, (p, q) in enumerate(zip(ref, outs)):
        for u, v in zip(p, q):
            if not torch.equal(u, v):
                d = (u.float() - v.float()).abs()
                bad.setdefault(i, []).append((int((d > 0).sum()), float(d.max())))
for i, v in sorted(bad.items()):
    op = plan.ops[i]
    print("op %3d %-60s kind %s tile %s: differs in %s" % (i, op["name"][-60:], op["kind"], op.get("tile"), v))
print("ops that differ between passes: %d of %d" % (len(bad), len(plan.ops)))
