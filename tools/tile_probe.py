#!/usr/bin/env python3
"""Time chosen conv launches of a plan under every tile configuration of their kernel family.
    python tools/tile_probe.py --ops 5 9 30 [--storage bf16] [--math f32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gvcnn_tf_amd as gv
from gvcnn_tf_amd import _lib, backbones
ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="inception_v3")
ap.add_argument("--ops", type=int, nargs="+", default=[5, 9, 10, 30])
ap.add_argument("--storage", default="bf16")
ap.add_argument("--math", default="f32")
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
nb = 32 * 12
plan = backbones.make_plan(a.backbone, nb, a.size, a.size, dev, dtype=a.storage, math=a.math)
plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
x = (torch.rand(nb, a.size, a.size, 3) - 0.5).to(dev)
plan.run(x)
torch.cuda.synchronize()
lib = _lib.load()
for op in a.ops:
    o = plan.ops[op]
    n = lib.gv_conv2d_num_tile_cfgs(-3 if o["x"].p3 else (-1 if a.storage != "f32" else plan.math_mode))
    row = []
    for t in range(n):
        lib.gv_conv2d_set_tile_override(t)
        try:
            row.append("%d:%.4f" % (t, plan.time_range(x, op, 1, a.reps)))
        except Exception:
            row.append("%d:--" % t)
    lib.gv_conv2d_set_tile_override(-1)
    print("op %d %s %.1f GF: %s" % (op, o["name"][-36:], o["flops"] / 1e9, " ".join(row)))
