#!/usr/bin/env python3
"""Run ONE conv launch of the Inception plan repeatedly (for rocprofv3 --pmc passes).
    python tools/one_conv.py --op 5 --tile 2 --dbg 0 --reps 20"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gvcnn_tf_amd as gv  # noqa: E402
from gvcnn_tf_amd import _lib, backbones  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="inception_v3")
ap.add_argument("--shapes", type=int, default=32)
ap.add_argument("--op", type=int, nargs="+", default=[5])
ap.add_argument("--tile", type=int, default=-1)
ap.add_argument("--dbg", type=int, nargs="+", default=[0])
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--storage", default="f32")
ap.add_argument("--math", default="f32")
a = ap.parse_args()
dev = torch.device("cuda:0")
nb = a.shapes * 12
plan = backbones.make_plan(a.backbone, nb, 224, 224, dev, dtype=a.storage, math=a.math)
plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
x = (torch.rand(nb, 224, 224, 3) - 0.5).to(dev)
plan.run(x)
torch.cuda.synchronize()
lib = _lib.load()
lib.gv_conv2d_set_tile_override(a.tile)
for op in a.op:
    for dbg in a.dbg:
        lib.gv_conv2d_set_debug(dbg)
        ms = plan.time_range(x, op, 1, a.reps)
        o = plan.ops[op]
        print("op %d %s tile %d dbg %d: %.4f ms  %.1f TF/s" % (op, o["name"], a.tile, dbg, ms, o["flops"] / ms / 1e9))
lib.gv_conv2d_set_debug(0)
