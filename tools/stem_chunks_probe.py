#!/usr/bin/env python3
"""The Inception stem (Conv2d_1a ... MaxPool_5a) over the whole batch layer by layer against slice by slice
(`make_plan(stem_chunks=C)`): the same launches on C slices of the batch, so that a layer's output is read back while it is
still in the Infinity Cache.  Autotuned, timed in sequence on one box; the final end point is compared bit for bit.
    python tools/stem_chunks_probe.py [--preset c3] [--chunks 1,2,3,4,6,8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gvcnn_tf_amd as gv  # noqa: E402
from gvcnn_tf_amd import backbones  # noqa: E402

PRESETS = {"c2": (12, 224, "f32", "bf16x3"), "c3": (12, 224, "bf16", "f32"), "c5": (20, 299, "f16", "f32")}
ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="c3")
ap.add_argument("--shapes", type=int, default=32)
ap.add_argument("--chunks", default="1,2,3,4,6,8")
ap.add_argument("--detail", action="store_true")
a = ap.parse_args()
V, size, storage, math = PRESETS[a.preset]
dev = torch.device("cuda:0")
nb = a.shapes * V
x = (torch.rand(nb, size, size, 3) - 0.5).to(dev)
ref = None
for ck in [int(c) for c in a.chunks.split(",")]:
    plan = backbones.make_plan("inception_v3", nb, size, size, dev, math=math, dtype=storage, lanes=False, stem_chunks=ck)
    plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
    plan.autotune(x)
    seq = [min(p, q) for p, q in zip(plan.time_each(x, 10), plan.time_each(x, 10))]
    stem = sum(s for s, op in zip(seq, plan.ops) if "Mixed" not in op["name"])
    rest = sum(s for s, op in zip(seq, plan.ops) if "Mixed" in op["name"])
    whole = min(plan.time_range(x, 0, len(plan.ops), 10), plan.time_range(x, 0, len(plan.ops), 10))
    plan.run(x)
    torch.cuda.synchronize()
    out = plan.view(plan.end_points["Mixed_7c"]).clone()
    same = "" if ref is None else ("  final end point identical: %s" % bool(torch.equal(out, ref)))
    ref = out if ref is None else ref
    print("%s stem_chunks %d: stem %.3f ms (%d launches), rest %.3f ms, whole pass %.3f ms = %.0f views/s%s"
          % (a.preset, ck, stem, sum(1 for op in plan.ops if "Mixed" not in op["name"]), rest, whole, nb / whole * 1e3, same),
          flush=True)
    if a.detail:
        per = {}
        for s, op in zip(seq, plan.ops):
            if "Mixed" not in op["name"]:
                k = op["name"].split("@")[0].split("/")[-1]
                per[k] = per.get(k, 0.0) + s
        print("     " + "  ".join("%s %.3f" % (k.replace("Conv2d_", "").replace("MaxPool_", "mp"), v) for k, v in per.items()))
    del plan
