#!/usr/bin/env python3
"""configs[1] (fp32 storage, bf16x3 math) with the block outputs kept fp32 (default) against three-plane block outputs
("concat": the sibling GEMMs and the strided 3x3 of Mixed_6a / 7a read three planes through the LDS-DMA kernel instead of
splitting fp32 values in their loader; every concat writer and pool stores 6 bytes per value).  Both autotuned, timed in
sequence on the same box.
    python tools/p3_concat_probe.py [--shapes 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gvcnn_tf_amd as gv  # noqa: E402
from gvcnn_tf_amd import backbones  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, default=32)
ap.add_argument("--detail", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
nb = a.shapes * 12
x = (torch.rand(nb, 224, 224, 3) - 0.5).to(dev)
for label, p3 in (("default", True), ("default+concat", set(backbones.P3_DEFAULT_BLOCKS) | {"concat"})):
    plan = backbones.make_plan("inception_v3", nb, 224, 224, dev, math="bf16x3", dtype="f32", lanes=False, p3=p3)
    plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
    plan.autotune(x)
    seq = [min(p, q) for p, q in zip(plan.time_each(x, 10), plan.time_each(x, 10))]
    conv = sum(s for s, op in zip(seq, plan.ops) if op["kind"] == "conv")
    other = sum(s for s, op in zip(seq, plan.ops) if op["kind"] != "conv")
    fl = sum(op["flops"] for op in plan.ops if op["kind"] == "conv")
    whole = min(plan.time_range(x, 0, len(plan.ops), 10), plan.time_range(x, 0, len(plan.ops), 10))
    print("%-16s conv %.3f ms = %.1f TF/s (frac %.4f), other ops %.3f ms, sum %.3f ms; whole pass %.3f ms = %.0f views/s"
          % (label, conv, fl / conv / 1e9, fl / conv / 1e9 / 416.7, other, conv + other, whole, nb / whole * 1e3), flush=True)
    if a.detail:
        for s, op in zip(seq, plan.ops):
            print("    %-70s %-5s %8.4f ms %6.0f TF/s%s" % (op["name"][-70:], op["kind"], s, op.get("flops", 0) / s / 1e9,
                                                          " p3in" if op["x"].p3 else ""))
