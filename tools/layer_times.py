#!/usr/bin/env python3
"""Per-launch timing of one backbone plan on the GPU (hipEvents on the launch stream).
    python tools/layer_times.py [--backbone inception_v3] [--shapes 32] [--views 12] [--size 224] [--tiles]
Prints one row per op: ms, TFLOP/s (convs) or GB/s (pools), and with --tiles the time of every
tile configuration of the conv kernel (tuning aid)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gvcnn_tf_amd as gv  # noqa: E402
from gvcnn_tf_amd import _lib, backbones  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="inception_v3")
ap.add_argument("--shapes", type=int, default=32)
ap.add_argument("--views", type=int, default=12)
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--tiles", action="store_true")
ap.add_argument("--json", default=None)
ap.add_argument("--ablate", action="store_true", help="time ablated kernels (no loads / no LDS writes / no stores)")
ap.add_argument("--autotune", action="store_true")
ap.add_argument("--ablate-bits", type=int, default=4)
ap.add_argument("--stored-preact", action="store_true", help="16-bit ResNet: every pre-activation stored by the unit before (A/B)")
ap.add_argument("--p3", default="default", help="three-plane intermediates: default | all | none | comma separated block names")
ap.add_argument("--no-fuse", action="store_true")
ap.add_argument("--math", default="f32")
ap.add_argument("--storage", default="f32")
a = ap.parse_args()

dev = torch.device("cuda:0")
nb = a.shapes * a.views
p3 = {"default": True, "all": "all", "none": False}.get(a.p3, tuple(a.p3.split(",")))
plan = backbones.make_plan(a.backbone, nb, a.size, a.size, dev, math=a.math, dtype=a.storage, p3=p3,
                           defer_preact=not a.stored_preact)
P = gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True)
plan.bind(P)
x = (torch.rand(nb, a.size, a.size, 3) - 0.5).to(dev)
plan.run(x)
torch.cuda.synchronize()
lib = _lib.load()
if a.autotune:
    plan.autotune(x)
ncfg = lib.gv_conv2d_num_tile_cfgs(plan.math_mode if a.storage == 'f32' else -1)
rows = []
tot = {"conv": 0.0, "pool": 0.0, "ssa": 0.0}
for i, op in enumerate(plan.ops):
    ms = plan.time_range(x, i, 1, a.iters)
    tot[op["kind"]] += ms
    row = {"i": i, "kind": op["kind"], "name": op["name"], "ms": ms}
    if op["kind"] == "conv":
        xx, y = op["x"], op["y"]
        row.update(M=y.npix, N=y.c, K=op["kh"] * op["kw"] * xx.c, tflops=op["flops"] / ms / 1e9,
                   gbs=op["bytes"] / ms / 1e6)
        if a.tiles:
            tt = []
            for t in range(lib.gv_conv2d_num_tile_cfgs(-3) if xx.p3 else ncfg):       # (three-plane input: the LDS-DMA kernel's own table)
                lib.gv_conv2d_set_tile_override(t)
                try:
                    tt.append(round(plan.time_range(x, i, 1, 3), 4))
                except _lib.GvError:                       # a kernel that does not take this layer (halo / stem strips)
                    tt.append(float("inf"))
            lib.gv_conv2d_set_tile_override(-1)
            row["tile_ms"] = tt
        if a.ablate:
            ab = []
            for bits in (a.ablate_bits,):
                lib.gv_conv2d_set_debug(bits)
                ab.append(round(plan.time_range(x, i, 1, 3), 4))
            lib.gv_conv2d_set_debug(0)
            row["ablate_ms"] = ab
        print("%3d conv %-52s M=%8d N=%4d K=%5d %8.4f ms %7.2f TF/s %7.1f GB/s %s" % (
            i, op["name"][-52:], row["M"], row["N"], row["K"], ms, row["tflops"], row["gbs"],
            str(row.get("tile_ms", "")) + (" ablate[nostore]=%s" % row["ablate_ms"] if a.ablate else "") + (" tile=%d" % (op.get("tile", 0) - 1))))
    else:
        row.update(gbs=op["bytes"] / ms / 1e6)
        print("%3d %-4s %-52s %8.4f ms %7.1f GB/s" % (i, op["kind"], op["name"][-52:], ms, row["gbs"]))
    rows.append(row)
flops = plan.total_flops
print("total: conv %.3f ms, pool %.3f ms, ssa %.3f ms; conv %.2f TF/s" % (
    tot["conv"], tot["pool"], tot["ssa"], flops / tot["conv"] / 1e9))
whole = plan.time_range(x, 0, len(plan.ops), a.iters)
print("whole plan back-to-back: %.3f ms (%.2f TF/s), activations %.1f MB" % (
    whole, flops / whole / 1e9, plan.act_bytes / 1e6))
if a.ablate:
    lib.gv_conv2d_set_debug(a.ablate_bits)
    wa = plan.time_range(x, 0, len(plan.ops), a.iters)
    lib.gv_conv2d_set_debug(0)
    print("whole plan with ablation bits %d: %.3f ms" % (a.ablate_bits, wa))
if a.json:
    json.dump({"rows": rows, "totals": tot, "whole_ms": whole}, open(a.json, "w"))
