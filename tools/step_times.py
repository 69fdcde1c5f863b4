#!/usr/bin/env python3
"""Per-launch, in-sequence times of the training step (gvcnn_tf_amd.steptime): one table row per op with its forward and
backward launches, then the sums per family.
    python tools/step_times.py [--shapes 32] [--storage bf16] [--backbone inception_v3] [--kind bn|conv|pool|all] [--tune]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import steptime  # noqa: E402
from gvcnn_tf_amd.training import TrainGVCNN  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", type=int, default=32)
ap.add_argument("--views", type=int, default=12)
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--backbone", default="inception_v3")
ap.add_argument("--storage", default="bf16")
ap.add_argument("--kind", default="all")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--tune", action="store_true")
ap.add_argument("--set", default="", help="comma-separated engine attributes to set, e.g. fuse_bn_stats=0")
a = ap.parse_args()
dev = torch.device("cuda:0")
eng = TrainGVCNN(a.backbone, a.shapes, a.views, a.size, a.size, 40, 7, device=dev, num_bins=7, storage=a.storage)
for kv in filter(None, a.set.split(",")):
    k, v = kv.split("=")
    setattr(eng, k, type(getattr(eng, k))(int(v)))
eng._plan_bn_fusion()                                     # (the fusion switches are read when the plan is laid out)
x = (torch.rand(a.shapes, a.views, a.size, a.size, 3, device=dev) - 0.5)
labels = torch.randint(0, 40, (a.shapes,), device=dev)
eng.train_step(x, labels, lr=1e-6)
if a.tune:
    eng.autotune()
eng.train_step(x, labels, lr=1e-6)
torch.cuda.synchronize()
steps = steptime.timed_step(eng, x, labels, steps=a.steps)
# average the steps launch by launch (same launch list every step)
n = len(steps[0])
assert all(len(s) == n for s in steps)
avg = [(steps[0][i][0], steps[0][i][1], steps[0][i][2], steps[0][i][3], sum(s[i][4] for s in steps) / len(steps)) for i in range(n)]
rows = {}
order = []
for fn, name, phase, kind, ms in avg:
    if name not in rows:
        rows[name] = dict(kind=kind, fwd=[], bwd=[])
        order.append(name)
    rows[name][phase].append((fn.replace("gv_", "").replace("_grouped_t", "").replace("conv2d_", ""), ms))
els = {op["name"]: (op["y"].npix, op["y"].c) for op in eng.plan.ops}
kdim = {op["name"]: op["kh"] * op["kw"] * op["x"].c for op in eng.plan.ops if op["kind"] == "conv"}
tiles = {op["name"]: (op.get("tile", 0), op.get("tile_d", 0), op.get("tile_w", 0)) for op in eng.plan.ops if op["kind"] == "conv"}
for name in order:
    r = rows[name]
    if a.kind != "all" and r["kind"] != a.kind:
        continue
    npix, c = els.get(name, (0, 0))
    f = " ".join("%s %.3f" % fm for fm in r["fwd"])
    b = " ".join("%s %.3f" % fm for fm in r["bwd"])
    extra = ""
    if name in kdim:                                      # GEMM depth, launch configurations (fwd, dgrad, wgrad) and the
        gf = 2.0 * npix * c * kdim[name] * 1e-9           # rate of each of the three GEMMs (all launches of a pass summed)
        tf = lambda ls: gf / max(sum(ms for fn, ms in ls if fn.startswith("fwd")), 1e-9)
        tw = gf / max(sum(ms for fn, ms in r["bwd"] if fn in ("wgrad", "wgrad_ws")), 1e-9)
        extra = " | K=%d cfg=%s TF/s fwd %.0f dgrad %.0f wgrad %.0f" % (kdim[name], tiles[name], tf(r["fwd"]), tf(r["bwd"]), tw)
    print("%-58s %-5s M=%8d c=%4d | fwd: %s | bwd: %s%s" % (name[-58:], r["kind"], npix, c, f, b, extra))
fam = steptime.by_family(avg)
tot = sum(t for t, _ in fam.values())
print("in-sequence sum %.2f ms: " % tot + ", ".join("%s %.2f ms / %d" % (k, t, cnt) for k, (t, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][0])))
