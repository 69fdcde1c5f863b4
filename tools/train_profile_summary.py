#!/usr/bin/env python3
"""Per-step kernel time by family from a rocprofv3 kernel trace of tools/train_bench.py (the LAST 3 steps = the timed
ones: everything after the autotune / warm-up is attributed by taking the trace's tail).
    python tools/train_profile_summary.py kt_kernel_trace.csv [steps=3]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps start at the last 3 occurrences of the loss kernel (one per step) minus one step's worth of kernels
idx = [i for i, r in enumerate(rows) if "softmax_ce" in r["Kernel_Name"]]
assert len(idx) >= steps + 1, "trace too short"
# a step = from just after the previous step's last kernel; use the gap between consecutive softmax_ce as the period
period = idx[-1] - idx[-2]
first = idx[-steps] - (idx[-1] - idx[-2]) + (len(rows) - idx[-1])   # same phase as the end of the trace
first = max(first, 0)
sel = rows[len(rows) - steps * period:]
FAM = [("filter gradient", ("conv_wgrad",)), ("filter gradient: slice reduce", ("dw_reduce_slices",)), ("conv fwd/dgrad", ("conv_igemm", "conv3x3_halo", "conv_stem_patch", "conv_dma", "conv_ws")),
       ("BN sums", ("grouped_sums",)), ("BN apply fwd/bwd", ("bn_stream", "scale_shift_act_grouped", "bn_bwd_apply")),
       ("pool fwd/bwd", ("pool2d", "maxpool", "avgpool3x3")), ("fills/copies", ("FillFunctor", "fillBuffer", "copyBuffer")),
       ("filter re-pack", ("pack_filter", "elementwise_kernel")), ("optimizer", ("sgd_momentum",)),
       ("BN small", ("bn_param_grads", "bn_finalize", "bn_update_moving"))]
tot = defaultdict(float)
cnt = defaultdict(int)
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    name = r["Kernel_Name"]
    fam = next((f for f, keys in FAM if any(k in name for k in keys)), "other")
    tot[fam] += d
    cnt[fam] += 1
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
busy = sum(tot.values())
print("%d kernels over the last %d steps: wall %.2f ms/step, kernels %.2f ms/step (GPU busy %.0f %%)"
      % (len(sel), steps, span / steps, busy / steps, 100 * busy / span))
for f, t in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-30s %7.2f ms/step  %5d launches/step" % (f, t / steps, cnt[f] // steps))
