#!/usr/bin/env python3
"""HBM traffic of every launch of one bench step against the launch's algorithmic bytes, from the FETCH_SIZE and WRITE_SIZE
passes of tools/pmc_bench.sh (single launch lane: the dispatches of a step are the plan's ops in order).  gfx950 corrections
as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2, KiB units).  The plan is rebuilt on the CPU (host logic only) for the
algorithmic side: one read of the input, the filter and the residual, one write of every output, every value 4 bytes (fp32
storage) or 2; "stored" prices a three-plane tensor (fp32 as three bf16 planes, 6 bytes per value: a format the plan chooses
where the consumer's time gains more than the bytes cost) at what it occupies, so measured / stored is what the KERNELS waste.
--trace: a rocprofv3 --kernel-trace CSV of the same single-lane step WITHOUT counters (tools/profile_round*.sh's `kt`): adds
each launch's duration (mean over the trace's last steps) and the HBM rate its measured bytes amount to.
    python tools/traffic_per_launch.py gpurun_out/pmc_TAG_c2 --preset c2 [--shapes 32] [--top 12] [--trace kt_kernel_trace.csv]"""
import argparse
import csv
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from bench import PRESETS, PRESET_STORAGE  # noqa: E402
from gvcnn_tf_amd import backbones  # noqa: E402

HEAD = ("view_score", "view_pool", "global_avg_pool", "dense_", "group_assign", "copyBuffer", "fillBuffer")


def last_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if "dense_f32" in r["Kernel_Name"] or "dense_lp" in r["Kernel_Name"]]
    rows = rows[ends[-2] + 1:ends[-1] + 1]
    return [r for r in rows if not any(h in r["Kernel_Name"] for h in HEAD)]


def durations(path, nops):
    """Mean duration (us) of each of the plan's launches over the trace's last (up to 10) steps."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "dense_f32" in r["Kernel_Name"] or "dense_lp" in r["Kernel_Name"]]
    steps = []
    for a_, b_ in list(zip(ends, ends[1:]))[-10:]:
        st = [r for r in rows[a_ + 1:b_ + 1] if not any(h in r["Kernel_Name"] for h in HEAD)]
        if len(st) == nops:
            steps.append([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in st])
    if not steps:
        sys.exit("no step of the trace has %d plan launches" % nops)
    return [sum(s[i] for s in steps) / len(steps) for i in range(nops)], len(steps)


def short(n):
    n = n.replace("void (anonymous namespace)::", "").split("(")[0]
    return n if len(n) <= 44 else n[:41] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--preset", default="c2", choices=sorted(PRESETS))
    ap.add_argument("--shapes", type=int, default=32)
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--trace", default=None)
    a = ap.parse_args()
    backbone, V, H, _, _ = PRESETS[a.preset]
    st = PRESET_STORAGE[a.preset]
    plan = backbones.make_plan(backbone, a.shapes * V, H, H, torch.device("cpu"), dtype=st,
                               math="bf16x3" if st == "f32" else "f32", lanes=False)
    f = last_step(os.path.join(a.dir, "f", "p_counter_collection.csv"), "FETCH_SIZE")
    w = last_step(os.path.join(a.dir, "w", "p_counter_collection.csv"), "WRITE_SIZE")
    ops = plan.ops
    if not (len(f) == len(w) == len(ops)):
        sys.exit("dispatches of the last step (%d fetch, %d write) do not line up with the plan's %d ops" % (len(f), len(w), len(ops)))
    us, nsteps = durations(a.trace, len(ops)) if a.trace else ([0.0] * len(ops), 0)
    rows = []
    for op, rf, rw, t_us in zip(ops, f, w, us):
        assert rf["Kernel_Name"] == rw["Kernel_Name"], (rf["Kernel_Name"], rw["Kernel_Name"])
        fb = float(rf["Counter_Value"]) * 1024 * 2
        wb = float(rw["Counter_Value"]) * 1024
        stored = float(op["bytes"]) + sum(2.0 * t.nb * t.h * t.w * t.c for t in (op.get(k) for k in ("x", "y", "y2", "res"))
                                          if t is not None and t.p3)
        rows.append((op["name"], op["kind"], short(rf["Kernel_Name"]), fb, wb, float(op["bytes"]), stored, t_us))
    tot_m = sum(r[3] + r[4] for r in rows)
    tot_a = sum(r[5] for r in rows)
    print("preset %s, %d x %d views: %d launches, measured %.1f MB (fetch %.1f + write %.1f), algorithmic %.1f MB: %.3fx"
          % (a.preset, a.shapes, V, len(rows), tot_m / 1e6, sum(r[3] for r in rows) / 1e6, sum(r[4] for r in rows) / 1e6,
             tot_a / 1e6, tot_m / tot_a))
    print("as stored (three-plane tensors at 6 bytes per value): %.1f MB: measured / stored %.3fx"
          % (sum(r[6] for r in rows) / 1e6, tot_m / sum(r[6] for r in rows)))
    conv = [r for r in rows if r[1] == "conv"]
    print("conv launches only: %d, measured %.1f MB, algorithmic %.1f MB: %.3fx"
          % (len(conv), sum(r[3] + r[4] for r in conv) / 1e6, sum(r[5] for r in conv) / 1e6,
             sum(r[3] + r[4] for r in conv) / sum(r[5] for r in conv)))
    print("conv launches, as stored: %.1f MB: measured / stored %.3fx"
          % (sum(r[6] for r in conv) / 1e6, sum(r[3] + r[4] for r in conv) / sum(r[6] for r in conv)))
    hdr = "%-58s %-44s %9s %9s %9s %7s %9s %7s %9s" % ("op", "kernel", "fetch MB", "write MB", "alg MB", "ratio", "stored MB",
                                                       "ratio", "excess MB")
    fmt = "%-58s %-44s %9.1f %9.1f %9.1f %7.2f %9.1f %7.2f %9.1f"
    if a.trace:
        hdr += " %8s %10s" % ("us", "HBM GB/s")
        print("durations: mean of %d steps of %s; HBM GB/s = measured bytes / duration" % (nsteps, a.trace))

    def line(r):
        m = r[3] + r[4]
        s = fmt % (r[0][-58:], r[2], r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, m / r[5], r[6] / 1e6, m / r[6], (m - r[6]) / 1e6)
        return s + (" %8.1f %10.0f" % (r[7], m / (r[7] * 1e-6) / 1e9) if a.trace else "")
    print("\nevery launch, in plan order (excess = measured - stored)\n" + hdr)
    for r in rows:
        print(line(r))
    print("\nthe %d largest excesses\n" % a.top + hdr)
    for r in sorted(rows, key=lambda r: -(r[3] + r[4] - r[6]))[:a.top]:
        print(line(r))


if __name__ == "__main__":
    main()
