#!/usr/bin/env python3
"""Summarise the passes of tools/pmc_bench.sh per kernel name over the LAST `steps` steps of the run."""
import collections
import csv
import glob
import re
import sys

root, steps = sys.argv[1], int(sys.argv[2])


def rows_of(tag):
    files = glob.glob("%s/%s/**/*counter_collection.csv" % (root, tag), recursive=True)
    rows = list(csv.DictReader(open(files[0]))) if files else []
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = list(by_disp.values())
    ends = [i for i, d in enumerate(disp) if "dense_f32" in d["name"]]          # a step ends with the classifier
    start = ends[-steps - 1] + 1 if len(ends) > steps else 0
    return disp[start:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_18conv_dmaIDF16(.)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", n)
    if m:
        t, np_, wm, wn, tm, tn, st, epi = m.groups()
        return "conv_dma<%s,NP=%s,%sx%s waves,%sx%s tiles,ST=%s,EPI=%s>" % ("bf16" if t == "b" else "f16", np_, wm, wn, tm, tn, st, epi)
    m = re.match(r"_ZN12_GLOBAL__N_113conv_chain_lpIDF16(.)Li(\d+)ELi(\d)ELi(\d)ELb(\d)E", n)
    if m:
        t, d, nw, rvs, front = m.groups()
        return "conv_chain_lp<%s,d=%s,%s waves,%s>" % ("bf16" if t == "b" else "f16", d, nw, "conv2 in front (unit)" if front == "1" else "chain")
    m = re.match(r"_ZN12_GLOBAL__N_17conv_wsIDF16(.)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d+)ELi(\d)E", n)
    if m:
        t, wm, wn, tm, tn, nb, st, gemm, kt, nl = m.groups()
        return "conv_ws<%s,%sx%s waves,%sx%s tiles,NB=%s,%s,KT=%s,NL=%s>" % ("bf16" if t == "b" else "f16", wm, wn, tm, tn, nb, "GEMM" if gemm == "1" else "strip", kt, nl)
    return n.split("(")[0].replace("void ", "")[:70]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for tag in ("f", "w", "s", "l"):
    for d in rows_of(tag):
        k = short(d["name"])
        if tag == "f":
            cnt[k] += 1
        for c, v in d.items():
            if c != "name":
                agg[k][c] += v
tot_f = tot_w = 0.0
print("per kernel, summed over the last %d steps (FETCH_SIZE x2 gfx950 correction, KiB -> bytes):" % steps)
print("%-62s %5s %9s %9s %8s %8s %7s %7s %7s" % ("kernel", "n", "fetchMB", "writeMB", "mfmaBusy", "valu/mfma", "waitAny", "waitIns", "L2hit"))
for k in sorted(agg, key=lambda k: -agg[k].get("GRBM_GUI_ACTIVE", 0)):
    a = agg[k]
    f = a.get("FETCH_SIZE", 0) * 1024 * 2
    w = a.get("WRITE_SIZE", 0) * 1024
    tot_f += f
    tot_w += w
    g = a.get("GRBM_GUI_ACTIVE", 0) / 8
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024) if g else 0
    vm = (a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)) / a["SQ_INSTS_MFMA"] if a.get("SQ_INSTS_MFMA") else 0
    wc = a.get("SQ_WAVE_CYCLES", 0)
    hit = a.get("TCC_HIT_sum", 0) / (a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 1e-9))
    print("%-62s %5d %9.1f %9.1f %8.3f %8.2f %7.3f %7.3f %7.3f" % (k, cnt[k], f / 1e6, w / 1e6, busy, vm, a.get("SQ_WAIT_ANY", 0) / wc if wc else 0,
                                                            a.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0, hit))
print("all kernels: fetch %.1f MB, write %.1f MB per step" % (tot_f / steps / 1e6, tot_w / steps / 1e6))
print("mfmaBusy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); valu/mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA;")
print("waitAny / waitIns = SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES; L2hit = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)")
