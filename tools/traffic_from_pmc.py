#!/usr/bin/env python3
"""HBM traffic per conv launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps S ...`: keeps only the dispatches of the last S steps (autotune and warm-up runs are
dropped), applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of the bytes of
wide coalesced reads; units are KiB) and prints bytes per conv launch.
    python tools/traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps>"""
import csv
import json
import sys


def last_steps(path, counter, steps):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # one step ends with the classifier kernel dense_f32: cut at the (steps+1)-th last one
    ends = [i for i, r in enumerate(rows) if "dense_f32" in r["Kernel_Name"]]
    start = ends[-steps - 1] + 1 if len(ends) > steps else 0
    return rows[start:]


fetch = last_steps(sys.argv[1], "FETCH_SIZE", int(sys.argv[3]))
write = last_steps(sys.argv[2], "WRITE_SIZE", int(sys.argv[3]))
steps = int(sys.argv[3])


def tot(rows, pred):
    return sum(float(r["Counter_Value"]) for r in rows if pred(r["Kernel_Name"]))


def is_conv(n):
    return "conv_igemm" in n or "conv3x3_halo" in n or "conv_stem_patch" in n or "conv_dma" in n or "conv_ws" in n or "conv_chain" in n


n_conv = sum(1 for r in fetch if is_conv(r["Kernel_Name"]))
f_conv = tot(fetch, is_conv) * 1024 * 2           # KiB -> B, x2 gfx950 FETCH_SIZE correction
w_conv = tot(write, is_conv) * 1024
f_all = tot(fetch, lambda n: True) * 1024 * 2
w_all = tot(write, lambda n: True) * 1024
print(json.dumps({"steps": steps, "conv_launches": n_conv,
                  "conv_hbm_bytes_per_launch": (f_conv + w_conv) / max(n_conv, 1),
                  "conv_fetch_bytes_per_step": f_conv / steps, "conv_write_bytes_per_step": w_conv / steps,
                  "all_fetch_bytes_per_step": f_all / steps, "all_write_bytes_per_step": w_all / steps,
                  "note": "FETCH_SIZE x2 (gfx950), KiB units; last %d steps only" % steps}))
