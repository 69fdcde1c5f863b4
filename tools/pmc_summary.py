#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if "conv_igemm" not in name and "--all" not in sys.argv:
            pass
        short = name.split("(")[0][-70:]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    print("%s  (n=%d)" % (k, n))
    for c, v in sorted(cs.items()):
        print("    %-32s mean %.4g" % (c, sum(v) / len(v)))
