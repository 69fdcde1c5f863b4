#!/usr/bin/env python3
"""Every conv launch of a tools/seq_vs_warm.py table against its PRACTICAL bound: max(FLOPs at 1.5 PFLOP/s — what the clean
16-bit MFMA loop reaches on random data, profiles/r6_mfma_shape_ab.txt —, algorithmic bytes at 5 TB/s — what streaming kernels
reach here).  Sorted by the time a launch spends above its bound; the last line is the whole plan.
    python tools/bound_table.py profiles/r6_seq_vs_warm_c4_default.txt [--top 20]"""
import re
import sys

path = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 1000
rows = []
for l in open(path):
    m = re.match(r"(\S+)\s+(\d+)\s+(\d+)\s+(\d+) \|\s+(-?\d+) \|\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+) \|\s+(\d+) /\s+(\d+) \|\s+(\d+)", l)
    if m:
        name, M, N, K, tile, warm, seq, plus, tfw, tfs, gbs = m.groups()
        t = float(seq)
        flops = 2.0 * int(M) * int(N) * int(K)                  # (a fused launch's row shows its first member's N, K)
        flops = int(tfs) * 1e12 * t * 1e-3 if int(tfs) > 0 else flops   # the table's own rate x time: the launch's real FLOPs
        nbytes = int(gbs) * 1e9 * t * 1e-3
        bound = max(flops / 1.5e15, nbytes / 5.0e12) * 1e3
        rows.append((t - bound, t, bound, name, int(M), int(N), int(K), int(tile), int(tfs), int(gbs)))
print("%-62s %9s %9s %6s %9s | %8s %5s %5s | %4s | %6s %6s" % ("launch (in sequence)", "ms", "bound ms", "x", "above ms", "M", "N", "K", "tile", "TF/s", "GB/s"))
for gap, t, b, name, M, N, K, tile, tfs, gbs in sorted(rows, reverse=True)[:top]:
    print("%-62s %9.4f %9.4f %6.2f %9.4f | %8d %5d %5d | %4d | %6d %6d" % (name[-62:], t, b, t / b, gap, M, N, K, tile, tfs, gbs))
tt, tb = sum(r[1] for r in rows), sum(r[2] for r in rows)
print("%d conv launches: %.3f ms in sequence, sum of bounds %.3f ms: %.2f of the practical bound" % (len(rows), tt, tb, tb / tt))
