#!/usr/bin/env python3
"""Summarise a tools/layer_times.py --tiles log: per conv layer the best register-staged tile vs the best LDS-DMA tile.
    python tools/lt_compare.py LOG [first_dma_index]"""
import re
import sys
first = int(sys.argv[2]) if len(sys.argv) > 2 else 13
tot_old = tot_new = tot_best = 0.0
for l in open(sys.argv[1]):
    m = re.match(r'\s*(\d+) conv (\S+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+([\d.]+) ms\s+([\d.]+) TF/s.*\[(.*)\] tile=(-?\d+)', l)
    if not m:
        continue
    tiles = [float(t) for t in m.group(8).split(',')]
    old = min(tiles[:first])
    new = min(tiles[first:]) if len(tiles) > first else float('inf')
    tot_old += old
    tot_new += min(new, old) if new == float('inf') else new
    tot_best += min(old, new)
    flops = 2.0 * int(m.group(3)) * int(m.group(4)) * int(m.group(5))
    print("%3s %-34s M=%7s N=%4s K=%5s old %.4f (t%2d) dma %.4f (t%2d) x%.2f  best %.0f TF/s" % (
        m.group(1), m.group(2)[-34:], m.group(3), m.group(4), m.group(5), old, tiles.index(old), new,
        tiles.index(new) if new != float('inf') else -1, old / new if new != float('inf') else 0, flops / min(old, new) / 1e9))
print("sum old-best %.3f ms, dma-best %.3f ms, best-of-both %.3f ms" % (tot_old, tot_new, tot_best))
