#!/usr/bin/env python3
"""Every tile configuration of one 16-bit convolution, plain and with the train-mode BatchNorm sums folded into its epilogue
(gv_conv2d_fwd_bnstats, forward and backward form): warm-repeat times per tile, so that the cost of the sums can be read per
kernel family (register-staged, LDS-DMA, wave-specialised).
    python tools/bnstats_tiles.py [KH KW CIN COUT H W NB]
GV_TILES=31,44: only those tiles; GV_DBG=2048 (no publish) | 4096 (no table adds) | 8192 (no sums): timing ablations of the
sums (conv_stats.h, ConvStats::dbg; results wrong)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gvcnn_tf_amd import _lib  # noqa: E402

args = [int(v) for v in sys.argv[1:8]] if len(sys.argv) >= 8 else [1, 7, 192, 192, 12, 12, 384]
kh, kw, cin, cout, h, w, nb = args
lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
code, td = _lib.GV_BF16, torch.bfloat16
V = 12
g = torch.Generator().manual_seed(0)
x = torch.randn(nb, h, w, cin, generator=g).to(td).to(dev)
wt = (torch.randn(kh, kw, cin, cout, generator=g) / np.sqrt(kh * kw * cin)).to(dev).contiguous()
n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, code, 0)
wp = torch.empty((n + 3) // 4, dtype=torch.int32, device=dev)
_lib.check(lib.gv_pack_filter_hwio(wt.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), code, 0, st()), "pack")
y = torch.empty(nb, h, w, cout, dtype=td, device=dev)
z = torch.randn(nb, h, w, cout, generator=g).to(td).to(dev)
ones, zeros = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
sc, sh = (torch.rand(V, cout, generator=g) + 0.5).to(dev), (torch.randn(V, cout, generator=g) * 0.3).to(dev)
acc = torch.zeros(V * cout * 2, dtype=torch.float64, device=dev)
flops = 2.0 * nb * h * w * cout * kh * kw * cin


def stats(mode):
    s = _lib.BnStats()
    s.mode, s.groups, s.nseg = mode, V, 1
    s.seg[0].c0, s.seg[0].c1, s.seg[0].acc = 0, cout, acc.data_ptr()
    if mode == _lib.GV_BN_STATS_BWD:
        s.seg[0].z, s.seg[0].z_ld, s.seg[0].scale, s.seg[0].shift = z.data_ptr(), cout, sc.data_ptr(), sh.data_ptr()
    return s


def timed(fn, iters=20):
    if fn() != 0:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("shape kh kw cin cout h w nb = %s; us per launch (warm repeats), '-' = the tile declines" % (args,))
print("%5s %10s %10s %10s" % ("tile", "plain", "+sums fwd", "+sums bwd"))
rows = []
if os.environ.get("GV_DBG"):
    lib.gv_conv2d_set_debug(int(os.environ["GV_DBG"]))
    print("debug bits", os.environ["GV_DBG"])
tiles = [int(v) for v in os.environ["GV_TILES"].split(",")] if os.environ.get("GV_TILES") else range(1, lib.gv_conv2d_num_tile_cfgs(-1) + 1)
for t in tiles:
    d = _lib.ConvDesc(nb, h, w, cin, cin, kh, kw, 1, (kh - 1) // 2, (kw - 1) // 2, h, w, cout, cout, 0, 0, 0, code, 0, t, 0, 0)
    p = timed(lambda: lib.gv_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None,
                                        y.data_ptr(), None, None, None, st()))
    sf, sb = stats(_lib.GV_BN_STATS_FWD), stats(_lib.GV_BN_STATS_BWD)
    f = timed(lambda: lib.gv_conv2d_fwd_bnstats(C.byref(d), x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None,
                                                y.data_ptr(), C.byref(sf), st()))
    b = timed(lambda: lib.gv_conv2d_fwd_bnstats(C.byref(d), x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None,
                                                y.data_ptr(), C.byref(sb), st()))
    rows.append((t, p, f, b))
    fmt = lambda v: "%10.1f" % v if v is not None else "%10s" % "-"
    print("%5d %s %s %s" % (t, fmt(p), fmt(f), fmt(b)))
for i, nm in ((1, "plain"), (2, "+sums fwd"), (3, "+sums bwd")):
    ok = [r for r in rows if r[i] is not None]
    best = min(ok, key=lambda r: r[i])
    print("best %-10s tile %2d: %.1f us = %.0f TF/s" % (nm, best[0], best[i], flops / best[i] / 1e6))
