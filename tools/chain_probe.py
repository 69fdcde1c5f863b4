#!/usr/bin/env python3
"""The bottleneck chain launch alone (csrc/conv_chain.hip) at the sizes of ResNet-v2-50's blocks 1 and 2 at 384 views,
beside the two launches it replaces (gv_conv2d_fwd conv3 + gv_conv2d_fwd_xpre conv1) — warm repeats, hipEvents.
    python tools/chain_probe.py [--d 64] [--iters 20] [--only chain]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--dbg", type=int, default=0)
a = ap.parse_args()
lib = _lib.load()
lib.gv_bottleneck_chain_set_debug(a.dbg)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
code, td = _lib.GV_BF16, torch.bfloat16


def pack(w):
    kh, kw, cin, cout = w.shape
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, code, 0)
    out = torch.empty((n + 3) // 4, dtype=torch.int32, device=dev)
    wd = w.to(dev).contiguous()
    _lib.check(lib.gv_pack_filter_hwio(wd.data_ptr(), kh, kw, cin, cout, out.data_ptr(), code, 0, st), "pack")
    return out


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for d, M in ((64, 384 * 56 * 56), (128, 384 * 28 * 28)):
    if a.d and a.d != d:
        continue
    n1 = 4 * d
    g = torch.Generator().manual_seed(d)
    x = torch.relu(torch.randn(M, d, generator=g)).to(td).to(dev)
    res = torch.randn(M, n1, generator=g).to(td).to(dev)
    y = torch.empty(M, n1, dtype=td, device=dev)
    z = torch.empty(M, d, dtype=td, device=dev)
    w3p = pack(torch.randn(1, 1, d, n1, generator=g) * (1.0 / d) ** 0.5)
    w1p = pack(torch.randn(1, 1, n1, d, generator=g) * (1.0 / n1) ** 0.5)
    one, b3 = torch.ones(n1, device=dev), torch.randn(n1, generator=g).to(dev) * 0.1
    ps, ph = (torch.rand(n1, generator=g) + 0.5).to(dev), (torch.randn(n1, generator=g) * 0.2).to(dev)
    s1, h1 = (torch.rand(d, generator=g) + 0.5).to(dev), (torch.randn(d, generator=g) * 0.1).to(dev)
    desc = _lib.ChainDesc(M, d, d, n1, n1, d, code, _lib.GV_CONV_RELU2, 0)
    d3 = _lib.ConvDesc(1, M, 1, d, d, 1, 1, 1, 0, 0, M, 1, n1, n1, n1, 0, 0, code, 0, 0, 0, 0)
    d1 = _lib.ConvDesc(1, M, 1, n1, n1, 1, 1, 1, 0, 0, M, 1, d, d, 0, 0, _lib.GV_CONV_RELU, code, 0, 0, 0, 0)

    def chain():
        _lib.check(lib.gv_bottleneck_chain_fwd(C.byref(desc), x.data_ptr(), w3p.data_ptr(), one.data_ptr(), b3.data_ptr(), res.data_ptr(),
                                               y.data_ptr(), ps.data_ptr(), ph.data_ptr(), w1p.data_ptr(), s1.data_ptr(), h1.data_ptr(),
                                               z.data_ptr(), st), "chain")

    def conv3():
        _lib.check(lib.gv_conv2d_fwd(C.byref(d3), x.data_ptr(), w3p.data_ptr(), one.data_ptr(), b3.data_ptr(), res.data_ptr(),
                                     y.data_ptr(), None, None, None, st), "conv3")

    def conv1():
        _lib.check(lib.gv_conv2d_fwd_xpre(C.byref(d1), y.data_ptr(), ps.data_ptr(), ph.data_ptr(), w1p.data_ptr(), s1.data_ptr(),
                                          h1.data_ptr(), None, z.data_ptr(), None, None, None, st), "conv1")
    gb_chain = 2.0 * M * (d + n1 + n1 + d) / 1e9
    t = timed(chain, a.iters)
    line = "d %3d  M %8d | chain %.4f ms  %.0f GB/s" % (d, M, t, gb_chain / t * 1e3)
    if a.only != "chain":
        t3, t1 = timed(conv3, a.iters), timed(conv1, a.iters)
        line += " | conv3 %.4f ms (%.0f GB/s) + conv1 %.4f ms (%.0f GB/s) = %.4f ms" % (
            t3, 2.0 * M * (d + 2 * n1) / 1e9 / t3 * 1e3, t1, 2.0 * M * (n1 + d) / 1e9 / t1 * 1e3, t3 + t1)
    print(line, flush=True)
