#!/usr/bin/env python3
"""Random shapes through every tile of the three-plane wave-specialised kernel (csrc/conv_ws_x3.hip): strip mode on
three-plane input against the LDS-DMA kernel, GEMM mode on fp32 input against the register-staged kernel; every result is
also repeated once and must be bitwise equal to itself.   python tools/ws_x3_fuzz.py [cases] [seed] [max pixels per case]"""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib, p3  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
X3 = _lib.GV_MATH_BF16X3
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAXPIX = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
NDMA = lib.gv_conv2d_num_tile_cfgs(-3) - 5
SP = lib.gv_conv2d_special_tile_cfg(X3)


def run(d, xin, wp, sc, sh, y, tile, res=None):
    y.fill_(-7.0)
    lib.gv_conv2d_set_tile_override(tile)
    try:
        rc = lib.gv_conv2d_fwd(C.byref(d), xin.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                               res.data_ptr() if res is not None else None, y.data_ptr(), None, None, None, st)
    finally:
        lib.gv_conv2d_set_tile_override(-1)
    torch.cuda.synchronize()
    return rc


ran = declined = 0
worst = 0.0
for case in range(cases):
    gemm = rng.random() < 0.3
    kh, kw = (1, 1) if gemm else rng.choice([(1, 3), (3, 1), (3, 3), (1, 7), (7, 1), (5, 5), (1, 5), (3, 5), (2, 2)])
    cin = 16 * rng.randint(1 if not gemm else 4, 12)
    cout = 8 * rng.randint(1, 40)
    ih, iw = rng.randint(1, 40), rng.randint(1, 40)
    nb = rng.randint(1, max(1, MAXPIX // (ih * iw)))
    relu = rng.random() < 0.7
    use_res = rng.random() < 0.2
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    if (kh, kw) == (2, 2):
        pt, pl = rng.randint(0, 1), rng.randint(0, 1)
    g = torch.Generator().manual_seed(case)
    x = torch.randn(nb, ih, iw, cin, generator=g).to(dev)
    w = (torch.randn(kh, kw, cin, cout, generator=g) * (1.0 / (kh * kw * cin) ** 0.5)).to(dev)
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, _lib.GV_F32, X3) // 4
    wp = torch.empty(n, device=dev)
    _lib.check(lib.gv_pack_filter_hwio(w.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), _lib.GV_F32, X3, st), "pack")
    sc = (torch.rand(cout, generator=g) + 0.5).to(dev)
    sh = (torch.randn(cout, generator=g) * 0.1).to(dev)
    res = torch.randn(nb, ih, iw, cout, generator=g).to(dev) if use_res else None
    flags = (1 if relu else 0) | (0 if gemm else _lib.GV_CONV_X_P3)
    d = _lib.ConvDesc(nb, ih, iw, cin, cin, kh, kw, 1, pt, pl, ih, iw, cout, cout, cout if use_res else 0, 0, flags, _lib.GV_F32,
                      0, 0, X3, 0, 0)
    xin = x if gemm else p3.to_p3(x)
    ref = torch.empty(nb, ih, iw, cout, device=dev)
    _lib.check(run(d, xin, wp, sc, sh, ref, 0, res), "reference tile")
    ref = ref.clone()
    scale = float(ref.abs().max()) + 1e-30
    tiles = range(SP + 1, lib.gv_conv2d_num_tile_cfgs(X3)) if gemm else range(NDMA, NDMA + 5)
    for t in tiles:
        y = torch.empty(nb, ih, iw, cout, device=dev)
        rc = run(d, xin, wp, sc, sh, y, t, res)
        if rc == _lib.GV_E_UNSUPPORTED:
            assert bool((y == -7.0).all()), ("declined but wrote", case, t)
            declined += 1
            continue
        _lib.check(rc, "tile %d" % t)
        y1 = y.clone()
        _lib.check(run(d, xin, wp, sc, sh, y, t, res), "repeat")
        assert torch.equal(y, y1), ("not repeatable", case, t, kh, kw, cin, cout, ih, iw, nb)
        err = float((y - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 6e-6, ("mismatch", case, t, kh, kw, cin, cout, ih, iw, nb, err)
        ran += 1
print("cases %d: %d launches compared (max relative difference %.1e), %d declined" % (cases, ran, worst, declined))
