"""gv_bottleneck_chain_fwd (csrc/conv_chain.hip): conv3 of a ResNet-v2 bottleneck unit (+ biases + shortcut,
nets/resnet_v2.py:87-91), the next unit's pre-activation (:75) and its conv1 + BatchNorm + ReLU (:83-84) as ONE launch —
against the CPU oracle at the storage type's rounding, and BIT FOR BIT against the two launches it replaces
(gv_conv2d_fwd, then gv_conv2d_fwd_xpre), directly and inside the whole ResNet-v2-50 plan."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvcnn_tf_amd as gv                      # noqa: E402
from gvcnn_tf_amd import _lib, backbones        # noqa: E402
from oracle import backbone as OB               # noqa: E402

from test_gpu_lowp import DEV, TYPES, close, lib, pack, rnd, st      # noqa: E402


def run_chain(x, w3, b3, res, ps, ph, w1, s1, h1, ty, x_ld=None, res_ld=None, y_ld=None, z_ld=None, expect=None):
    code, td, _ = TYPES[ty]
    M, d = x.shape
    n1 = 4 * d
    x_ld, res_ld, y_ld, z_ld = x_ld or d, res_ld or n1, y_ld or n1, z_ld or d

    def padded(t, ld, fill):
        b = torch.full((t.shape[0], ld), fill)
        b[:, :t.shape[1]] = t
        return b.to(td).to(DEV)
    xd, rd = padded(x, x_ld, 7.0), padded(res, res_ld, 5.0)
    yd = torch.full((M, y_ld), -77.0, dtype=td, device=DEV)
    zd = torch.full((M, z_ld), -55.0, dtype=td, device=DEV)
    w3p, w1p = pack(w3.view(1, 1, d, n1), code), pack(w1.view(1, 1, n1, d), code)
    f = lambda t: t.to(DEV).float().contiguous()
    one, b3d, psd, phd, s1d, h1d = f(torch.ones(n1)), f(b3), f(ps), f(ph), f(s1), f(h1)
    desc = _lib.ChainDesc(M, d, x_ld, res_ld, y_ld, z_ld, code, _lib.GV_CONV_RELU2, 0)
    rc = lib().gv_bottleneck_chain_fwd(C.byref(desc), xd.data_ptr(), w3p.data_ptr(), one.data_ptr(), b3d.data_ptr(), rd.data_ptr(),
                                       yd.data_ptr(), psd.data_ptr(), phd.data_ptr(), w1p.data_ptr(), s1d.data_ptr(),
                                       h1d.data_ptr(), zd.data_ptr(), st())
    torch.cuda.synchronize()
    if expect is not None:
        assert rc == expect, rc
        assert bool((yd.float() == -77.0).all()) and bool((zd.float() == -55.0).all())
        return None
    _lib.check(rc, "gv_bottleneck_chain_fwd")
    assert bool((yd[:, n1:].float() == -77.0).all()) and bool((zd[:, d:].float() == -55.0).all())     # padding untouched
    # the two launches it replaces, on the same device operands
    y2 = torch.full((M, y_ld), -77.0, dtype=td, device=DEV)
    z2 = torch.full((M, z_ld), -55.0, dtype=td, device=DEV)
    d3 = _lib.ConvDesc(1, M, 1, d, x_ld, 1, 1, 1, 0, 0, M, 1, n1, y_ld, res_ld, 0, 0, code, 0, 0, 0, 0)
    _lib.check(lib().gv_conv2d_fwd(C.byref(d3), xd.data_ptr(), w3p.data_ptr(), one.data_ptr(), b3d.data_ptr(), rd.data_ptr(),
                                   y2.data_ptr(), None, None, None, st()), "conv3")
    d1 = _lib.ConvDesc(1, M, 1, n1, y_ld, 1, 1, 1, 0, 0, M, 1, d, z_ld, 0, 0, _lib.GV_CONV_RELU, code, 0, 0, 0, 0)
    _lib.check(lib().gv_conv2d_fwd_xpre(C.byref(d1), y2.data_ptr(), psd.data_ptr(), phd.data_ptr(), w1p.data_ptr(), s1d.data_ptr(),
                                        h1d.data_ptr(), None, z2.data_ptr(), None, None, None, st()), "conv1 (xpre)")
    torch.cuda.synchronize()
    return yd[:, :n1].float().cpu(), zd[:, :d].float().cpu(), y2[:, :n1].float().cpu(), z2[:, :d].float().cpu()


def operands(M, d, td, seed):
    g = torch.Generator().manual_seed(seed)
    n1 = 4 * d
    x = rnd(torch.relu(torch.randn(M, d, generator=g)), td)                    # conv2's output is post-ReLU
    w3 = rnd(torch.randn(d, n1, generator=g) * (1.0 / d) ** 0.5, td)
    b3 = torch.randn(n1, generator=g) * 0.1
    res = rnd(torch.randn(M, n1, generator=g), td)
    ps, ph = torch.rand(n1, generator=g) + 0.5, torch.randn(n1, generator=g) * 0.2
    w1 = rnd(torch.randn(n1, d, generator=g) * (1.0 / n1) ** 0.5, td)
    s1, h1 = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    return x, w3, b3, res, ps, ph, w1, s1, h1


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("d,M", [(64, 128 * 5), (64, 1000), (64, 37), (128, 128 * 3 + 96), (128, 1)])
def test_chain_vs_oracle_and_bitwise_vs_the_two_launches(d, M, ty):
    """Whole and ragged 128-row tiles, a single row, a tile's last wave without rows; channel-slice operands (pixel strides
    wider than the tensors, padding untouched)."""
    code, td, ulp = TYPES[ty]
    x, w3, b3, res, ps, ph, w1, s1, h1 = operands(M, d, td, seed=d + M)
    y, z, y2, z2 = run_chain(x, w3, b3, res, ps, ph, w1, s1, h1, ty, x_ld=d + 8, res_ld=4 * d + 16, y_ld=4 * d + 24, z_ld=d + 8)
    assert torch.equal(y, y2), "unit output differs from gv_conv2d_fwd's"
    assert torch.equal(z, z2), "next conv1 differs from gv_conv2d_fwd_xpre's"
    # oracle: fp32 arithmetic on the rounded operands; y rounded once, the pre-activation of the ROUNDED y rounded once
    yo = x @ w3 + b3 + res
    close(y, yo.numpy(), ulp)
    pre = rnd(torch.relu(y * ps + ph), td)                  # (from the device's y: the chain's second half on its own)
    zo = torch.relu((pre @ w1) * s1 + h1)
    close(z, zo.numpy(), ulp)
    # ... and end to end against the oracle's ops with its own y (two roundings deep)
    pre_o = rnd(torch.relu(rnd(yo, td) * ps + ph), td)
    close(z, torch.relu((pre_o @ w1) * s1 + h1).numpy(), 4 * ulp, extra=4e-3)


def test_chain_declines_what_it_does_not_serve():
    U = _lib.GV_E_UNSUPPORTED
    ops = operands(64, 256, torch.bfloat16, seed=1)
    run_chain(*ops, "bf16", expect=U)                       # depth 256: the caller keeps the two launches
    ops = operands(64, 64, torch.bfloat16, seed=2)
    run_chain(*ops, "bf16", x_ld=68, expect=U)              # pixels not 16-byte aligned
    d = _lib.ChainDesc(64, 64, 64, 256, 256, 64, _lib.GV_F32, 0, 0)
    assert lib().gv_bottleneck_chain_fwd(C.byref(d), 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, None) == U
    d = _lib.ChainDesc(64, 64, 32, 256, 256, 64, _lib.GV_BF16, 0, 0)          # x_ld < d
    assert lib().gv_bottleneck_chain_fwd(C.byref(d), 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, None) == _lib.GV_E_BADARG
    assert lib().gv_bottleneck_chain_fwd(None, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, None) == _lib.GV_E_BADARG


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("M", [256 * 4, 1000, 37, 1])
def test_projection_chain_vs_oracle_and_the_launches_it_replaces(M, ty):
    """GV_CHAIN_PROJ (the first unit of ResNet-v2-50: nets/resnet_v2.py:79-81 inside :87-91): x = [conv2 output | the unit's
    pre-activation] (128 channels), the K-concatenated [conv3 ; shortcut] filter, the summed biases, no shortcut operand.
    Against the oracle at the storage type's rounding (ONE rounding of y: closer to the fp32 reference than the separate
    launches, which round the shortcut first), against those launches within that extra rounding, the second half (pre-activation
    of the ROUNDED y, conv1) bit for bit gv_conv2d_fwd_xpre on the launch's own y; whole and ragged 256-row tiles, one row;
    channel-slice operands."""
    code, td, ulp = TYPES[ty]
    d, n1 = 64, 256
    g = torch.Generator().manual_seed(M)
    x2 = rnd(torch.relu(torch.randn(M, d, generator=g)), td)                  # conv2's output
    x0 = rnd(torch.relu(torch.randn(M, d, generator=g)), td)                  # the unit's pre-activation
    w3 = rnd(torch.randn(d, n1, generator=g) * (1.0 / d) ** 0.5, td)
    ws = rnd(torch.randn(d, n1, generator=g) * (1.0 / d) ** 0.5, td)
    b3, bs = torch.randn(n1, generator=g) * 0.1, torch.randn(n1, generator=g) * 0.1
    ps, ph = torch.rand(n1, generator=g) + 0.5, torch.randn(n1, generator=g) * 0.2
    w1 = rnd(torch.randn(n1, d, generator=g) * (1.0 / n1) ** 0.5, td)
    s1, h1 = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    x_ld, y_ld, z_ld = 2 * d + 8, n1 + 24, d + 8
    xb = torch.full((M, x_ld), 7.0)
    xb[:, :d], xb[:, d:2 * d] = x2, x0
    xd = xb.to(td).to(DEV)
    yd = torch.full((M, y_ld), -77.0, dtype=td, device=DEV)
    zd = torch.full((M, z_ld), -55.0, dtype=td, device=DEV)
    wcat = torch.cat([w3, ws], dim=0)                                         # [2d, 4d]: conv3's rows, then the shortcut's
    wp, w1p = pack(wcat.view(1, 1, 2 * d, n1), code), pack(w1.view(1, 1, n1, d), code)
    f = lambda t: t.to(DEV).float().contiguous()
    one, bd, psd, phd, s1d, h1d = f(torch.ones(n1)), f(b3 + bs), f(ps), f(ph), f(s1), f(h1)
    desc = _lib.ChainDesc(M, d, x_ld, 0, y_ld, z_ld, code, _lib.GV_CONV_RELU2 | _lib.GV_CHAIN_PROJ, 0)
    _lib.check(lib().gv_bottleneck_chain_fwd(C.byref(desc), xd.data_ptr(), wp.data_ptr(), one.data_ptr(), bd.data_ptr(), None,
                                             yd.data_ptr(), psd.data_ptr(), phd.data_ptr(), w1p.data_ptr(), s1d.data_ptr(),
                                             h1d.data_ptr(), zd.data_ptr(), st()), "gv_bottleneck_chain_fwd (proj)")
    torch.cuda.synchronize()
    assert bool((yd[:, n1:].float() == -77.0).all()) and bool((zd[:, d:].float() == -55.0).all())     # padding untouched
    y, z = yd[:, :n1].float().cpu(), zd[:, :d].float().cpu()
    yo = x2 @ w3 + x0 @ ws + b3 + bs
    close(y, yo.numpy(), ulp)
    # the second half on the launch's own y: bit for bit gv_conv2d_fwd_xpre
    z2 = torch.full((M, z_ld), -55.0, dtype=td, device=DEV)
    d1 = _lib.ConvDesc(1, M, 1, n1, y_ld, 1, 1, 1, 0, 0, M, 1, d, z_ld, 0, 0, _lib.GV_CONV_RELU, code, 0, 0, 0, 0)
    _lib.check(lib().gv_conv2d_fwd_xpre(C.byref(d1), yd.data_ptr(), psd.data_ptr(), phd.data_ptr(), w1p.data_ptr(), s1d.data_ptr(),
                                        h1d.data_ptr(), None, z2.data_ptr(), None, None, None, st()), "conv1 (xpre)")
    torch.cuda.synchronize()
    assert torch.equal(zd, z2), "next conv1 differs from gv_conv2d_fwd_xpre on the same y"
    # the launches it replaces: the shortcut as its own convolution (rounded to the storage type), then conv3 + residual
    sc = torch.empty((M, n1), dtype=td, device=DEV)
    dsc = _lib.ConvDesc(1, M, 1, d, x_ld, 1, 1, 1, 0, 0, M, 1, n1, n1, 0, 0, 0, code, 0, 0, 0, 0)
    wsp, w3p = pack(ws.view(1, 1, d, n1), code), pack(w3.view(1, 1, d, n1), code)
    _lib.check(lib().gv_conv2d_fwd(C.byref(dsc), xd.data_ptr() + 2 * d, wsp.data_ptr(), one.data_ptr(), f(bs).data_ptr(), None,
                                   sc.data_ptr(), None, None, None, st()), "shortcut")
    ysep = torch.empty((M, n1), dtype=td, device=DEV)
    d3 = _lib.ConvDesc(1, M, 1, d, x_ld, 1, 1, 1, 0, 0, M, 1, n1, n1, n1, 0, 0, code, 0, 0, 0, 0)
    _lib.check(lib().gv_conv2d_fwd(C.byref(d3), xd.data_ptr(), w3p.data_ptr(), one.data_ptr(), f(b3).data_ptr(), sc.data_ptr(),
                                   ysep.data_ptr(), None, None, None, st()), "conv3")
    torch.cuda.synchronize()
    # the shortcut's own rounding is the only difference: half a step of the SHORTCUT's magnitude, which may move y's
    # rounding by one whole step of its own
    ys, scv = ysep.float().cpu().numpy(), sc.float().cpu().numpy()
    assert bool((np.abs(y.numpy() - ys) <= 1.01 * ulp * (np.abs(scv) + 2 * np.abs(ys)) + 1e-6).all())
    same = float((y.numpy() == ys).mean())
    assert same > 0.5, same


def test_projection_chain_declines_what_it_does_not_serve():
    B, U = _lib.GV_E_BADARG, _lib.GV_E_UNSUPPORTED
    P = _lib.GV_CONV_RELU2 | _lib.GV_CHAIN_PROJ
    call = lambda desc, res: lib().gv_bottleneck_chain_fwd(C.byref(desc), 16, 16, 16, 16, res, 16, 16, 16, 16, 16, 16, 16, None)
    assert call(_lib.ChainDesc(64, 64, 128, 0, 256, 64, _lib.GV_BF16, P, 0), 16) == B          # a shortcut operand AND the flag
    assert call(_lib.ChainDesc(64, 64, 64, 0, 256, 64, _lib.GV_BF16, P, 0), None) == B         # x narrower than 2d
    assert call(_lib.ChainDesc(64, 128, 256, 0, 512, 128, _lib.GV_BF16, P, 0), None) == U      # d = 128: not this form
    assert call(_lib.ChainDesc(64, 64, 128, 256, 256, 64, _lib.GV_BF16, _lib.GV_CONV_RELU2, 0), None) == B   # no flag, no shortcut


def run_unit(x4, w2, s2, h2, w3, b3, res, ps, ph, w1, s1, h1, ty, expect=None):
    """x4 [nb, ih, iw, d]: the unit's conv1 output.  Returns (y, z) of gv_bottleneck_unit_fwd and (c2, y, z) of the launches it
    replaces: gv_conv2d_fwd (conv2 3x3 + BN + ReLU), then gv_bottleneck_chain_fwd."""
    code, td, _ = TYPES[ty]
    nb, ih, iw, d = x4.shape
    M, n1 = nb * ih * iw, 4 * d
    xd, rd = x4.to(td).to(DEV).contiguous(), res.to(td).to(DEV).contiguous()
    yd = torch.full((M, n1), -77.0, dtype=td, device=DEV)
    zd = torch.full((M, d), -55.0, dtype=td, device=DEV)
    w2p, w3p, w1p = pack(w2, code), pack(w3.view(1, 1, d, n1), code), pack(w1.view(1, 1, n1, d), code)
    f = lambda t: t.to(DEV).float().contiguous()
    one, b3d, psd, phd, s1d, h1d, s2d, h2d = f(torch.ones(n1)), f(b3), f(ps), f(ph), f(s1), f(h1), f(s2), f(h2)
    desc = _lib.UnitDesc(nb, ih, iw, d, d, n1, n1, d, code, _lib.GV_CONV_RELU2, 0)
    rc = lib().gv_bottleneck_unit_fwd(C.byref(desc), xd.data_ptr(), w2p.data_ptr(), s2d.data_ptr(), h2d.data_ptr(), w3p.data_ptr(),
                                      one.data_ptr(), b3d.data_ptr(), rd.data_ptr(), yd.data_ptr(), psd.data_ptr(), phd.data_ptr(),
                                      w1p.data_ptr(), s1d.data_ptr(), h1d.data_ptr(), zd.data_ptr(), st())
    torch.cuda.synchronize()
    if expect is not None:
        assert rc == expect, rc
        return None
    _lib.check(rc, "gv_bottleneck_unit_fwd")
    c2 = torch.empty((nb, ih, iw, d), dtype=td, device=DEV)
    dc = _lib.ConvDesc(nb, ih, iw, d, d, 3, 3, 1, 1, 1, ih, iw, d, d, 0, 0, _lib.GV_CONV_RELU, code, 0, 0, 0, 0)
    _lib.check(lib().gv_conv2d_fwd(C.byref(dc), xd.data_ptr(), w2p.data_ptr(), s2d.data_ptr(), h2d.data_ptr(), None, c2.data_ptr(),
                                   None, None, None, st()), "conv2")
    y2 = torch.empty_like(yd)
    z2 = torch.empty_like(zd)
    dch = _lib.ChainDesc(M, d, d, n1, n1, d, code, _lib.GV_CONV_RELU2, 0)
    _lib.check(lib().gv_bottleneck_chain_fwd(C.byref(dch), c2.data_ptr(), w3p.data_ptr(), one.data_ptr(), b3d.data_ptr(), rd.data_ptr(),
                                             y2.data_ptr(), psd.data_ptr(), phd.data_ptr(), w1p.data_ptr(), s1d.data_ptr(),
                                             h1d.data_ptr(), z2.data_ptr(), st()), "chain")
    torch.cuda.synchronize()
    return yd.float().cpu(), zd.float().cpu(), c2.float().cpu(), y2.float().cpu(), z2.float().cpu()


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("d,nb,ih,iw", [(64, 3, 14, 14), (64, 2, 9, 23), (64, 5, 3, 3), (128, 2, 12, 10), (128, 1, 1, 5)])
def test_unit_vs_oracle_and_vs_the_launches_it_replaces(d, nb, ih, iw, ty):
    """conv2 (3x3 / 1 SAME + BN + ReLU) in front of the chain, one launch: maps narrower and wider than a wave's 32 pixels,
    waves that span images, a single row of pixels, ragged 128- / 256-row tiles.  Against the oracle's ops on the rounded
    operands, and against gv_conv2d_fwd -> gv_bottleneck_chain_fwd: conv2 sums k in another order here (tap-major), so a c2
    element may round the other way (one unit in the last place of the storage type) and that moves y / z by their own
    rounding at most — checked as: almost all elements identical, the rest within two roundings."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(d + nb * ih + iw)
    M, n1 = nb * ih * iw, 4 * d
    x4 = rnd(torch.relu(torch.randn(nb, ih, iw, d, generator=g)), td)
    w2 = rnd(torch.randn(3, 3, d, d, generator=g) * (1.0 / (9 * d)) ** 0.5, td)
    s2, h2 = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    _, w3, b3, res, ps, ph, w1, s1, h1 = operands(M, d, td, seed=d + M)
    y, z, c2, y2, z2 = run_unit(x4, w2, s2, h2, w3, b3, res, ps, ph, w1, s1, h1, ty)
    # oracle: conv2 of the reference graph on the rounded operands
    c2o = torch.relu(OB.conv2d(x4, w2, 1, "SAME") * s2 + h2)
    close(c2, c2o.numpy(), ulp)
    c2r = rnd(c2o, td).reshape(M, d)
    yo = c2r @ w3 + b3 + res
    close(y, yo.numpy(), 2 * ulp, extra=2e-3)
    pre = rnd(torch.relu(y * ps + ph), td)
    close(z, torch.relu((pre @ w1) * s1 + h1).numpy(), ulp)                 # the second half on the device's own y
    # against the separate launches
    for got, ref, what in ((y, y2, "y"), (z, z2, "z")):
        same = float((got == ref).float().mean())
        assert same > 0.97, (what, same)
        scale = float(ref.abs().max())
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=4 * ulp, atol=2e-3 * scale)


def test_unit_declines_what_it_does_not_serve():
    U = _lib.GV_E_UNSUPPORTED
    d = _lib.UnitDesc(1, 8, 8, 256, 256, 1024, 1024, 256, _lib.GV_BF16, 0, 0)
    args = [16] * 15 + [None]
    assert lib().gv_bottleneck_unit_fwd(C.byref(d), *args) == U                                   # depth 256
    d = _lib.UnitDesc(1, 8, 8, 64, 64, 256, 256, 64, _lib.GV_F32, 0, 0)
    assert lib().gv_bottleneck_unit_fwd(C.byref(d), *args) == U                                   # fp32 storage
    d = _lib.UnitDesc(1, 8, 8, 64, 60, 256, 256, 64, _lib.GV_BF16, 0, 0)
    assert lib().gv_bottleneck_unit_fwd(C.byref(d), *args) == _lib.GV_E_BADARG                    # x_ld < d
    assert lib().gv_bottleneck_unit_fwd(None, *args) == _lib.GV_E_BADARG


@pytest.mark.parametrize("ty,size,nb", [("bf16", 64, 6), ("f16", 224, 12), ("bf16", 97, 4)])
def test_resnet_plan_with_and_without_the_chain(ty, size, nb):
    """The whole 16-bit ResNet-v2-50 plan with the five chain launches of blocks 1 and 2 equals the plan with the separate
    conv3 / conv1 launches bit for bit at block3 / block4 (the taps of nets/model.py:144,149), and matches the oracle."""
    x = (torch.rand(nb, size, size, 3, generator=torch.Generator().manual_seed(size)) - 0.5)
    outs, nops = [], []
    for fuse in (True, False):
        plan = backbones.make_plan("resnet_v2_50", nb, size, size, torch.device(DEV), dtype=ty, lanes=False, fuse_chain=fuse,
                                   fuse_unit=False, fuse_proj=False)   # (the projection form rounds once less: its own test)
        P = gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True)
        plan.bind(P)
        plan.run(x.to(DEV))
        torch.cuda.synchronize()
        # (only the tapped end points are persistent: block1 / block2 buffers are re-used by later layers)
        outs.append({k: plan.view(plan.end_points[k]).clone() for k in ("resnet_v2_50/block3", "resnet_v2_50/block4")})
        nops.append(len(plan.ops))
        assert sum(1 for op in plan.ops if op.get("chain")) == (5 if fuse else 0)
        if fuse:
            plan_ops_fused = plan.ops
    assert nops[0] == nops[1] - 5 - sum(1 for op in plan_ops_fused if op.get("split"))   # (+ the conv1 + shortcut pairs)
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    _, ep = OB.resnet_v2_50(x[:1], P)
    ref = ep["resnet_v2_50/block4"][0].numpy()
    got = outs[0]["resnet_v2_50/block4"][0].float().cpu().numpy()
    rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    assert rel < (3e-2 if ty == "bf16" else 4e-3), rel


@pytest.mark.parametrize("ty,size,nb", [("bf16", 64, 6), ("f16", 224, 12), ("bf16", 97, 4)])
def test_resnet_plan_with_whole_unit_launches(ty, size, nb):
    """The default 16-bit ResNet-v2-50 plan: one launch per bottleneck unit inside blocks 1 and 2 (conv2 in front of the
    chain).  Against the oracle at the storage type's bound, and against the plan of separate launches: conv2's k order
    differs, so block3 / block4 agree to the storage rounding of the ~40 layers in between, not bit for bit."""
    x = (torch.rand(nb, size, size, 3, generator=torch.Generator().manual_seed(size)) - 0.5)
    outs = []
    for fuse in (True, False):
        plan = backbones.make_plan("resnet_v2_50", nb, size, size, torch.device(DEV), dtype=ty, lanes=False, fuse_chain=fuse,
                                   fuse_unit="all", fuse_proj=False)    # (every chain with its conv2 in front: d = 128 too)
        P = gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True)
        plan.bind(P)
        plan.run(x.to(DEV))
        torch.cuda.synchronize()
        outs.append({k: plan.view(plan.end_points[k]).float().cpu() for k in ("resnet_v2_50/block3", "resnet_v2_50/block4")})
        assert sum(1 for op in plan.ops if op.get("chain") and op["chain"].get("front")) == (5 if fuse else 0)
    bound = 3e-2 if ty == "bf16" else 4e-3
    _, ep = OB.resnet_v2_50(x[:2], P)
    for k in outs[0]:
        a, b = outs[0][k].numpy(), outs[1][k].numpy()
        assert float(np.linalg.norm(a - b) / np.linalg.norm(b)) < bound / 2, k
        ref = ep[k].numpy()
        for o in outs:
            assert float(np.linalg.norm(o[k][:2].numpy() - ref) / np.linalg.norm(ref)) < bound, k


@pytest.mark.parametrize("ty,size,nb", [("bf16", 64, 6), ("f16", 224, 12), ("bf16", 97, 4)])
def test_resnet_plan_with_the_projection_shortcut_inside_conv3(ty, size, nb):
    """The DEFAULT 16-bit ResNet-v2-50 plan: the first unit's projection shortcut is part of its conv3 GEMM (GV_CHAIN_PROJ: no
    shortcut launch, no shortcut tensor; conv2 and the pre-activation write the two halves of one 128-channel buffer).  Against
    the oracle at the storage type's bound, and against the plan with the shortcut as its own launch: one rounding fewer in
    front of ~45 layers, so block3 / block4 agree to the storage rounding, not bit for bit.  Also with the stand-alone
    pre-activation (no fused max pool: the 97-pixel map keeps conv1, pool1 and the pre-activation as three launches)."""
    x = (torch.rand(nb, size, size, 3, generator=torch.Generator().manual_seed(size)) - 0.5)
    outs = []
    for proj in (True, False):
        plan = backbones.make_plan("resnet_v2_50", nb, size, size, torch.device(DEV), dtype=ty, lanes=False, fuse_proj=proj)
        P = gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True)
        plan.bind(P)
        plan.run(x.to(DEV))
        torch.cuda.synchronize()
        outs.append({k: plan.view(plan.end_points[k]).float().cpu() for k in ("resnet_v2_50/block3", "resnet_v2_50/block4")})
        assert sum(1 for op in plan.ops if op.get("chain") and op["chain"].get("proj")) == (1 if proj else 0)
        assert any(op["name"].endswith("block1/unit_1/bottleneck_v2/shortcut") for op in plan.ops) == (not proj)
    bound = 3e-2 if ty == "bf16" else 4e-3
    _, ep = OB.resnet_v2_50(x[:2], P)
    for k in outs[0]:
        a, b = outs[0][k].numpy(), outs[1][k].numpy()
        assert float(np.linalg.norm(a - b) / np.linalg.norm(b)) < bound, k    # (two 16-bit plans, each within `bound` of the oracle)
        ref = ep[k].numpy()
        for o in outs:
            assert float(np.linalg.norm(o[k][:2].numpy() - ref) / np.linalg.norm(ref)) < bound, k


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("d,hw,nb", [(64, (9, 10), 3), (128, (7, 7), 5), (256, (14, 14), 3), (256, (3, 5), 1)])
def test_tail_form_of_conv1_behind_a_preactivation(d, hw, nb, ty):
    """gv_conv2d_fwd_xpre's class `1x1, cin = 4 * cout, BatchNorm + ReLU` (the conv1 of a bottleneck identity unit reading the
    unit input through its folded pre-activation, nets/resnet_v2.py:75,83-84) as the TAIL form of the bottleneck launch (the
    special tile index): bit for bit the register-staged kernel, within the storage rounding of the oracle; other classes are
    declined under that index."""
    from test_gpu_lowp import oracle_conv, run_conv, special_tile
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(d + hw[0])
    cin = 4 * d
    x = rnd(torch.randn(nb, hw[0], hw[1], cin, generator=g), td)
    w = rnd(torch.randn(1, 1, cin, d, generator=g) * (1.0 / cin) ** 0.5, td)
    xs, xh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    scale, shift = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    sp = special_tile() + 1
    kw = dict(xpre=(xs, xh), x_ld=cin + 8, x_off=8, y_ld=d + 8, y_off=0)
    y_ref = run_conv(x, w, 1, (0, 0), hw, scale, shift, True, ty, **kw)
    y_tail = run_conv(x, w, 1, (0, 0), hw, scale, shift, True, ty, tile_cfg=sp, **kw)
    assert np.array_equal(y_tail, y_ref)
    # (the oracle's pre-activation is a multiply and an add, the kernels' one fma: an element of it may round the other way,
    # which K = 4d products turn into ~1e-3 of the output's scale on a handful of elements)
    pre = rnd(torch.relu(x * xs + xh), td)
    close(y_tail, oracle_conv(pre, w, 1, "VALID", scale, shift, True).numpy(), ulp, extra=2e-3)
    # not this form's: no ReLU, a residual, cin != 4 * cout
    U = _lib.GV_E_UNSUPPORTED
    run_conv(x, w, 1, (0, 0), hw, scale, shift, False, ty, tile_cfg=sp, expect=U, **kw)
    res = rnd(torch.randn(nb, hw[0], hw[1], d, generator=g), td)
    run_conv(x, w, 1, (0, 0), hw, scale, shift, True, ty, tile_cfg=sp, residual=res, expect=U, **kw)
    w2 = rnd(torch.randn(1, 1, cin, 2 * d, generator=g) * 0.05, td)
    run_conv(x, w2, 1, (0, 0), hw, torch.ones(2 * d), torch.zeros(2 * d), True, ty, tile_cfg=sp, expect=U, xpre=(xs, xh))
