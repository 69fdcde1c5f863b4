"""Train-mode BatchNorm sums folded into the producing launch (gv_conv2d_fwd_bnstats, csrc/conv_stats.h).

The reference gets fused batch norm from slim (nets/inception_utils.py:52-62, nets/resnet_utils.py:230) with one
graph copy — one set of batch statistics — per view (nets/model.py:129-141).  The separate passes
(gv_bn_sums_grouped_t, gv_bn_relu_bwd_sums_grouped_t) are pinned to the oracle's autograd in test_gpu_train_lp.py; here
the fused forms are held against them on the SAME stored tensors (so only the summation order differs), against an
fp64 recount of the stored tensors, and shown bitwise reproducible run to run.  (Product against product: this file is
collected AFTER the oracle-parity files, tests/conftest.py.)"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gvcnn_tf_amd import _lib                       # noqa: E402
from gvcnn_tf_amd.training import TrainGVCNN        # noqa: E402

DEV = "cuda:0"
TYPES = {"bf16": (_lib.GV_BF16, torch.bfloat16), "f16": (_lib.GV_F16, torch.float16)}


def lib():
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def pack(w, code):
    kh, kw, cin, cout = w.shape
    n = lib().gv_packed_filter_bytes(kh, kw, cin, cout, code, 0)
    out = torch.zeros((n + 3) // 4, dtype=torch.int32, device=DEV)
    wd = w.to(DEV).contiguous()
    _lib.check(lib().gv_pack_filter_hwio(wd.data_ptr(), kh, kw, cin, cout, out.data_ptr(), code, 0, st()), "pack")
    torch.cuda.synchronize()
    return out


def conv_tiles():
    """Every tile configuration of the 16-bit convolution (register-staged, strip / halo, LDS-DMA)."""
    return list(range(1, lib().gv_conv2d_num_tile_cfgs(-1) + 1))


def segments(cout, cuts):
    """[(c0, c1)] covering [0, cout) with the given interior cuts."""
    edges = [0] + list(cuts) + [cout]
    return list(zip(edges[:-1], edges[1:]))


def run_fused(ty, nb, hw, cin, cout, k, pad, V, tile_cfg, segs, mode, seed=0, residual=False, holes=()):
    """One convolution through gv_conv2d_fwd_bnstats.  Returns (rc, y, [acc per segment], context for the reference)."""
    code, td = TYPES[ty]
    h, w = hw
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(nb, h, w, cin, generator=g)).to(td)
    wt = (torch.randn(k, k, cin, cout, generator=g) / np.sqrt(k * k * cin)).to(td).float()
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    xd = x.to(DEV)
    wp = pack(wt, code)
    yd = torch.full((nb, oh, ow, cout), -7.0, dtype=td, device=DEV)
    res = (torch.randn(nb, oh, ow, cout, generator=g)).to(td).to(DEV) if residual else None
    ones, zeros = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    stt = _lib.BnStats()
    stt.mode, stt.groups, stt.nseg = mode, V, len(segs)
    accs, zs, scs, shs = [], [], [], []
    for i, (c0, c1) in enumerate(segs):
        cs = c1 - c0
        acc = torch.zeros(V * cs * 2, dtype=torch.float64, device=DEV)
        sg = stt.seg[i]
        sg.c0, sg.c1 = c0, c1
        sg.acc = None if i in holes else acc.data_ptr()
        if mode == _lib.GV_BN_STATS_BWD:
            z = (torch.randn(nb, oh, ow, cs + 8, generator=g)).to(td).to(DEV)      # pixel stride cs + 8
            sc = (torch.rand(V, cs, generator=g) + 0.5).to(DEV)
            sh = (torch.randn(V, cs, generator=g) * 0.3).to(DEV)
            sg.z, sg.z_ld = z.data_ptr(), cs + 8
            if i % 2 == 0:                                 # every other segment without ReLU (mask = 1)
                sg.scale, sg.shift = sc.data_ptr(), sh.data_ptr()
            else:
                sc, sh = None, None
            zs.append(z), scs.append(sc), shs.append(sh)
        accs.append(acc)
    d = _lib.ConvDesc(nb, h, w, cin, cin, k, k, 1, pad, pad, oh, ow, cout, cout, cout if residual else 0, 0, 0, code, 0,
                      tile_cfg, 0, 0)
    rc = lib().gv_conv2d_fwd_bnstats(C.byref(d), xd.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(),
                                     res.data_ptr() if res is not None else None, yd.data_ptr(), C.byref(stt), st())
    torch.cuda.synchronize()
    return rc, yd, accs, dict(x=xd, wp=wp, d=d, ones=ones, zeros=zeros, res=res, zs=zs, scs=scs, shs=shs, oh=oh, ow=ow)


def reference_sums(ty, yd, segs, V, mode, ctx, holes=()):
    """The same sums from the STORED tensor, in fp64 on the host."""
    nb, oh, ow, cout = yd.shape
    y = yd.double().cpu()
    grp = torch.arange(nb) % V
    out = []
    for i, (c0, c1) in enumerate(segs):
        cs = c1 - c0
        acc = torch.zeros(V, cs, 2, dtype=torch.float64)
        if i not in holes:
            for v in range(V):
                yy = y[grp == v][..., c0:c1].reshape(-1, cs)
                if mode == _lib.GV_BN_STATS_FWD:
                    acc[v, :, 0] = yy.sum(0)
                    acc[v, :, 1] = (yy * yy).sum(0)
                else:
                    z = ctx["zs"][i].double().cpu()[grp == v][..., :cs].reshape(-1, cs)
                    if ctx["scs"][i] is not None:
                        sc, sh = ctx["scs"][i][v].cpu(), ctx["shs"][i][v].cpu()
                        # the kernel's mask is an fp32 fma on the 16-bit z: the sign of the exact z*scale + shift
                        m = (z * sc.double() + sh.double() > 0).double()
                    else:
                        m = torch.ones_like(z)
                    gg = yy * m
                    acc[v, :, 0] = gg.sum(0)
                    acc[v, :, 1] = (gg * z).sum(0)
        out.append(acc.reshape(-1))
    return out


def check_sums(accs, refs, tol=2e-6):
    for a, r in zip(accs, refs):
        a, r = a.cpu().numpy(), r.numpy()
        scale = max(float(np.abs(r).max()), 1e-30)
        err = float(np.abs(a - r).max())
        assert err <= tol * scale, "sums differ: %.3e of scale %.3e" % (err, scale)


# geometry classes: images larger than a row tile (two images per tile at most), images of a few dozen pixels (several
# per tile: the slot table), images smaller than a wave's 32 rows
GEOS = [((19, 23), 24, 1, 0, 16), ((7, 9), 24, 3, 1, 4), ((5, 5), 24, 1, 0, 0), ((12, 12), 24, 1, 0, 8)]


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("hw,nb,k,pad,min_folded", GEOS)
@pytest.mark.parametrize("mode", [_lib.GV_BN_STATS_FWD, _lib.GV_BN_STATS_BWD])
def test_fused_sums_on_every_tile_configuration(ty, hw, nb, k, pad, min_folded, mode):
    """Every tile configuration either folds the sums — then they equal the sums of the tensor it stored, and that tensor
    equals the plain launch's bit for bit — or says GV_E_UNSUPPORTED and writes nothing."""
    V, cin, cout = 6, 32, 104
    segs = segments(cout, (32, 40, 96))                    # four BatchNorm layers' columns, one without an accumulator
    folded = 0
    for t in conv_tiles():
        rc, yd, accs, ctx = run_fused(ty, nb, hw, cin, cout, k, pad, V, t, segs, mode, holes=(1,))
        if rc == _lib.GV_E_UNSUPPORTED:
            assert float(yd.float().min()) == -7.0 and all(float(a.abs().max()) == 0.0 for a in accs)
            continue
        _lib.check(rc, "gv_conv2d_fwd_bnstats tile %d" % t)
        folded += 1
        plain = torch.full_like(yd, -3.0)
        _lib.check(lib().gv_conv2d_fwd(C.byref(ctx["d"]), ctx["x"].data_ptr(), ctx["wp"].data_ptr(), ctx["ones"].data_ptr(),
                                       ctx["zeros"].data_ptr(), None, plain.data_ptr(), None, None, None, st()), "plain")
        torch.cuda.synchronize()
        assert torch.equal(plain, yd), "tile %d: the fused launch stores a different tensor" % t
        check_sums(accs, reference_sums(ty, yd, segs, V, mode, ctx, holes=(1,)))
        assert float(accs[1].abs().max()) == 0.0           # the segment without an accumulator
    assert folded >= min_folded, "only %d tile configurations fold the sums" % folded


@pytest.mark.parametrize("mode", [_lib.GV_BN_STATS_FWD, _lib.GV_BN_STATS_BWD])
def test_fused_sums_with_a_residual_and_at_the_tail_of_the_batch(mode):
    """The accumulate form of a data gradient (residual = the gradient so far) and a pixel count that is no multiple of
    any tile height: the sums are those of the final stored values, rows past the end contribute nothing."""
    V, cin, cout = 3, 64, 64
    for t in (1, 2, 7, 14, 17):
        rc, yd, accs, ctx = run_fused("bf16", 9, (11, 13), cin, cout, 1, 0, V, t, [(0, cout)], mode, residual=True, seed=3)
        if rc == _lib.GV_E_UNSUPPORTED:
            continue
        _lib.check(rc, "fused")
        check_sums(accs, reference_sums("bf16", yd, [(0, cout)], V, mode, ctx))


def test_fused_sums_are_bitwise_reproducible_and_match_the_separate_pass():
    """Run to run the accumulators are IDENTICAL (fixed-point partials, exact fp64 additions: no dependence on the order
    the workgroups arrive in), and they agree with gv_bn_sums_grouped_t on the stored tensor to summation order."""
    V, nb, hw, cin, cout = 4, 32, (27, 31), 64, 96
    first = None
    for rep in range(4):
        rc, yd, accs, ctx = run_fused("bf16", nb, hw, cin, cout, 3, 1, V, 1, [(0, cout)], _lib.GV_BN_STATS_FWD, seed=5)
        _lib.check(rc, "fused")
        if first is None:
            first = accs[0].clone()
            sep = torch.zeros_like(first)
            _lib.check(lib().gv_bn_sums_grouped_t(yd.data_ptr(), nb, hw[0] * hw[1], cout, cout, V, sep.data_ptr(),
                                                  _lib.GV_BF16, st()), "sums")
            torch.cuda.synchronize()
            a, b = first.cpu().numpy(), sep.cpu().numpy()
            assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
        else:
            assert torch.equal(accs[0], first), "the fused sums changed between two runs of the same launch"


def test_raw_z_accumulators_give_the_same_gradients():
    """gv_bn_relu_bwd_apply_grouped_t with GV_ACCUM_RAW_Z (sum g*z in the accumulator) against the plain call (sum g*zhat):
    dz, dbeta and dgamma agree to rounding."""
    dt, tdt = _lib.GV_BF16, torch.bfloat16
    g = torch.Generator().manual_seed(1)
    N, V, h, w, c = 4, 3, 9, 7, 40
    nb, hwp = N * V, h * w
    z = (torch.randn(nb, h, w, c, generator=g) * 1.5 + 0.7).to(tdt).to(DEV)
    dy = torch.randn(nb, h, w, c, generator=g).to(tdt).to(DEV)
    beta, gamma = torch.randn(c, generator=g).to(DEV), (torch.rand(c, generator=g) + 0.5).to(DEV)
    counts = torch.full((V,), N * hwp, dtype=torch.int32, device=DEV)
    acc = torch.zeros(2 * V * c, dtype=torch.float64, device=DEV)
    stt = {k: torch.empty(V, c, device=DEV) for k in ("mean", "var", "inv", "scale", "shift")}
    y = torch.empty_like(z)
    _lib.check(lib().gv_bn_sums_grouped_t(z.data_ptr(), nb, hwp, c, c, V, acc.data_ptr(), dt, st()), "sums")
    _lib.check(lib().gv_bn_finalize_apply_grouped_t(acc.data_ptr(), counts.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3,
                                                    z.data_ptr(), nb, hwp, c, c, V, 1, y.data_ptr(), c, stt["mean"].data_ptr(),
                                                    stt["var"].data_ptr(), stt["inv"].data_ptr(), stt["scale"].data_ptr(),
                                                    stt["shift"].data_ptr(), dt, st()), "apply")
    accb = torch.zeros_like(acc)
    _lib.check(lib().gv_bn_relu_bwd_sums_grouped_t(dy.data_ptr(), c, None, c, z.data_ptr(), c, stt["mean"].data_ptr(),
                                                   stt["inv"].data_ptr(), nb, hwp, c, V, accb.data_ptr(),
                                                   stt["scale"].data_ptr(), stt["shift"].data_ptr(), dt, st()), "bwd sums")
    # the raw accumulator the fused data gradient would have produced: sum g, sum g*z
    zf, dyf = z.double(), dy.double()
    grp = (torch.arange(nb, device=DEV) % V)
    mask = (zf * stt["scale"][grp][:, None, None, :].double() + stt["shift"][grp][:, None, None, :].double() > 0).double()
    gg = dyf * mask
    raw = torch.zeros(V, c, 2, dtype=torch.float64, device=DEV)
    for v in range(V):
        raw[v, :, 0] = gg[grp == v].reshape(-1, c).sum(0)
        raw[v, :, 1] = (gg * zf)[grp == v].reshape(-1, c).sum(0)
    outs = []
    for a, flag in ((accb, 0), (raw.reshape(-1).contiguous(), _lib.GV_ACCUM_RAW_Z)):
        dz = torch.empty_like(z)
        dbeta, dgamma = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        _lib.check(lib().gv_bn_relu_bwd_apply_grouped_t(dy.data_ptr(), c, None, c, z.data_ptr(), c, stt["mean"].data_ptr(),
                                                        stt["inv"].data_ptr(), gamma.data_ptr(), counts.data_ptr(), nb, hwp, c,
                                                        V, a.data_ptr(), dz.data_ptr(), c, dbeta.data_ptr(), dgamma.data_ptr(),
                                                        stt["scale"].data_ptr(), stt["shift"].data_ptr(), 0, dt | flag, st()),
                   "bwd apply")
        torch.cuda.synchronize()
        outs.append((dz.float().cpu().numpy(), dbeta.cpu().numpy(), dgamma.cpu().numpy()))
    (dz0, db0, dg0), (dz1, db1, dg1) = outs
    assert np.abs(db0 - db1).max() <= 1e-6 * np.abs(db0).max()
    assert np.abs(dg0 - dg1).max() <= 1e-5 * np.abs(dg0).max()
    assert np.abs(dz0 - dz1).max() <= 2.0 ** -7 * np.abs(dz0).max()          # one bf16 rounding of dz at most
    assert (dz0 != dz1).mean() < 0.01


@pytest.mark.parametrize("backbone,size,min_f,min_b", [("inception_v3", 139, 0.5, 0.3), ("resnet_v2_50", 96, 0.4, 0.2)])
def test_engine_with_folded_sums_reproduces_the_separate_passes(backbone, size, min_f, min_b):
    """The whole bf16 training step with the sums folded into the producing launches against the same engine running the
    separate sums passes (the form pinned to the oracle in test_gpu_train_lp.py): same loss, same statistics to
    summation order, gradients within the rounding the two forms may differ by; and every BatchNorm sum of the folded
    step — forward and backward, hence every activation and activation gradient — and, since the filter gradients add
    their pixel slices in slice order (gv_conv2d_wgrad_ws), EVERY gradient is bitwise reproducible run to run."""
    N, V = 4, 3
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(N, V, size, size, 3, generator=g) - 0.5).to(DEV)
    labels = torch.randint(0, 10, (N,), generator=g).to(DEV)
    res = {}
    for fuse in (False, True, True):
        eng = TrainGVCNN(backbone, N, V, size, size, 10, 5, device=DEV, num_bins=5, storage="bf16", seed=4)
        eng.fuse_bn_stats = fuse
        eng.forward(x, labels, check=False)
        eng.backward()
        torch.cuda.synchronize()
        bns = [op for op in eng.plan.ops if op["kind"] == "bn"]
        convs = [op for op in eng.plan.ops if op["kind"] == "conv"]
        out = dict(loss=float(eng.loss), flat=eng._flat_g.clone(), means=[op["stat"]["mean"].clone() for op in bns],
                   vars=[op["stat"]["var"].clone() for op in bns],
                   nf=sum(1 for op in convs if op.get("_st_f_done")), nb=sum(1 for op in convs if op.get("_st_b_done")),
                   accf=eng._accum_f.clone(), accb=eng._accum_b.clone())
        if fuse and True in res:
            a = res[True]
            assert out["loss"] == a["loss"], "the folded step is not reproducible"
            assert torch.equal(out["accf"], a["accf"]) and torch.equal(out["accb"], a["accb"])
            assert torch.equal(out["flat"], a["flat"])
        res[fuse] = out
    a, b = res[False], res[True]
    assert a["nf"] == 0 and a["nb"] == 0
    # (maps of fewer pixels than a wave's 32 / 64 rows keep the separate pass: the last blocks at this small input size)
    assert b["nf"] >= min_f * len(convs), "only %d of %d convolutions folded forward sums" % (b["nf"], len(convs))
    assert b["nb"] >= min_b * len(convs), "only %d of %d data gradients folded backward sums" % (b["nb"], len(convs))
    # The two forms differ by the summation order of the statistics (1e-7 relative): the first layers agree to that; a
    # flipped bf16 rounding then grows layer by layer through the train-mode statistics (DESIGN: ~9 % per layer on a
    # randomly initialised network), so the deep layers are held loosely and the tight statement is the per-launch one.
    for li, (ma, mb, va, vb) in enumerate(zip(a["means"], b["means"], a["vars"], b["vars"])):
        tol = 1e-5 if li < 3 else 0.3
        assert float((ma - mb).abs().max()) <= tol * max(float(ma.abs().max()), 1e-3), li
        assert float((va - vb).abs().max()) <= tol * max(float(va.abs().max()), 1e-3), li
    assert abs(a["loss"] - b["loss"]) <= 3e-2 * abs(a["loss"])
    ga, gb = a["flat"].double(), b["flat"].double()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    print("folded vs separate sums: loss %.6f / %.6f, cosine of the filter gradients %.4f" % (a["loss"], b["loss"], cos))
    assert cos > 0.2, cos                                 # (coarse: same direction; see the note above)


# ---- the strip kernels of the stem (conv3x3_halo_lp, conv_stem_patch_lp): whole-image strips, sums leave once per workgroup
@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,hw,nb,mode", [(32, 32, (23, 41), 6, _lib.GV_BN_STATS_FWD), (32, 64, (23, 41), 6, _lib.GV_BN_STATS_FWD),
                                                 (32, 64, (6, 70), 5, _lib.GV_BN_STATS_FWD), (32, 32, (23, 41), 6, _lib.GV_BN_STATS_BWD),
                                                 (64, 32, (19, 37), 6, _lib.GV_BN_STATS_BWD)])
def test_fused_sums_in_the_halo_kernel(ty, cin, cout, hw, nb, mode):
    """Conv2d_2a / 2b (3x3, 32 input channels) run on the halo-tiled strip kernel; it folds the FORWARD sums per image
    strip.  Same checks as the tile kernels: the stored tensor equals the plain launch's, the sums are those of the stored
    tensor.  The backward sums are declined (measured slower than the separate pass: csrc/conv_lp.hip launch_halo)."""
    special = lib().gv_conv2d_special_tile_cfg(-1) + 1
    V = 3
    rc, yd, accs, ctx = run_fused(ty, nb, hw, cin, cout, 3, 1, V, special, [(0, cout)], mode, seed=9)
    if mode == _lib.GV_BN_STATS_BWD:
        assert rc == _lib.GV_E_UNSUPPORTED and float(yd.float().min()) == -7.0 and float(accs[0].abs().max()) == 0.0
        return
    _lib.check(rc, "halo kernel with folded sums")
    plain = torch.full_like(yd, -3.0)
    _lib.check(lib().gv_conv2d_fwd(C.byref(ctx["d"]), ctx["x"].data_ptr(), ctx["wp"].data_ptr(), ctx["ones"].data_ptr(),
                                   ctx["zeros"].data_ptr(), None, plain.data_ptr(), None, None, None, st()), "plain")
    torch.cuda.synchronize()
    assert torch.equal(plain, yd)
    check_sums(accs, reference_sums(ty, yd, [(0, cout)], V, mode, ctx))
    # two segments: the strip kernels decline (they keep one accumulator set per lane)
    rc2, yd2, accs2, _ = run_fused(ty, nb, hw, cin, cout, 3, 1, V, special, [(0, 16), (16, cout)], mode, seed=9)
    assert rc2 == _lib.GV_E_UNSUPPORTED and float(yd2.float().min()) == -7.0


@pytest.mark.parametrize("ty", ["bf16", "f16"])
def test_fused_sums_in_the_stem_strip_kernel(ty):
    """Conv2d_1a 3x3/2 on the fp32 images (GV_CONV_X_F32) with the forward sums folded in."""
    code, td = TYPES[ty]
    special = lib().gv_conv2d_special_tile_cfg(-1) + 1
    g = torch.Generator().manual_seed(2)
    nb, h, w, cout, V = 5, 47, 75, 32, 4
    x = (torch.rand(nb, h, w, 3, generator=g) - 0.5).to(DEV)
    wt = (torch.randn(3, 3, 3, cout, generator=g) * 0.3).to(td).float()
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    wp = pack(wt, code)
    ones, zeros = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    outs = []
    for fused in (True, False):
        yd = torch.full((nb, oh, ow, cout), -7.0, dtype=td, device=DEV)
        acc = torch.zeros(V * cout * 2, dtype=torch.float64, device=DEV)
        d = _lib.ConvDesc(nb, h, w, 3, 3, 3, 3, 2, 0, 0, oh, ow, cout, cout, 0, 0, _lib.GV_CONV_X_F32, code, 0, special, 0, 0)
        if fused:
            stt = _lib.BnStats()
            stt.mode, stt.groups, stt.nseg = _lib.GV_BN_STATS_FWD, V, 1
            stt.seg[0].c0, stt.seg[0].c1, stt.seg[0].acc = 0, cout, acc.data_ptr()
            _lib.check(lib().gv_conv2d_fwd_bnstats(C.byref(d), x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(),
                                                   None, yd.data_ptr(), C.byref(stt), st()), "stem strip + sums")
        else:
            _lib.check(lib().gv_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None,
                                           yd.data_ptr(), None, None, None, st()), "stem strip")
        torch.cuda.synchronize()
        outs.append((yd, acc))
    assert torch.equal(outs[0][0], outs[1][0])
    check_sums([outs[0][1]], reference_sums(ty, outs[0][0], [(0, cout)], V, _lib.GV_BN_STATS_FWD, {}))
