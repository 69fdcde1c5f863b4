"""Parity of each HIP kernel (called through the C ABI) against the CPU oracle.
Tolerance for the fp32 float path: 1e-3 (BASELINE.json north_star), tightened to 2e-4 where the
arithmetic is a plain fp32 reduction; integer / index results are bit-exact."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvcnn_tf_amd as gv                      # noqa: E402
from gvcnn_tf_amd import _lib                   # noqa: E402
from oracle import backbone as OB               # noqa: E402
from oracle import grouping as OG               # noqa: E402
from oracle import naive                        # noqa: E402

DEV = "cuda:0"
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grouping_golden.json")))


def lib():
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def pack_filter(w_hwio, math=0):
    kh, kw, cin, cout = w_hwio.shape
    n = lib().gv_packed_filter_bytes(kh, kw, cin, cout, _lib.GV_F32, math) // 4
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    wd = w_hwio.to(DEV).contiguous()
    _lib.check(lib().gv_pack_filter_hwio(wd.data_ptr(), kh, kw, cin, cout, out.data_ptr(), _lib.GV_F32, math,
                                         st()), "pack")
    torch.cuda.synchronize()
    return out


def run_conv(x, w, stride, pads, out_hw, scale, shift, relu, residual=None, y_ld=None, y_off=0,
             x_ld=None, x_off=0, second=None, tile=None, math=0):
    """x [nb,ih,iw,cin] cpu tensor.  Returns y [nb,oh,ow,cout] (and y2) as numpy."""
    nb, ih, iw, cin = x.shape
    kh, kw, _, cout = w.shape
    oh, ow = out_hw
    x_ld = x_ld or cin
    y_ld = y_ld or cout
    xb = torch.full((nb, ih, iw, x_ld), 7.0)          # poison the unused channels
    xb[..., x_off:x_off + cin] = x
    xd = xb.to(DEV)
    yd = torch.full((nb, oh, ow, y_ld), -77.0, device=DEV)
    y2d = torch.full((nb, oh, ow, cout), -55.0, device=DEV) if second else None
    wp = pack_filter(w, math)
    sc, sh = scale.to(DEV), shift.to(DEV)
    sc2 = second[0].to(DEV) if second else None          # keep the device copies alive over the launch
    sh2 = second[1].to(DEV) if second else None
    rd = residual.to(DEV).contiguous() if residual is not None else None
    flags = (_lib.GV_CONV_RELU if relu else 0) | (_lib.GV_CONV_RELU2 if second else 0)
    d = _lib.ConvDesc(nb, ih, iw, cin, x_ld, kh, kw, stride, pads[0], pads[1], oh, ow, cout, y_ld,
                      cout if residual is not None else 0, cout if second else 0, flags, _lib.GV_F32, 0, 0, math)
    if tile is not None:
        lib().gv_conv2d_set_tile_override(tile)
    try:
        rc = lib().gv_conv2d_fwd(C.byref(d), xd.data_ptr() + 4 * x_off, wp.data_ptr(), sc.data_ptr(),
                                 sh.data_ptr(), rd.data_ptr() if rd is not None else None,
                                 yd.data_ptr() + 4 * y_off, y2d.data_ptr() if second else None,
                                 sc2.data_ptr() if second else None,
                                 sh2.data_ptr() if second else None, st())
    finally:
        lib().gv_conv2d_set_tile_override(-1)
    _lib.check(rc, "gv_conv2d_fwd")
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    if y_ld != cout:                                   # nothing outside the slice may be touched
        mask = np.ones(y_ld, bool)
        mask[y_off:y_off + cout] = False
        assert (y[..., mask] == -77.0).all()
    y = y[..., y_off:y_off + cout]
    return (y, y2d.cpu().numpy()) if second else y


def oracle_conv(x, w, stride, padding, scale, shift, relu, residual=None):
    y = OB.conv2d(x, w, stride, padding) * scale + shift
    if residual is not None:
        y = y + residual
    return (torch.relu(y) if relu else y)


def tf_pads(size, k, stride, padding):
    if padding == "SAME":
        return OB.same_pads(size, k, stride)[0]
    if padding == "VALID":
        return 0
    return None


# every (kernel, stride, padding) combination on the path, cin multiple of 16 and cin=3
COMBOS = [
    ((3, 3), 2, "VALID", 3, 32), ((3, 3), 1, "VALID", 32, 32), ((3, 3), 1, "SAME", 32, 64),
    ((1, 1), 1, "SAME", 64, 80), ((3, 3), 1, "VALID", 80, 192), ((1, 1), 1, "SAME", 192, 48),
    ((5, 5), 1, "SAME", 48, 64), ((3, 3), 2, "VALID", 96, 96), ((1, 7), 1, "SAME", 128, 128),
    ((7, 1), 1, "SAME", 160, 192), ((1, 3), 1, "SAME", 384, 384), ((3, 1), 1, "SAME", 448, 384),
    ((1, 1), 2, "VALID", 256, 512), ((7, 7), 2, (3, 3, 3, 3), 3, 64), ((3, 3), 2, (1, 1, 1, 1), 64, 64),
    ((1, 1), 1, "SAME", 2048, 320),
]


@pytest.mark.parametrize("k,stride,padding,cin,cout", COMBOS)
def test_conv_combos_vs_oracle(k, stride, padding, cin, cout):
    g = torch.Generator().manual_seed(hash((k, stride, cin, cout)) % 1000)
    ih, iw = (23, 20) if cin <= 64 else (9, 10)
    x = torch.randn(3, ih, iw, cin, generator=g)
    w = torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = oracle_conv(x, w, stride, padding, scale, shift, True)
    if isinstance(padding, str):
        pads = (tf_pads(ih, k[0], stride, padding), tf_pads(iw, k[1], stride, padding))
    else:
        pads = (padding[0], padding[2])
    y = run_conv(x, w, stride, pads, ref.shape[1:3], scale, shift, True)
    np.testing.assert_allclose(y, ref.numpy(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("tile", list(range(12)))
@pytest.mark.parametrize("cout", [32, 48, 80, 192, 200])
def test_conv_every_tile_config(tile, cout):
    """All tile shapes give the same answer, incl. ragged M and N not a multiple of 32."""
    g = torch.Generator().manual_seed(tile * 100 + cout)
    x = torch.randn(2, 13, 11, 32, generator=g)                     # M = 286: ragged for every BM
    w = torch.randn(3, 3, 32, cout, generator=g) * 0.06
    scale, shift = torch.ones(cout), torch.zeros(cout)
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, False)
    y = run_conv(x, w, 1, (1, 1), (13, 11), scale, shift, False, tile=tile)
    np.testing.assert_allclose(y, ref.numpy(), rtol=2e-4, atol=2e-4)
    # fp32 MFMA sums k in the same order under every configuration: bitwise identical results
    y0 = run_conv(x, w, 1, (1, 1), (13, 11), scale, shift, False, tile=1)
    np.testing.assert_array_equal(y, y0)


@pytest.mark.parametrize("tile", [1, 2, 7, 8, 10])
@pytest.mark.parametrize("cin", [48, 80, 96])
def test_conv_deep_ktile_tap_straddle(tile, cin):
    """32-deep k-tiles hold two 16-channel chunks that may sit on different filter taps (cin=48, 80)
    and K that is not a multiple of 32 leaves a zero chunk at the end."""
    g = torch.Generator().manual_seed(tile * 10 + cin)
    x = torch.randn(2, 10, 9, cin, generator=g)
    w = torch.randn(5, 5, cin, 64, generator=g) * 0.03
    scale, shift = torch.ones(64), torch.zeros(64)
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, True)
    y = run_conv(x, w, 1, (2, 2), (10, 9), scale, shift, True, tile=tile)
    np.testing.assert_allclose(y, ref.numpy(), rtol=2e-4, atol=2e-4)


def test_conv_split_output_fused_siblings():
    """GV_CONV_SPLIT: three sibling 1x1 convs as one GEMM; first goes to a concat slice, the rest to
    a scratch tensor (nets/inception_v3.py:140-146)."""
    g = torch.Generator().manual_seed(21)
    nb, h, wd, cin = 2, 7, 9, 64
    couts = [64, 48, 96]
    total = sum(couts)
    x = torch.randn(nb, h, wd, cin, generator=g)
    ws = [torch.randn(1, 1, cin, c, generator=g) * 0.1 for c in couts]
    scale = torch.rand(total, generator=g) + 0.5
    shift = torch.randn(total, generator=g) * 0.1
    wcat = torch.cat(ws, dim=3)
    ref = oracle_conv(x, wcat, 1, "SAME", scale, shift, True).numpy()
    xd = x.to(DEV)
    wp = pack_filter(wcat)
    # packing the three filters separately and back to back gives the same packed image
    parts = torch.cat([pack_filter(wi) for wi in ws])
    assert torch.equal(parts, wp)
    yd = torch.full((nb, h, wd, 256), -7.0, device=DEV)
    y2d = torch.full((nb, h, wd, total - couts[0]), -5.0, device=DEV)
    sc, sh = scale.to(DEV), shift.to(DEV)
    d = _lib.ConvDesc(nb, h, wd, cin, cin, 1, 1, 1, 0, 0, h, wd, total, 256, 0, total - couts[0],
                      _lib.GV_CONV_RELU | _lib.GV_CONV_SPLIT, _lib.GV_F32, couts[0], 0, 0)
    _lib.check(lib().gv_conv2d_fwd(C.byref(d), xd.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                   None, yd.data_ptr() + 4 * 32, y2d.data_ptr(), None, None, st()), "conv split")
    torch.cuda.synchronize()
    y, y2 = yd.cpu().numpy(), y2d.cpu().numpy()
    np.testing.assert_allclose(y[..., 32:32 + 64], ref[..., :64], rtol=2e-4, atol=2e-4)
    assert (y[..., :32] == -7.0).all() and (y[..., 96:] == -7.0).all()
    np.testing.assert_allclose(y2, ref[..., 64:], rtol=2e-4, atol=2e-4)


def test_conv_naive_c_crosscheck():
    """Second CPU implementation (direct loops in C, double accumulation) as arbiter."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 12, 12, 16, generator=g)
    w = torch.randn(3, 3, 16, 40, generator=g) * 0.1
    b = torch.randn(40, generator=g)
    y = run_conv(x, w, 2, (0, 0), (5, 5), torch.ones(40), b, False)
    yn = naive.conv2d_nhwc(x.numpy(), w.numpy(), 2, (0, 0), (5, 5), b.numpy())
    np.testing.assert_allclose(y, yn, rtol=1e-4, atol=1e-4)


def test_conv_concat_slice_residual_and_second_output():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 9, 9, 32, generator=g)
    w = torch.randn(1, 1, 32, 64, generator=g) * 0.2
    scale, shift = torch.ones(64), torch.randn(64, generator=g)
    res = torch.randn(2, 9, 9, 64, generator=g)
    s2, h2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    pre = oracle_conv(x, w, 1, "SAME", scale, shift, False, residual=res)
    # input is a channel slice [16:48) of an 80-wide buffer; output lands at [96:160) of a 192-wide one
    y, y2 = run_conv(x, w, 1, (0, 0), (9, 9), scale, shift, False, residual=res, y_ld=192, y_off=96,
                     x_ld=80, x_off=16, second=(s2, h2))
    np.testing.assert_allclose(y, pre.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(y2, torch.relu(pre * s2 + h2).numpy(), rtol=2e-4, atol=2e-4)


def test_conv_exact_small_integers():
    """fp32 MFMA is an exact fmaf chain: integer-valued data must come out exact (layout check with
    an asymmetric filter, cf. the A=I / asymmetric-B rule)."""
    g = torch.Generator().manual_seed(11)
    x = torch.randint(-3, 4, (1, 8, 8, 16), generator=g).float()
    w = torch.randint(-2, 3, (3, 3, 16, 96), generator=g).float()
    ref = oracle_conv(x, w, 1, "SAME", torch.ones(96), torch.zeros(96), False)
    y = run_conv(x, w, 1, (1, 1), (8, 8), torch.ones(96), torch.zeros(96), False)
    np.testing.assert_array_equal(y, ref.numpy())


@pytest.mark.parametrize("k,stride,padding,mode", [(3, 2, "VALID", "max"), (3, 2, "SAME", "max"),
                                                    (3, 1, "SAME", "avg"), (1, 2, "VALID", "max")])
@pytest.mark.parametrize("shape", [(2, 14, 14, 64), (3, 9, 7, 6), (1, 5, 5, 2048), (2, 23, 17, 32), (1, 35, 36, 8)])
def test_pool_vs_oracle(k, stride, padding, mode, shape):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g)
    if mode == "max":
        ref = OB.max_pool2d(x, k, stride, padding)
    else:
        ref = OB.avg_pool2d_same3(x)
    nb, ih, iw, c = shape
    oh, ow = ref.shape[1:3]
    pt = OB.same_pads(ih, k, stride)[0] if padding == "SAME" else 0
    pl = OB.same_pads(iw, k, stride)[0] if padding == "SAME" else 0
    y_ld = c + 8
    xd = x.to(DEV)
    yd = torch.full((nb, oh, ow, y_ld), -9.0, device=DEV)
    d = _lib.PoolDesc(nb, ih, iw, c, c, k, k, stride, pt, pl, oh, ow, y_ld,
                      _lib.GV_POOL_MAX if mode == "max" else _lib.GV_POOL_AVG, _lib.GV_F32)
    _lib.check(lib().gv_pool2d_fwd(C.byref(d), xd.data_ptr(), yd.data_ptr() + 16, st()), "pool")
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    assert (y[..., :4] == -9.0).all() and (y[..., 4 + c:] == -9.0).all()
    if mode == "max":
        np.testing.assert_array_equal(y[..., 4:4 + c], ref.numpy())          # max is rounding free
    else:
        np.testing.assert_allclose(y[..., 4:4 + c], ref.numpy(), rtol=1e-5, atol=1e-6)


def test_scale_shift_act_and_gap():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 7, 7, 64, generator=g)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    xd, yd = x.to(DEV), torch.empty(3, 7, 7, 64, device=DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    _lib.check(lib().gv_scale_shift_act(xd.data_ptr(), 3 * 49, 64, 64, scd.data_ptr(),
                                        shd.data_ptr(), 1, yd.data_ptr(), 64, _lib.GV_F32, st()), "ssa")
    np.testing.assert_allclose(yd.cpu().numpy(), torch.relu(x * sc + sh).numpy(), rtol=1e-6, atol=1e-6)
    gd = torch.empty(3, 64, device=DEV)
    _lib.check(lib().gv_global_avg_pool(xd.data_ptr(), 3, 49, 64, 64, gd.data_ptr(), _lib.GV_F32, st()), "gap")
    np.testing.assert_allclose(gd.cpu().numpy(), OG.global_average_pool(x.numpy()), rtol=1e-5, atol=1e-6)


# ---- grouping module ---------------------------------------------------------------------------
def _scores(case):
    return np.array([np.frombuffer(bytes.fromhex(h), dtype=np.float32)[0]
                     for h in case["scores_f32_hex"]], dtype=np.float32)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_group_assign_bit_exact_vs_reference_golden(case):
    """The integer path against vectors produced by the reference's own nets/model.py:16-41."""
    s = torch.from_numpy(_scores(case)).to(DEV)
    if case["error"] == "IndexError":
        with pytest.raises(IndexError):
            gv.grouping_module(s, case["G"])
        with pytest.raises(IndexError):
            gv.group_scheme([_scores(case)], case["G"], case["V"])
        return
    gidx, scheme, weight = gv.grouping_module(s, case["G"])
    assert scheme.cpu().numpy().tolist() == case["scheme"]
    assert weight.cpu().numpy().tolist() == case["weight"]
    assert gidx.cpu().numpy().tolist() == OG.group_index(_scores(case)).tolist()
    # reference-shaped host functions (train.py:277-278)
    sch = gv.group_scheme([_scores(case)], case["G"], case["V"])
    assert sch.dtype == np.int64 and sch.tolist() == case["scheme"]
    w = gv.group_weight(sch)
    assert w.dtype == np.float32 and w.tolist() == case["weight"]


def test_group_assign_nan_and_bins():
    with pytest.raises(ValueError):
        gv.grouping_module(torch.tensor([0.1, float("nan")], device=DEV), 10)
    # num_bins = G generalisation: 5 equal sub-ranges
    s = torch.tensor([0.05, 0.21, 0.59, 0.61, 0.99], device=DEV)
    gidx, _, w = gv.grouping_module(s, 5, num_bins=5)
    assert gidx.tolist() == [0, 1, 2, 3, 4] and w.tolist() == [2.0] * 5


def test_kat1_through_the_hip_path():
    k = GOLD["kat1"]
    D = np.array(k["final_view_descriptors"], dtype=np.float32)
    sch = np.array(k["group_scheme"])
    views = [torch.from_numpy(D[v]).reshape(1, 1, 1, 4).to(DEV) for v in range(5)]
    gd = gv.view_pooling(views, sch)
    for g, exp in k["model_py_group_max"].items():
        assert gd[int(g)].reshape(-1).tolist() == exp
    w = gv.group_weight(sch)
    assert w.tolist() == k["model_py_weight"]
    S = gv.group_fusion(gd, w)
    np.testing.assert_allclose(S.cpu().numpy().reshape(-1), k["model_py_shape_descriptor"], rtol=1e-6)
    # unit_test.py semantics (mean, zeros) in float
    gm = gv.view_pooling(views, sch, pool="mean", empty_fill=0.0)
    np.testing.assert_allclose(gm[3].cpu().numpy().reshape(-1), [5 / 3, 31 / 3, 85.0, 31 / 3], rtol=1e-6)
    assert gm[2].reshape(-1).tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("V,N,G,shape", [(6, 2, 5, (5, 5, 64)), (12, 3, 7, (5, 5, 2048)), (20, 2, 10, (3, 3, 30)),
                                         (5, 1, 10, (1, 1, 3))])
@pytest.mark.parametrize("mode,fill", [("max", 1.0), ("mean", 0.0)])
def test_view_pool_fuse_vs_oracle(V, N, G, shape, mode, fill):
    rng = np.random.RandomState(V * 7 + N)
    F = rng.randn(V, N, *shape).astype(np.float32)
    scores = rng.uniform(0, G / 10.0 - 1e-3, size=V).astype(np.float32)
    scheme = OG.group_scheme([scores], G, V)
    weight = OG.group_weight(scheme)
    gd_o = OG.view_pooling(list(F), scheme, pool=mode, empty_fill=fill)
    S_o = OG.group_fusion(gd_o, weight)
    Fd = torch.from_numpy(F).to(DEV)
    gd = gv.view_pooling(Fd, scheme, pool=mode, empty_fill=fill)
    for g in range(G):
        if mode == "max":
            np.testing.assert_array_equal(gd[g].cpu().numpy(), gd_o[g])
        else:
            np.testing.assert_allclose(gd[g].cpu().numpy(), gd_o[g], rtol=1e-6, atol=1e-6)
    S = gv.group_fusion(gd, weight)
    np.testing.assert_allclose(S.cpu().numpy(), S_o, rtol=1e-6, atol=1e-6)
    # C arbiter
    _, S_c = naive.view_pool_fuse(F, scheme, weight, mode, fill)
    np.testing.assert_allclose(S.cpu().numpy(), S_c, rtol=1e-5, atol=1e-6)


def test_view_score_and_dense():
    g = torch.Generator().manual_seed(9)
    N, V, hw, cr = 3, 4, (4, 4), 32
    raw = torch.randn(N, V, hw[0], hw[1], cr, generator=g)
    kern = torch.randn(V, cr, generator=g) * 0.3
    bias = torch.randn(V, generator=g)
    rd = raw.reshape(N * V, 16, cr).to(DEV).contiguous()
    r_img = torch.empty(N * V, device=DEV)
    kd, bd = kern.to(DEV), bias.to(DEV)
    _lib.check(lib().gv_view_score_partial(rd.data_ptr(), N * V, 16, cr, cr, kd.data_ptr(),
                                           bd.data_ptr(), V, _lib.GV_ORDER_SHAPE_MAJOR,
                                           r_img.data_ptr(), _lib.GV_F32, st()), "score_partial")
    sc = torch.empty(V, device=DEV)
    _lib.check(lib().gv_view_score_finalize(r_img.data_ptr(), N, V, _lib.GV_ORDER_SHAPE_MAJOR,
                                            sc.data_ptr(), st()), "score_finalize")
    want = [OG.view_score(raw[:, v].numpy(), kern[v].numpy(), float(bias[v])) for v in range(V)]
    np.testing.assert_allclose(sc.cpu().numpy(), np.array(want), rtol=2e-5, atol=1e-6)
    # view-major order gives the same scores
    rd2 = raw.permute(1, 0, 2, 3, 4).reshape(N * V, 16, cr).to(DEV).contiguous()
    _lib.check(lib().gv_view_score_partial(rd2.data_ptr(), N * V, 16, cr, cr, kd.data_ptr(),
                                           bd.data_ptr(), V, _lib.GV_ORDER_VIEW_MAJOR,
                                           r_img.data_ptr(), _lib.GV_F32, st()), "score_partial")
    sc2 = torch.empty(V, device=DEV)
    _lib.check(lib().gv_view_score_finalize(r_img.data_ptr(), N, V, _lib.GV_ORDER_VIEW_MAJOR,
                                            sc2.data_ptr(), st()), "score_finalize")
    np.testing.assert_allclose(sc2.cpu().numpy(), sc.cpu().numpy(), rtol=1e-6)
    # zero response -> score exactly 0 (log 0 = -inf)
    z = torch.zeros(V, device=DEV)
    _lib.check(lib().gv_view_score_finalize(z.data_ptr(), 1, V, 0, sc.data_ptr(), st()), "score_finalize")
    assert sc.cpu().tolist() == [0.0] * V
    # dense
    x = torch.randn(5, 2048, generator=g)
    Wk, b = torch.randn(2048, 40, generator=g) * 0.02, torch.randn(40, generator=g)
    yd = torch.empty(5, 40, device=DEV)
    xdd, Wd, bdd = x.to(DEV), Wk.to(DEV), b.to(DEV)
    _lib.check(lib().gv_dense_fwd(xdd.data_ptr(), 5, 2048, Wd.data_ptr(),
                                  bdd.data_ptr(), 40, yd.data_ptr(), st()), "dense")
    np.testing.assert_allclose(yd.cpu().numpy(), OG.dense(x.numpy(), Wk.numpy(), b.numpy()), rtol=1e-4, atol=1e-4)


# ---- fp32 evaluated through bf16 planes (GV_MATH_BF16X*) ----------------------------------------
BF16S_TOL = {1: 2e-5, 2: 2e-3, 3: 3e-2}          # x3: fp32-level; x2: ~2^-16; x1: plain bf16 products


@pytest.mark.parametrize("math", [1, 2, 3])
@pytest.mark.parametrize("k,stride,padding,cin,cout", [c for c in COMBOS if c[3] in (3, 32, 48, 80, 128, 384)])
def test_conv_bf16_split_vs_oracle(math, k, stride, padding, cin, cout):
    g = torch.Generator().manual_seed(hash((k, stride, cin, cout)) % 1000)
    ih, iw = (23, 20) if cin <= 64 else (9, 10)
    x = torch.randn(3, ih, iw, cin, generator=g)
    w = torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = oracle_conv(x, w, stride, padding, scale, shift, True)
    if isinstance(padding, str):
        pads = (tf_pads(ih, k[0], stride, padding), tf_pads(iw, k[1], stride, padding))
    else:
        pads = (padding[0], padding[2])
    y = run_conv(x, w, stride, pads, ref.shape[1:3], scale, shift, True, math=math)
    tol = BF16S_TOL[math]
    np.testing.assert_allclose(y, ref.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("tile", list(range(11)))
@pytest.mark.parametrize("cout", [32, 80, 200])
def test_conv_bf16x3_every_tile_config(tile, cout):
    g = torch.Generator().manual_seed(tile * 100 + cout)
    x = torch.randn(2, 13, 11, 32, generator=g)
    w = torch.randn(3, 3, 32, cout, generator=g) * 0.06
    scale, shift = torch.ones(cout), torch.zeros(cout)
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, False)
    y = run_conv(x, w, 1, (1, 1), (13, 11), scale, shift, False, tile=tile, math=1)
    np.testing.assert_allclose(y, ref.numpy(), rtol=2e-5, atol=2e-5)


def test_conv_bf16x3_small_integers_exact_and_split_outputs():
    """Integers up to 2^8 fit one bf16 plane, products are exact in fp32: the split path must be exact."""
    g = torch.Generator().manual_seed(12)
    x = torch.randint(-3, 4, (1, 8, 8, 16), generator=g).float()
    w = torch.randint(-2, 3, (3, 3, 16, 96), generator=g).float()
    ref = oracle_conv(x, w, 1, "SAME", torch.ones(96), torch.zeros(96), False)
    y = run_conv(x, w, 1, (1, 1), (8, 8), torch.ones(96), torch.zeros(96), False, math=1)
    np.testing.assert_array_equal(y, ref.numpy())
    # residual + second output + channel-slice in/out on the split-bf16 path
    x = torch.randn(2, 9, 9, 32, generator=g)
    w = torch.randn(1, 1, 32, 64, generator=g) * 0.2
    scale, shift = torch.ones(64), torch.randn(64, generator=g)
    res = torch.randn(2, 9, 9, 64, generator=g)
    s2, h2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    pre = oracle_conv(x, w, 1, "SAME", scale, shift, False, residual=res)
    y, y2 = run_conv(x, w, 1, (0, 0), (9, 9), scale, shift, False, residual=res, y_ld=192, y_off=96,
                     x_ld=80, x_off=16, second=(s2, h2), math=1)
    np.testing.assert_allclose(y, pre.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(y2, torch.relu(pre * s2 + h2).numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("math", [0, 1])
def test_pooled_branch_commutes_with_the_1x1_conv(math):
    """inception_v3.py:152-154: relu(BN(conv1x1(avgpool(x)))) == relu(avgpool(BN(conv1x1(x)))) — the form the
    plan runs (relu_cols: trailing GEMM columns without ReLU, then GV_POOL_AVG_RELU)."""
    g = torch.Generator().manual_seed(21)
    nb, h, wd, cin, couts = 2, 9, 11, 64, (32, 48, 40)             # last = the pooled branch
    total = sum(couts)
    x = torch.randn(nb, h, wd, cin, generator=g)
    wcat = torch.randn(1, 1, cin, total, generator=g) * 0.1
    scale = torch.rand(total, generator=g) + 0.5
    shift = torch.randn(total, generator=g) * 0.1
    n_relu = couts[0] + couts[1]
    ref_plain = oracle_conv(x, wcat[..., :n_relu], 1, "SAME", scale[:n_relu], shift[:n_relu], True).numpy()
    ref_pooled = oracle_conv(OB.avg_pool2d_same3(x), wcat[..., n_relu:], 1, "SAME", scale[n_relu:], shift[n_relu:],
                             True).numpy()
    xd, wp, sc, sh = x.to(DEV), pack_filter(wcat, math), scale.to(DEV), shift.to(DEV)
    yd = torch.empty(nb, h, wd, couts[0], device=DEV)
    y2d = torch.empty(nb, h, wd, total - couts[0], device=DEV)
    d = _lib.ConvDesc(nb, h, wd, cin, cin, 1, 1, 1, 0, 0, h, wd, total, couts[0], 0, total - couts[0],
                      _lib.GV_CONV_RELU | _lib.GV_CONV_SPLIT, _lib.GV_F32, couts[0], 0, math, 0, n_relu)
    _lib.check(lib().gv_conv2d_fwd(C.byref(d), xd.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                   None, yd.data_ptr(), y2d.data_ptr(), None, None, st()), "conv")
    out = torch.empty(nb, h, wd, couts[2], device=DEV)
    pd = _lib.PoolDesc(nb, h, wd, couts[2], total - couts[0], 3, 3, 1, 1, 1, h, wd, couts[2], _lib.GV_POOL_AVG_RELU,
                       _lib.GV_F32)
    _lib.check(lib().gv_pool2d_fwd(C.byref(pd), y2d.data_ptr() + 4 * couts[1], out.data_ptr(), st()), "pool")
    torch.cuda.synchronize()
    np.testing.assert_allclose(yd.cpu().numpy(), ref_plain[..., :couts[0]], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(y2d.cpu().numpy()[..., :couts[1]], ref_plain[..., couts[0]:], rtol=2e-4, atol=2e-4)
    z = y2d.cpu().numpy()[..., couts[1]:]
    assert (z < 0).any()                                           # the pooled columns left the GEMM without ReLU
    np.testing.assert_allclose(out.cpu().numpy(), ref_pooled, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("weight_mode", [0, 1])
@pytest.mark.parametrize("pool", ["max", "mean"])
def test_per_shape_assign_and_fuse_vs_oracle(weight_mode, pool):
    """SURVEY §8 f1 kernels on random per-shape scores (every shape gets its own scheme, some groups empty,
    one shape with all-zero scores => zero weights under mean_score)."""
    rng = np.random.RandomState(11 + weight_mode)
    N, V, G, E = 5, 7, 10, 2 * 3 * 8
    scores = rng.uniform(0.0, 0.99, size=(N, V)).astype(np.float32)
    scores[3] = 0.0                                            # r == 0 -> score 0 for every view of shape 3
    F = rng.randn(N, V, E).astype(np.float32)
    sd = torch.from_numpy(scores).to(DEV)
    gidx = torch.empty(N, V, dtype=torch.int32, device=DEV)
    scheme = torch.empty(N, G, V, dtype=torch.int32, device=DEV)
    weight = torch.empty(N, G, device=DEV)
    status = torch.ones(1, dtype=torch.int32, device=DEV)
    _lib.check(lib().gv_group_assign_per_shape(sd.data_ptr(), N, V, G, 10, weight_mode, gidx.data_ptr(),
                                               scheme.data_ptr(), weight.data_ptr(), status.data_ptr(), st()), "assign")
    assert int(status.item()) == 0
    Fd = torch.from_numpy(F).to(DEV)
    S = torch.empty(N, E, device=DEV)
    D = torch.empty(G, N, E, device=DEV)
    mode = _lib.GV_VIEWPOOL_MAX if pool == "max" else _lib.GV_VIEWPOOL_MEAN
    _lib.check(lib().gv_view_pool_fuse_fwd_per_shape(Fd.data_ptr(), V, N, E, E, V * E, scheme.data_ptr(), G,
                                                     weight.data_ptr(), mode, 1.0, D.data_ptr(), S.data_ptr(),
                                                     _lib.GV_F32, st()), "fuse")
    assert len({tuple(r) for r in gidx.cpu().tolist()}) > 1
    for n in range(N):
        sch = OG.group_scheme([scores[n]], G, V)
        assert scheme[n].cpu().numpy().tolist() == sch.tolist()                    # integer path: exact
        w = OG.group_weight(sch) if weight_mode == 0 else OG.group_weight_mean_score(sch, scores[n])
        np.testing.assert_allclose(weight[n].cpu().numpy(), w, rtol=1e-6, atol=0)
        gd = OG.view_pooling([F[n:n + 1, v].reshape(1, 1, 1, E) for v in range(V)], sch, pool=pool)
        for g in range(G):
            np.testing.assert_allclose(D[g, n].cpu().numpy(), gd[g].reshape(E), rtol=1e-6, atol=1e-6)
        want = OG.group_fusion(gd, w).reshape(E) if float(w.sum()) != 0 else np.zeros(E, np.float32)
        np.testing.assert_allclose(S[n].cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    if weight_mode == 1:
        assert float(weight[3].sum()) == 0.0 and float(S[3].abs().max()) == 0.0
    # an out-of-range bin in ONE shape is reported (the reference's IndexError, model.py:23)
    sd2 = sd.clone()
    sd2[2, 4] = 0.995
    _lib.check(lib().gv_group_assign_per_shape(sd2.data_ptr(), N, V, 9, 10, weight_mode, gidx.data_ptr(),
                                               scheme.data_ptr(), weight.data_ptr(), status.data_ptr(), st()), "assign")
    assert int(status.item()) & 1


@pytest.mark.parametrize("cout,pad,hw", [(32, 0, (23, 41)), (64, 1, (23, 41)), (64, 1, (8, 32)), (48, 0, (5, 70)), (64, 1, (61, 95)),
                                         (32, 0, (3, 3)), (32, 1, (1, 1)), (32, 0, (21, 32)), (64, 1, (13, 60)), (64, 0, (9, 92))])
def test_halo_stem_kernel_fp32_storage(cout, pad, hw):
    """The halo-tiled 3x3 kernel of Conv2d_2a/2b on fp32 storage (GV_MATH_BF16X3, tile configuration 11): ragged
    strips (widths that pick the 16-pixel two-rows-per-wave form and widths that pick the 30-pixel form), VALID and SAME,
    residual, channel-slice output — fp32-level agreement with the oracle and with the implicit-GEMM kernel."""
    g = torch.Generator().manual_seed(cout + pad)
    ih, iw = hw
    x = torch.randn(3, ih, iw, 32, generator=g)
    w = torch.randn(3, 3, 32, cout, generator=g) * 0.06
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    oh, ow = ih + 2 * pad - 2, iw + 2 * pad - 2
    res = torch.randn(3, oh, ow, cout, generator=g)
    ref = oracle_conv(x, w, 1, "SAME" if pad else "VALID", scale, shift, True, residual=res).numpy()
    y = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, residual=res, tile=11, math=1, y_ld=cout + 8, y_off=4)
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)
    y0 = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, residual=res, tile=0, math=1, y_ld=cout + 8, y_off=4)
    np.testing.assert_allclose(y, y0, rtol=2e-5, atol=2e-5)
    # no residual, whole 32-channel column tiles, aligned rows: the branch-free epilogue (buffer stores)
    for relu in (True, False):
        yp = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, relu, tile=11, math=1)
        yq = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, relu, tile=0, math=1)
        np.testing.assert_allclose(yp, yq, rtol=1e-5, atol=1e-5)
    ys = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, tile=11, math=1, y_ld=cout + 32, y_off=16)
    assert np.array_equal(ys, run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, tile=11, math=1))
    # the 32x32x16 form of the wide-strip kernel (debug bit 8; the default is the 16x16x32 form) accumulates the plane
    # products in the implicit-GEMM kernel's order: BITWISE the same result
    lib().gv_conv2d_set_debug(8)
    try:
        y8 = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, tile=11, math=1)
    finally:
        lib().gv_conv2d_set_debug(0)
    assert np.array_equal(y8, run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, tile=0, math=1))


@pytest.mark.parametrize("k,pad,cout,hw", [(3, 0, 32, (47, 75)), (7, 3, 64, (47, 75)), (7, 3, 64, (224, 64)), (3, 0, 24, (9, 131))])
def test_stem_strip_kernel_fp32_storage(k, pad, cout, hw):
    """The strip kernel of the 3-channel stems on fp32 storage (GV_MATH_BF16X3, tile configuration 11): Conv2d_1a
    (3x3/2 VALID) and ResNet conv1 (7x7/2, explicit pad 3), ragged strips — fp32-level agreement with the oracle."""
    g = torch.Generator().manual_seed(k + cout)
    ih, iw = hw
    x = torch.rand(2, ih, iw, 3, generator=g) - 0.5
    w = torch.randn(k, k, 3, cout, generator=g) * 0.2
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    oh, ow = (ih + 2 * pad - k) // 2 + 1, (iw + 2 * pad - k) // 2 + 1
    xp = torch.nn.functional.pad(x, (0, 0, pad, pad, pad, pad)) if pad else x
    ref = torch.relu(OB.conv2d(xp, w, 2, "VALID") * scale + shift).numpy()
    y = run_conv(x, w, 2, (pad, pad), (oh, ow), scale, shift, True, tile=11, math=1)
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("seed", list(range(24)))
def test_halo_stem_kernel_random_geometries(seed):
    """The rolling-ring halo kernel against the implicit-GEMM kernel on random image sizes (1 .. 75, both strip forms,
    partial last tiles in both directions), paddings, output widths and batch sizes."""
    rng = np.random.RandomState(1000 + seed)
    pad = int(rng.randint(0, 2))
    ih, iw = int(rng.randint(3 - 2 * pad, 76)), int(rng.randint(3 - 2 * pad, 76))
    nb, cout = int(rng.randint(1, 4)), int(rng.choice([32, 64, 24, 48]))
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(nb, ih, iw, 32, generator=g)
    w = torch.randn(3, 3, 32, cout, generator=g) * 0.06
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    oh, ow = ih + 2 * pad - 2, iw + 2 * pad - 2
    relu = bool(rng.randint(0, 2))
    y = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, relu, tile=11, math=1)
    y0 = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, relu, tile=0, math=1)
    np.testing.assert_allclose(y, y0, rtol=1e-5, atol=1e-5)
