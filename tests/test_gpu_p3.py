"""Three-plane ("P3") storage of fp32 activations under GV_MATH_BF16X3 (csrc/conv_x3_epi.h, csrc/conv_dma.hip).

Every fp32 value is kept as three bf16 planes whose sum is the value EXACTLY, so P3 is a storage format, not a precision
change: a convolution that reads P3 through the LDS-DMA loader issues the same six plane products in the same order as the
register-staged kernel that splits fp32 in its loader.  The tests therefore demand BITWISE equality between the two
paths (kernel by kernel and for the whole Inception plan) on top of the usual 2e-4 parity with the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvcnn_tf_amd as gv                      # noqa: E402
from gvcnn_tf_amd import _lib, backbones, p3   # noqa: E402
from oracle import backbone as OB               # noqa: E402

DEV = "cuda:0"
X3 = _lib.GV_MATH_BF16X3


def lib():
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def pack_filter(w):
    kh, kw, cin, cout = w.shape
    n = lib().gv_packed_filter_bytes(kh, kw, cin, cout, _lib.GV_F32, X3) // 4
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    wd = w.to(DEV).contiguous()
    _lib.check(lib().gv_pack_filter_hwio(wd.data_ptr(), kh, kw, cin, cout, out.data_ptr(), _lib.GV_F32, X3, st()), "pack")
    torch.cuda.synchronize()
    return out


def conv(x, w, stride, pads, out_hw, scale, shift, relu, x_p3=False, y_p3=False, split=0, y2_p3=False, relu_cols=0,
         residual=None, tile=None, x_ld=None, x_off=0, y_ld=None, y_off=0):
    """x [nb,ih,iw,cin] fp32 cpu.  Returns y (and y2 with split) as fp32 cpu tensors, whatever the storage format."""
    nb, ih, iw, cin = x.shape
    kh, kw, _, cout = w.shape
    oh, ow = out_hw
    x_ld = x_ld or cin
    n1 = split if split else cout
    n2 = cout - split
    y_ld = y_ld or n1
    xb = torch.full((nb, ih, iw, x_ld), 7.0)
    xb[..., x_off:x_off + cin] = x
    xd = (p3.to_p3(xb) if x_p3 else xb).to(DEV)
    yd = torch.full((nb, oh, ow, y_ld), -77.0)
    yd = (p3.to_p3(yd) if y_p3 else yd).to(DEV)
    y2d = None
    if split:
        y2d = torch.full((nb, oh, ow, n2), -55.0)
        y2d = (p3.to_p3(y2d) if y2_p3 else y2d).to(DEV)
    wp = pack_filter(w)
    sc, sh = scale.to(DEV), shift.to(DEV)
    rd = residual.to(DEV).contiguous() if residual is not None else None
    flags = (_lib.GV_CONV_RELU if relu else 0) | (_lib.GV_CONV_SPLIT if split else 0) | \
            (_lib.GV_CONV_X_P3 if x_p3 else 0) | (_lib.GV_CONV_Y_P3 if y_p3 else 0) | (_lib.GV_CONV_Y2_P3 if y2_p3 else 0)
    d = _lib.ConvDesc(nb, ih, iw, cin, x_ld, kh, kw, stride, pads[0], pads[1], oh, ow, cout, y_ld,
                      cout if residual is not None else 0, n2 if split else 0, flags, _lib.GV_F32, split, 0, X3, 0, relu_cols)
    if tile is not None:
        lib().gv_conv2d_set_tile_override(tile)
    try:
        rc = lib().gv_conv2d_fwd(C.byref(d), xd.data_ptr() + (6 if x_p3 else 4) * x_off, wp.data_ptr(), sc.data_ptr(),
                                 sh.data_ptr(), rd.data_ptr() if rd is not None else None,
                                 yd.data_ptr() + (6 if y_p3 else 4) * y_off, y2d.data_ptr() if split else None, None, None, st())
    finally:
        lib().gv_conv2d_set_tile_override(-1)
    _lib.check(rc, "gv_conv2d_fwd")
    torch.cuda.synchronize()
    y = (p3.from_p3(yd) if y_p3 else yd).cpu()
    if y_ld != n1:                                       # nothing outside the slice may be touched
        mask = torch.ones(y_ld, dtype=torch.bool)
        mask[y_off:y_off + n1] = False
        assert bool((y[..., mask] == -77.0).all())
    y = y[..., y_off:y_off + n1]
    if split:
        return y, (p3.from_p3(y2d) if y2_p3 else y2d).cpu()
    return y


def oracle_conv(x, w, stride, padding, scale, shift, relu):
    y = OB.conv2d(x, w, stride, padding) * scale + shift
    return torch.relu(y) if relu else y


def close(a, d, tol=2e-4):
    a, d = np.asarray(a, np.float64), np.asarray(d, np.float64)
    assert float(np.abs(a - d).max()) <= tol * max(float(np.abs(d).max()), 1e-30)


def test_p3_round_trip_is_exact():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 7, 48, generator=g) * torch.logspace(-6, 6, 48)
    assert torch.equal(p3.from_p3(p3.to_p3(x)), x)


P3_COMBOS = [((3, 3), 1, "VALID", 32, 64), ((3, 3), 1, "VALID", 80, 192), ((5, 5), 1, "SAME", 48, 64),
             ((3, 3), 2, "VALID", 96, 96), ((1, 7), 1, "SAME", 128, 128), ((7, 1), 1, "SAME", 160, 192),
             ((1, 3), 1, "SAME", 384, 384), ((3, 1), 1, "SAME", 448, 384), ((3, 3), 1, "SAME", 64, 96)]


@pytest.mark.parametrize("tile", [0, 3, 4, 7, 8, 9, 10, 12, 13, 14, 15, 16, 17, 18])
@pytest.mark.parametrize("k,stride,padding,cin,cout", P3_COMBOS)
def test_dma_conv_on_three_plane_input(k, stride, padding, cin, cout, tile):
    """The LDS-DMA kernel on P3 input (fp32 output and P3 output) against the CPU oracle and against the register-staged
    kernel on the same values as fp32 (bitwise in the same k order); input and output are channel slices of wider tensors."""
    g = torch.Generator().manual_seed(hash((k, stride, cin, cout)) % 1000)
    ih, iw = (23, 20) if cin <= 64 else (9, 10)
    x = torch.randn(3, ih, iw, cin, generator=g)
    w = torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = oracle_conv(x, w, stride, padding, scale, shift, True)
    pads = (OB.same_pads(ih, k[0], stride)[0], OB.same_pads(iw, k[1], stride)[0]) if padding == "SAME" else (0, 0)
    hw = ref.shape[1:3]
    base = conv(x, w, stride, pads, hw, scale, shift, True)                                  # fp32 in, fp32 out
    y = conv(x, w, stride, pads, hw, scale, shift, True, x_p3=True, tile=tile, x_ld=cin + 32, x_off=16, y_ld=cout + 8, y_off=4)
    close(y, ref)
    # k-tiles run channel-chunk-major here (filter taps innermost, for L2 locality) where 16 | cin: the same products in
    # another summation order, so fp32-rounding-level agreement with the tap-major register-staged kernel ...
    sc = float(base.abs().max())
    assert float((y - base).abs().max()) <= 4e-6 * sc
    yp = conv(x, w, stride, pads, hw, scale, shift, True, x_p3=True, y_p3=True, tile=tile, x_ld=cin + 32, x_off=16,
              y_ld=cout + 32, y_off=16)
    assert torch.equal(yp, y)
    # ... and BITWISE agreement in the tap-major order (debug bit 128)
    lib().gv_conv2d_set_debug(128)
    try:
        yt = conv(x, w, stride, pads, hw, scale, shift, True, x_p3=True, tile=tile, x_ld=cin + 32, x_off=16, y_ld=cout + 8, y_off=4)
    finally:
        lib().gv_conv2d_set_debug(0)
    assert torch.equal(yt, base)


WS_X3_COMBOS = [((1, 7), 128, 128, (17, 17)), ((7, 1), 160, 192, (17, 17)), ((3, 3), 64, 96, (23, 20)), ((5, 5), 48, 64, (12, 35)),
                ((1, 3), 384, 384, (8, 8)), ((3, 1), 448, 384, (5, 9)), ((3, 3), 32, 200, (9, 40)), ((1, 7), 16, 8, (3, 70))]


def ws_x3_tiles():
    n = lib().gv_conv2d_num_tile_cfgs(-3)
    return list(range(n - 5, n))


@pytest.mark.parametrize("k,cin,cout,hw", WS_X3_COMBOS)
def test_ws_strip_kernel_on_three_plane_input(k, cin, cout, hw):
    """csrc/conv_ws_x3.hip (loader waves + consumer waves, one LDS strip per 16-channel chunk serves every tap): every tile
    on the stride-1 same-grid layer classes of Inception-v3 (nets/inception_v3.py:226-338) against the CPU oracle and the
    register-staged kernel: maps narrower and wider than a wave's 32 pixels, tiles spanning images, ragged M and cout,
    channel-slice operands, fp32 and three-plane destinations, residual; what the kernel declines stays untouched."""
    g = torch.Generator().manual_seed(hash((k, cin, cout)) % 1000)
    ih, iw = hw
    nb = 5 if ih * iw > 300 else 11
    x = torch.randn(nb, ih, iw, cin, generator=g)
    w = torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, True)
    pads = (OB.same_pads(ih, k[0], 1)[0], OB.same_pads(iw, k[1], 1)[0])
    base = conv(x, w, 1, pads, hw, scale, shift, True)                                       # fp32 in, fp32 out
    ran = 0
    for tile in ws_x3_tiles():
        try:
            y = conv(x, w, 1, pads, hw, scale, shift, True, x_p3=True, tile=tile, x_ld=cin + 32, x_off=16, y_ld=cout + 8, y_off=4)
        except _lib.GvError as e:
            assert e.code == _lib.GV_E_UNSUPPORTED, (tile, e)
            continue
        ran += 1
        close(y, ref)
        assert float((y - base).abs().max()) <= 4e-6 * float(base.abs().max()), tile
        if cout % 16 == 0:
            yp = conv(x, w, 1, pads, hw, scale, shift, True, x_p3=True, y_p3=True, tile=tile, x_ld=cin + 32, x_off=16,
                      y_ld=cout + 32, y_off=16)
            assert torch.equal(yp, y), tile
    assert ran >= 2, "every layer class here is served by at least two tiles"
    res = torch.randn(nb, ih, iw, cout, generator=g)
    tile = ws_x3_tiles()[1]
    yr = conv(x, w, 1, pads, hw, scale, shift, False, x_p3=True, residual=res, tile=tile)
    close(yr, OB.conv2d(x, w, 1, "SAME") * scale + shift + res)


def test_ws_strip_kernel_declines_what_it_cannot_run():
    g = torch.Generator().manual_seed(1)
    for k, stride, padding, cin in (((3, 3), 2, "VALID", 32), ((3, 3), 1, "VALID", 32), ((1, 1), 1, "SAME", 64)):
        x = torch.randn(2, 9, 9, cin, generator=g)
        w = torch.randn(k[0], k[1], cin, 32, generator=g) * 0.1
        oh = OB.conv2d(x, w, stride, padding).shape[1]
        for tile in ws_x3_tiles():
            with pytest.raises(_lib.GvError) as ei:
                conv(x, w, stride, (0, 0), (oh, oh), torch.ones(32), torch.zeros(32), True, x_p3=True, tile=tile)
            assert ei.value.code == _lib.GV_E_UNSUPPORTED


@pytest.mark.parametrize("ih,iw,pad,cout,nb", [(109, 109, 1, 64, 2), (111, 113, 0, 64, 1), (101, 97, 1, 32, 3), (30, 140, 1, 64, 2), (7, 99, 1, 64, 2),
                                               (25, 60, 1, 64, 2), (11, 30, 1, 64, 3), (25, 61, 1, 64, 1), (9, 96, 1, 64, 1), (12, 47, 1, 64, 1),
                                               (9, 121, 1, 64, 1), (8, 85, 1, 64, 2)])
def test_fp32_conv_and_max_pool_as_one_launch(ih, iw, pad, cout, nb):
    """GV_CONV_MAXPOOL3S2 on fp32 storage (three-plane math; csrc/conv_bf16s.hip, the halo kernel's 30-pixel strip form):
    Conv2d_2b_3x3 -> MaxPool_3a_3x3 (nets/inception_v3.py:111-113) in one launch equals the two launches BITWISE — pooled
    rows that straddle the kernel's four-row tiles, pooled columns in the last (partial) strip, odd and even maps, a
    channel-slice destination; maps the 16-pixel strip form serves, other channel counts and a residual are declined."""
    g = torch.Generator().manual_seed(ih * 7 + iw)
    cin = 32
    x = torch.randn(nb, ih, iw, cin, generator=g)
    w = torch.randn(3, 3, cin, cout, generator=g) * (1.0 / (9 * cin) ** 0.5)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    oh, ow = ih + 2 * pad - 2, iw + 2 * pad - 2
    ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
    xd, wp, sc, sh = x.to(DEV), pack_filter(w), scale.to(DEV), shift.to(DEV)
    y_ld, y_off = cout + 16, 8
    sp = lib().gv_conv2d_special_tile_cfg(X3)

    def desc(flags, ld):
        return _lib.ConvDesc(nb, ih, iw, cin, cin, 3, 3, 1, pad, pad, oh, ow, cout, ld, 0, 0, flags, _lib.GV_F32, 0, sp + 1, X3, 0, 0)

    full = torch.empty(nb, oh, ow, cout, device=DEV)
    _lib.check(lib().gv_conv2d_fwd(C.byref(desc(_lib.GV_CONV_RELU, cout)), xd.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                   None, full.data_ptr(), None, None, None, st()), "conv")
    two = torch.full((nb, ph, pw, y_ld), -3.0, device=DEV)
    pd = _lib.PoolDesc(nb, oh, ow, cout, cout, 3, 3, 2, 0, 0, ph, pw, y_ld, _lib.GV_POOL_MAX, _lib.GV_F32)
    _lib.check(lib().gv_pool2d_fwd(C.byref(pd), full.data_ptr(), two.data_ptr() + 4 * y_off, st()), "pool")
    one = torch.full((nb, ph, pw, y_ld), -3.0, device=DEV)
    rc = lib().gv_conv2d_fwd(C.byref(desc(_lib.GV_CONV_RELU | _lib.GV_CONV_MAXPOOL3S2, y_ld)), xd.data_ptr(), wp.data_ptr(),
                             sc.data_ptr(), sh.data_ptr(), None, one.data_ptr() + 4 * y_off, None, None, None, st())
    torch.cuda.synchronize()
    # served where the 30-pixel strip form beats the 16-pixel one (the predicate of bf16s_halo_pool_ok / fused_maxpool_ok,
    # include/gvcnn_hip.h GV_CONV_MAXPOOL3S2): ceil(ow/16)*80 > ceil(ow/30)*128 — every map wider than 96 pixels, and the
    # narrow ranges 17..30, 49..60, 65..90 (not e.g. 61..64, 91..96, 121..128); elsewhere the caller keeps the two launches
    served = -(-ow // 16) * 80 > -(-ow // 30) * 128
    if not served:
        assert rc == _lib.GV_E_UNSUPPORTED and bool((one == -3.0).all())
        return
    _lib.check(rc, "fused")
    assert torch.equal(one, two)
    ref = torch.nn.functional.max_pool2d(oracle_conv(x, w, 1, "SAME" if pad else "VALID", scale, shift, True).permute(0, 3, 1, 2), 3, 2)
    close(one[..., y_off:y_off + cout].cpu(), ref.permute(0, 2, 3, 1))


def test_fp32_fused_max_pool_declines_other_classes():
    g = torch.Generator().manual_seed(5)
    sp = lib().gv_conv2d_special_tile_cfg(X3)
    for cin, cout, k, stride, extra in ((32, 96, 3, 1, 0), (64, 64, 3, 1, 0), (32, 64, 1, 1, 0), (32, 64, 3, 2, 0),
                                        (32, 64, 3, 1, _lib.GV_CONV_X_P3)):
        ih = iw = 120
        pad = k // 2
        oh = (ih + 2 * pad - k) // stride + 1
        x = torch.randn(1, ih, iw, max(cin, 64), generator=g).to(DEV)
        w = torch.randn(k, k, cin, cout, generator=g) * 0.05
        wp = pack_filter(w)
        y = torch.full((1, (oh - 3) // 2 + 1, (oh - 3) // 2 + 1, cout), -3.0, device=DEV)
        d = _lib.ConvDesc(1, ih, iw, cin, 64 if extra else cin, k, k, stride, pad, pad, oh, oh, cout, cout, 0, 0,
                          _lib.GV_CONV_RELU | _lib.GV_CONV_MAXPOOL3S2 | extra, _lib.GV_F32, 0, sp + 1, X3, 0, 0)
        rc = lib().gv_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), torch.ones(cout, device=DEV).data_ptr(),
                                 torch.zeros(cout, device=DEV).data_ptr(), None, y.data_ptr(), None, None, None, st())
        torch.cuda.synchronize()
        assert rc == _lib.GV_E_UNSUPPORTED and bool((y == -3.0).all()), (cin, cout, k, stride)


def wsg_tiles():
    sp = lib().gv_conv2d_special_tile_cfg(X3)
    return list(range(sp + 1, lib().gv_conv2d_num_tile_cfgs(X3)))


@pytest.mark.parametrize("cin,cout,hw,nb", [(64, 64, (9, 10), 3), (192, 208, (25, 25), 2), (768, 704, (12, 12), 3), (288, 200, (7, 5), 9),
                                            (1280, 384, (5, 5), 11)])
def test_ws_gemm_mode_on_fp32_input(cin, cout, hw, nb):
    """csrc/conv_ws_x3.hip, GEMM mode: 1x1 convolutions on plain fp32 input (the fused sibling GEMMs of an Inception block,
    nets/inception_v3.py:137-204): the loader waves split fp32 into three planes on the way into LDS.  Against the CPU
    oracle and the register-staged kernel; ragged M and cout, the shortest k-loop, channel-slice operands, fp32 and
    three-plane destinations, the block's four-way split."""
    g = torch.Generator().manual_seed(cin + cout)
    ih, iw = hw
    x = torch.randn(nb, ih, iw, cin, generator=g)
    w = torch.randn(1, 1, cin, cout, generator=g) * (1.0 / cin ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, True)
    base = conv(x, w, 1, (0, 0), hw, scale, shift, True)
    tiles = wsg_tiles()
    assert len(tiles) == 2
    for tile in tiles:
        y = conv(x, w, 1, (0, 0), hw, scale, shift, True, tile=tile, x_ld=cin + 8, x_off=4, y_ld=cout + 8, y_off=4)
        close(y, ref)
        assert float((y - base).abs().max()) <= 4e-6 * float(base.abs().max()), tile
        if cout % 16 == 0:
            yp = conv(x, w, 1, (0, 0), hw, scale, shift, True, y_p3=True, tile=tile, y_ld=cout + 32, y_off=16)
            assert torch.equal(yp, y), tile
        if cout >= 128:
            split, relu_cols = 64, cout - 32
            full = OB.conv2d(x, w, 1, "SAME") * scale + shift
            want = torch.cat([torch.relu(full[..., :relu_cols]), full[..., relu_cols:]], dim=3)
            y0, s0 = conv(x, w, 1, (0, 0), hw, scale, shift, True, split=split, relu_cols=relu_cols, tile=tile)
            close(torch.cat([y0, s0], dim=3), want)
            if (cout - split) % 16 == 0:
                y1, s1 = conv(x, w, 1, (0, 0), hw, scale, shift, True, split=split, relu_cols=relu_cols, y2_p3=True, tile=tile,
                              y_ld=256, y_off=32)
                assert torch.equal(y1, y0) and torch.equal(s1, s0)


def test_ws_kernels_repeat_bitwise_under_load():
    """Run-to-run determinism of the wave-specialised kernels while another stream keeps the device busy (a DMA still in
    flight when the epilogue re-used the LDS ring once showed only inside whole plans): 20 repeats each of a GEMM-mode, a
    strip-mode and a fused conv + max pool launch, bitwise equal."""
    g = torch.Generator().manual_seed(9)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    sp = lib().gv_conv2d_special_tile_cfg(X3)
    cases = []
    x = torch.randn(24, 12, 12, 768, generator=g)
    w = torch.randn(1, 1, 768, 704, generator=g) * 0.03
    cases.append((x, w, (0, 0), (12, 12), False, sp + 1 + 0))
    cases.append((x, w, (0, 0), (12, 12), False, sp + 1 + 1))
    x7 = torch.randn(24, 12, 12, 192, generator=g)
    w7 = torch.randn(1, 7, 192, 192, generator=g) * 0.03
    for t in ws_x3_tiles()[:3]:
        cases.append((x7, w7, (0, 3), (12, 12), True, t))
    for xx, ww, pads, hw, xp3, tile in cases:
        cout = ww.shape[3]
        sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
        first = None
        for rep in range(20):
            with torch.cuda.stream(side):
                junk = junk @ junk * 1e-4
            y = conv(xx, ww, 1, pads, hw, sc, sh, True, x_p3=xp3, tile=tile)
            if first is None:
                first = y
            assert torch.equal(y, first), (tile, rep)
    torch.cuda.synchronize()


def test_ws_gemm_mode_declines_what_it_cannot_run():
    g = torch.Generator().manual_seed(2)
    for k, stride, cin in (((3, 3), 1, 64), ((1, 1), 2, 64), ((1, 1), 1, 48), ((1, 1), 1, 40)):
        x = torch.randn(2, 9, 9, cin, generator=g)
        w = torch.randn(k[0], k[1], cin, 32, generator=g) * 0.1
        oh = OB.conv2d(x, w, stride, "SAME").shape[1]
        pads = (OB.same_pads(9, k[0], stride)[0],) * 2
        for tile in wsg_tiles():
            with pytest.raises(_lib.GvError) as ei:
                conv(x, w, stride, pads, (oh, oh), torch.ones(32), torch.zeros(32), True, tile=tile)
            assert ei.value.code == _lib.GV_E_UNSUPPORTED


@pytest.mark.parametrize("tile", [0, 3, 4, 9])
def test_staged_kernel_writes_three_planes(tile):
    """fp32 input through the register-staged kernel, P3 output through the LDS-staged epilogue: the stored planes sum
    to exactly the value the fp32 epilogue stores."""
    g = torch.Generator().manual_seed(tile)
    x = torch.randn(2, 13, 11, 64, generator=g)
    w = torch.randn(1, 1, 64, 80, generator=g) * 0.1
    scale, shift = torch.rand(80, generator=g) + 0.5, torch.randn(80, generator=g) * 0.1
    base = conv(x, w, 1, (0, 0), (13, 11), scale, shift, True, tile=tile)
    close(base, oracle_conv(x, w, 1, "VALID", scale, shift, True))
    yp = conv(x, w, 1, (0, 0), (13, 11), scale, shift, True, y_p3=True, tile=tile, y_ld=112, y_off=16)
    assert torch.equal(yp, base)


@pytest.mark.parametrize("tile", [0, 3, 9])
def test_sibling_gemm_with_three_plane_scratch(tile):
    """GV_CONV_SPLIT as the Inception plan uses it: the first member's columns into an fp32 concat slice, the other
    members (and the commuted pooled branch WITHOUT ReLU: relu_cols) into a three-plane scratch tensor."""
    g = torch.Generator().manual_seed(3)
    cin, couts = 192, (64, 48, 64, 32)
    total, split = sum(couts), couts[0]
    x = torch.randn(2, 9, 10, cin, generator=g)
    w = torch.randn(1, 1, cin, total, generator=g) * 0.07
    scale, shift = torch.rand(total, generator=g) + 0.5, torch.randn(total, generator=g) * 0.1
    relu_cols = total - couts[-1]
    full = OB.conv2d(x, w, 1, "SAME") * scale + shift
    ref = torch.cat([torch.relu(full[..., :relu_cols]), full[..., relu_cols:]], dim=3)
    y0, s0 = conv(x, w, 1, (0, 0), (9, 10), scale, shift, True, split=split, relu_cols=relu_cols, tile=tile)
    close(torch.cat([y0, s0], dim=3), ref)
    y1, s1 = conv(x, w, 1, (0, 0), (9, 10), scale, shift, True, split=split, relu_cols=relu_cols, y2_p3=True, tile=tile,
                  y_ld=256, y_off=32)
    assert torch.equal(y1, y0) and torch.equal(s1, s0)


@pytest.mark.parametrize("xp3,yp3", [(True, False), (False, True), (True, True)])
@pytest.mark.parametrize("mode,k,stride,pad", [(_lib.GV_POOL_MAX, 3, 2, 0), (_lib.GV_POOL_MAX, 1, 2, 0), (_lib.GV_POOL_AVG, 3, 1, 1),
                                               (_lib.GV_POOL_AVG_RELU, 3, 1, 1)])
def test_pools_on_three_plane_tensors(mode, k, stride, pad, xp3, yp3):
    """Max / average pools reading and / or writing three-plane tensors (channel slices of wider ones) equal the fp32
    pools BITWISE: the planes carry the fp32 value exactly."""
    g = torch.Generator().manual_seed(k * 10 + stride)
    nb, h, w, c, x_ld, x_off, y_ld, y_off = 2, 13, 12, 48, 96, 32, 80, 16
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    z = torch.randn(nb, h, w, x_ld, generator=g)
    outs = []
    for xp, yp in ((False, False), (xp3, yp3)):
        zd = (p3.to_p3(z) if xp else z).to(DEV)
        y0 = torch.full((nb, oh, ow, y_ld), -9.0)
        yd = (p3.to_p3(y0) if yp else y0).to(DEV)
        d = _lib.PoolDesc(nb, h, w, c, x_ld, k, k, stride, pad, pad, oh, ow, y_ld,
                          mode | (_lib.GV_POOL_X_P3 if xp else 0) | (_lib.GV_POOL_Y_P3 if yp else 0), _lib.GV_F32)
        _lib.check(lib().gv_pool2d_fwd(C.byref(d), zd.data_ptr() + (6 if xp else 4) * x_off,
                                       yd.data_ptr() + (6 if yp else 4) * y_off, st()), "pool")
        torch.cuda.synchronize()
        y = (p3.from_p3(yd) if yp else yd).cpu()
        assert bool((y[..., :y_off] == -9.0).all()) and bool((y[..., y_off + c:] == -9.0).all())
        outs.append(y[..., y_off:y_off + c])
    assert torch.equal(outs[0], outs[1])


def test_sibling_gemm_on_three_plane_input_with_mixed_destinations():
    """The fused sibling GEMM on the LDS-DMA kernel: three-plane block input, first member into a three-plane concat
    slice, the rest (+ the commuted pooled member without ReLU) into a three-plane scratch — and the fp32 variants."""
    g = torch.Generator().manual_seed(11)
    cin, couts = 256, (64, 48, 64, 64)
    total, split = sum(couts), couts[0]
    x = torch.randn(2, 9, 10, cin, generator=g)
    w = torch.randn(1, 1, cin, total, generator=g) * 0.06
    scale, shift = torch.rand(total, generator=g) + 0.5, torch.randn(total, generator=g) * 0.1
    relu_cols = total - couts[-1]
    y0, s0 = conv(x, w, 1, (0, 0), (9, 10), scale, shift, True, split=split, relu_cols=relu_cols)
    for yp, y2p in ((True, True), (False, True), (False, False)):
        for tile in (0, 3, 8):
            y1, s1 = conv(x, w, 1, (0, 0), (9, 10), scale, shift, True, split=split, relu_cols=relu_cols, x_p3=True,
                          y_p3=yp, y2_p3=y2p, tile=tile, x_ld=cin + 16, x_off=16, y_ld=288, y_off=224)
            assert torch.equal(y1, y0) and torch.equal(s1, s0), (yp, y2p, tile)


def test_average_pool_relu_on_three_plane_input():
    g = torch.Generator().manual_seed(5)
    nb, h, w, c, ld, off = 2, 12, 11, 32, 144, 112
    z = torch.randn(nb, h, w, ld, generator=g)
    outs = []
    for xp3 in (False, True):
        zd = (p3.to_p3(z) if xp3 else z).to(DEV)
        yd = torch.full((nb, h, w, c), -1.0, device=DEV)
        d = _lib.PoolDesc(nb, h, w, c, ld, 3, 3, 1, 1, 1, h, w, c, _lib.GV_POOL_AVG_RELU | (_lib.GV_POOL_X_P3 if xp3 else 0),
                          _lib.GV_F32)
        _lib.check(lib().gv_pool2d_fwd(C.byref(d), zd.data_ptr() + (6 if xp3 else 4) * off, yd.data_ptr(), st()), "pool")
        torch.cuda.synchronize()
        outs.append(yd.cpu())
    ref = torch.relu(OB.avg_pool2d_same3(z[..., off:off + c]))
    close(outs[0], ref, 1e-5)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("taps", [("Mixed_6e", "Mixed_7c"), ("Conv2d_4a_3x3", "Mixed_5d"), ("Conv2d_2b_3x3", "Mixed_6a"),
                                  ("Mixed_6c", "Mixed_7a")])
@pytest.mark.parametrize("size", [75, 107])
def test_inception_plan_with_and_without_three_plane_intermediates(size, taps):
    """The whole Inception-v3 plan under GV_MATH_BF16X3 with three-plane conv -> conv intermediates (LDS-DMA kernels)
    equals the plan that keeps everything fp32 to fp32 rounding (2e-5 of the end point's scale after ~45 layers) — at the
    tapped (pinned) end points, several pairs of them — and both match the CPU oracle."""
    nb = 3
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(nb, size, size, 3, generator=g) - 0.5)
    ends = {}
    for use in (False, "all"):
        plan = backbones.make_plan("inception_v3", nb, size, size, torch.device(DEV), math="bf16x3", p3=use,
                                   raw_tap=taps[0], final_tap=taps[1])
        assert any(op["x"].p3 for op in plan.ops if op["kind"] == "conv") == bool(use)
        P = gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True)
        plan.bind(P)
        plan.run(x.to(DEV))
        torch.cuda.synchronize()
        ends[use] = {k: plan.view(plan.end_points[k]).clone().cpu() for k in taps}
    for k in taps:                                         # (summation order differs in the chunk-major layers: fp32 rounding)
        sc = float(ends[False][k].abs().max())
        assert float((ends["all"][k] - ends[False][k]).abs().max()) <= 2e-5 * sc, k
    _, ep = OB.inception_v3_base(x, P, taps[1])
    for k in taps:
        d = ep[k].numpy()
        np.testing.assert_allclose(ends["all"][k].numpy(), d, rtol=1e-3, atol=1e-5 * float(np.abs(d).max()))


def test_plan_rebuilds_when_the_library_declines_a_fused_max_pool(monkeypatch):
    """The builder fuses Conv2d_2b_3x3 -> MaxPool_3a_3x3 on the word of a Python copy of the kernel's predicate
    (BackbonePlan.fused_maxpool_ok) and never allocates the un-pooled tensor.  If the two ever disagree the KERNEL wins:
    make_plan launches each fused op once, and a GV_E_UNSUPPORTED answer rebuilds the plan with the two launches.  Forced
    here by a predicate that always says yes on a 61-pixel map (which the fp32 fused kernel declines): the plan that comes
    back has the pool as its own op and gives the bits of the plan built without the fusion."""
    nb, size = 2, 128                                      # Conv2d_2b: 61 x 61
    x = (torch.rand(nb, size, size, 3, generator=torch.Generator().manual_seed(3)) - 0.5).to(DEV)
    ref = backbones.make_plan("inception_v3", nb, size, size, torch.device(DEV), math="bf16x3", lanes=False, fuse_maxpool=False)
    with monkeypatch.context() as m:
        m.setattr(backbones.BackbonePlan, "fused_maxpool_ok", lambda self, *a, **k: bool(self.fuse_maxpool))   # "yes" whenever fusion is on
        plan = backbones.make_plan("inception_v3", nb, size, size, torch.device(DEV), math="bf16x3", lanes=False)
    assert not any(op.get("maxpool") for op in plan.ops if op["kind"] == "conv")
    assert [op["name"] for op in plan.ops] == [op["name"] for op in ref.ops]
    outs = []
    for p in (plan, ref):
        p.bind(gv.params.init_backbone_params(p.param_shapes(), seed=2, perturb_bn=True))
        p.run(x)
        torch.cuda.synchronize()
        outs.append(p.view(p.end_points["Mixed_7c"]).clone())
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 1e-3
