"""Parity of the 16-bit storage path (GV_BF16 / GV_F16: configs c3-c5) against the CPU oracle, through the
C ABI.  Inputs and filters are rounded to the storage type FIRST, so the oracle (fp32 arithmetic on those
rounded values) and the kernel (exact 16-bit products, fp32 accumulate) differ only by fp32 summation
order and by the one rounding of each output to the storage type:

    tolerance = 1 ulp of the storage type (2^-8 relative for bf16, 2^-11 for fp16) of the value,
                + 1e-4 of the tensor's scale for the summation order (north_star's fp32 1e-3 cannot apply
                to values that are themselves stored with 8 or 11 significant bits).

Max pooling and view max-pooling are rounding-free: bit-exact."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvcnn_tf_amd as gv                      # noqa: E402
from gvcnn_tf_amd import _lib                   # noqa: E402
from oracle import backbone as OB               # noqa: E402
from oracle import grouping as OG               # noqa: E402
from oracle import model as OM                  # noqa: E402

DEV = "cuda:0"
TYPES = {"bf16": (_lib.GV_BF16, torch.bfloat16, 2.0 ** -8), "f16": (_lib.GV_F16, torch.float16, 2.0 ** -11)}


def lib():
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def special_tile():
    """Index of the strip kernels (halo-tiled 3x3, 3-channel stems) among the tile configurations."""
    return lib().gv_conv2d_special_tile_cfg(-1)


def dma_tiles():
    """The LDS-DMA tile configurations (csrc/conv_dma.hip) follow the strip kernels' index."""
    return list(range(lib().gv_conv2d_special_tile_cfg(-1) + 1, lib().gv_conv2d_num_tile_cfgs(-1)))


def rnd(t, td):
    return t.to(td).to(torch.float32)


def close(actual, desired, ulp, extra=1e-4):
    actual, desired = np.asarray(actual, np.float32), np.asarray(desired, np.float32)
    scale = max(float(np.abs(desired).max()), 1e-30)
    np.testing.assert_allclose(actual, desired, rtol=1.01 * ulp, atol=extra * scale)


def pack(w, code):
    kh, kw, cin, cout = w.shape
    n = lib().gv_packed_filter_bytes(kh, kw, cin, cout, code, 0)
    assert n > 0 and n % 64 == 0 or n > 0
    out = torch.empty((n + 3) // 4, dtype=torch.int32, device=DEV)
    wd = w.to(DEV).contiguous()
    _lib.check(lib().gv_pack_filter_hwio(wd.data_ptr(), kh, kw, cin, cout, out.data_ptr(), code, 0, st()), "pack")
    torch.cuda.synchronize()
    return out


def run_conv(x, w, stride, pads, out_hw, scale, shift, relu, ty, residual=None, second=None, split=0,
             tile=None, x_f32=False, x_ld=None, x_off=0, y_ld=None, y_off=0, xpre=None, tile_cfg=0, expect=None, pooled=None,
             pooled_same=False, pool_act=None):
    """pooled = (ph, pw): GV_CONV_MAXPOOL3S2 (pooled_same: _SAME) — the destination is the pooled tensor, out_hw stays the
    convolution's; pool_act = (scale2, shift2, relu2): GV_CONV_POOL_ACT2, the pooled tensor through a second activation."""
    code, td, _ = TYPES[ty]
    nb, ih, iw, cin = x.shape
    kh, kw, _, cout = w.shape
    oh, ow = out_hw
    x_ld = x_ld or cin
    n1 = split if split else cout
    y_ld = y_ld or n1
    xb = torch.full((nb, ih, iw, x_ld), 7.0)
    xb[..., x_off:x_off + cin] = x
    xd = xb.to(DEV) if x_f32 else xb.to(td).to(DEV)
    xes = 4 if x_f32 else 2
    yd = torch.full((nb,) + (tuple(pooled) if pooled else (oh, ow)) + (y_ld,), -77.0, dtype=td, device=DEV)
    n2 = cout - split if split else cout
    y2d = torch.full((nb, oh, ow, n2), -55.0, dtype=td, device=DEV) if (second or split) else None
    wp = pack(w, code)
    sc, sh = scale.to(DEV), shift.to(DEV)
    sc2 = second[0].to(DEV) if second else None
    sh2 = second[1].to(DEV) if second else None
    if pool_act is not None:
        assert not second
        sc2 = pool_act[0].to(DEV) if pool_act[0] is not None else None
        sh2 = pool_act[1].to(DEV) if pool_act[1] is not None else None
    rd = residual.to(td).to(DEV).contiguous() if residual is not None else None
    flags = (_lib.GV_CONV_RELU if relu else 0) | (_lib.GV_CONV_RELU2 if second else 0) | \
            (_lib.GV_CONV_SPLIT if split else 0) | (_lib.GV_CONV_X_F32 if x_f32 else 0) | \
            ((_lib.GV_CONV_MAXPOOL3S2_SAME if pooled_same else _lib.GV_CONV_MAXPOOL3S2) if pooled else 0) | \
            ((_lib.GV_CONV_POOL_ACT2 | (_lib.GV_CONV_RELU2 if pool_act[2] else 0)) if pool_act is not None else 0)
    d = _lib.ConvDesc(nb, ih, iw, cin, x_ld, kh, kw, stride, pads[0], pads[1], oh, ow, cout, y_ld,
                      cout if residual is not None else 0, n2 if y2d is not None else 0, flags, code, split, tile_cfg, 0, 0)
    if xpre is not None:                    # the input is read as relu(x * xscale + xshift) (gv_conv2d_fwd_xpre)
        xs, xh = xpre[0].to(DEV), xpre[1].to(DEV)
        rc = lib().gv_conv2d_fwd_xpre(C.byref(d), xd.data_ptr() + xes * x_off, xs.data_ptr(), xh.data_ptr(), wp.data_ptr(),
                                      sc.data_ptr(), sh.data_ptr(), rd.data_ptr() if rd is not None else None,
                                      yd.data_ptr() + 2 * y_off, y2d.data_ptr() if y2d is not None else None,
                                      sc2.data_ptr() if second else None, sh2.data_ptr() if second else None, st())
        if expect is not None:
            assert rc == expect, rc
            return None
        _lib.check(rc, "gv_conv2d_fwd_xpre")
        torch.cuda.synchronize()
        y = yd.float().cpu().numpy()[..., y_off:y_off + n1]
        return (y, y2d.float().cpu().numpy()) if y2d is not None else y
    if tile is not None:
        lib().gv_conv2d_set_tile_override(tile)
    try:
        rc = lib().gv_conv2d_fwd(C.byref(d), xd.data_ptr() + xes * x_off, wp.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                 rd.data_ptr() if rd is not None else None, yd.data_ptr() + 2 * y_off,
                                 y2d.data_ptr() if y2d is not None else None,
                                 sc2.data_ptr() if sc2 is not None else None, sh2.data_ptr() if sh2 is not None else None, st())
    finally:
        lib().gv_conv2d_set_tile_override(-1)
    if expect is not None:
        assert rc == expect, rc
        torch.cuda.synchronize()
        assert (yd.float() == -77.0).all()              # nothing was written
        return None
    _lib.check(rc, "gv_conv2d_fwd")
    torch.cuda.synchronize()
    y = yd.float().cpu().numpy()
    if y_ld != n1:
        mask = np.ones(y_ld, bool)
        mask[y_off:y_off + n1] = False
        assert (y[..., mask] == -77.0).all()
    y = y[..., y_off:y_off + n1]
    return (y, y2d.float().cpu().numpy()) if y2d is not None else y


def oracle_conv(x, w, stride, padding, scale, shift, relu, residual=None):
    y = OB.conv2d(x, w, stride, padding) * scale + shift
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y


def tf_pads(size, k, stride, padding):
    return OB.same_pads(size, k, stride)[0] if padding == "SAME" else 0


COMBOS = [
    ((3, 3), 2, "VALID", 3, 32), ((3, 3), 1, "VALID", 32, 32), ((3, 3), 1, "SAME", 32, 64),
    ((1, 1), 1, "SAME", 64, 80), ((3, 3), 1, "VALID", 80, 192), ((1, 1), 1, "SAME", 192, 48),
    ((5, 5), 1, "SAME", 48, 64), ((3, 3), 2, "VALID", 96, 96), ((1, 7), 1, "SAME", 128, 128),
    ((7, 1), 1, "SAME", 160, 192), ((1, 3), 1, "SAME", 384, 384), ((3, 1), 1, "SAME", 448, 384),
    ((1, 1), 2, "VALID", 256, 512), ((7, 7), 2, (3, 3, 3, 3), 3, 64), ((3, 3), 2, (1, 1, 1, 1), 64, 64),
    ((1, 1), 1, "SAME", 2048, 320), ((1, 1), 1, "SAME", 8, 24), ((3, 3), 1, "SAME", 40, 16),
]


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("k,stride,padding,cin,cout", COMBOS)
def test_lp_conv_combos_vs_oracle(k, stride, padding, cin, cout, ty):
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(hash((k, stride, cin, cout)) % 1000)
    ih, iw = (23, 20) if cin <= 64 else (9, 10)
    x = rnd(torch.randn(3, ih, iw, cin, generator=g), td)
    w = rnd(torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5), td)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = oracle_conv(x, w, stride, padding, scale, shift, True)
    if isinstance(padding, str):
        pads = (tf_pads(ih, k[0], stride, padding), tf_pads(iw, k[1], stride, padding))
    else:
        pads = (padding[0], padding[2])
    y = run_conv(x, w, stride, pads, ref.shape[1:3], scale, shift, True, ty)
    close(y, ref.numpy(), ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("tile", list(range(12)))
@pytest.mark.parametrize("cout", [32, 48, 200])
def test_lp_conv_every_tile_config(tile, cout, ty):
    """All tile shapes, ragged M (286) and N not a multiple of 32; K = 360 is not a multiple of the k-tile."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(tile * 100 + cout)
    x = rnd(torch.randn(2, 13, 11, 40, generator=g), td)
    w = rnd(torch.randn(3, 3, 40, cout, generator=g) * 0.06, td)
    scale, shift = torch.ones(cout), torch.zeros(cout)
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, False)
    y = run_conv(x, w, 1, (1, 1), (13, 11), scale, shift, False, ty, tile=tile)
    close(y, ref.numpy(), ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("tile", list(range(13, 25)))
@pytest.mark.parametrize("cout", [32, 48, 200])
def test_lp_conv_every_dma_tile_config(tile, cout, ty):
    """The LDS-DMA loader (global_load_lds, swizzled LDS image, counted vmcnt ring): all its tile shapes on the same
    ragged problem as above (M = 286, N not a multiple of 32, K = 360 not a multiple of the k-tile, padding taps
    through the zero page)."""
    assert tile in dma_tiles()
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(tile * 100 + cout)
    x = rnd(torch.randn(2, 13, 11, 40, generator=g), td)
    w = rnd(torch.randn(3, 3, 40, cout, generator=g) * 0.06, td)
    scale, shift = torch.ones(cout), torch.zeros(cout)
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, False)
    y = run_conv(x, w, 1, (1, 1), (13, 11), scale, shift, False, ty, tile=tile)
    close(y, ref.numpy(), ulp)


DMA_COMBOS = [c for c in COMBOS if c[3] % 8 == 0]


@pytest.mark.parametrize("k,stride,padding,cin,cout", DMA_COMBOS)
def test_lp_dma_conv_combos_vs_oracle(k, stride, padding, cin, cout):
    """Every (kernel, stride, padding) combination of the two backbones whose input has whole 8-channel chunks,
    through the LDS-DMA loader (a 4-wave, an 8-wave and a 4x1-wave tile), with residual, second output / split epilogue
    on a channel-slice destination."""
    ty = "bf16"
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(hash((k, stride, cin, cout)) % 1000)
    ih, iw = (23, 20) if cin <= 64 else (9, 10)
    x = rnd(torch.randn(3, ih, iw, cin, generator=g), td)
    w = rnd(torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5), td)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    if isinstance(padding, str):
        pads = (tf_pads(ih, k[0], stride, padding), tf_pads(iw, k[1], stride, padding))
    else:
        pads = (padding[0], padding[2])
    ref0 = oracle_conv(x, w, stride, padding, scale, shift, False)
    res = rnd(torch.randn(ref0.shape, generator=g), td)
    ref = oracle_conv(x, w, stride, padding, scale, shift, True, residual=res)
    for tile_i in (0, 3, 8, 9, 22, 23):                # (every tile on the same operands and the same oracle result)
        y = run_conv(x, w, stride, pads, ref.shape[1:3], scale, shift, True, ty, residual=res, tile=dma_tiles()[tile_i],
                     x_ld=cin + 16, x_off=8, y_ld=cout + 24, y_off=16)
        close(y, ref.numpy(), ulp)


K64_COMBOS = [c for c in COMBOS if c[3] % 64 == 0] + [((3, 3), 1, "SAME", 64, 96), ((7, 1), 1, "SAME", 192, 192)]


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("k,stride,padding,cin,cout", K64_COMBOS)
def test_lp_dma_k64_tiles_vs_oracle(k, stride, padding, cin, cout, ty):
    """The 64-deep k-tile form of the LDS-DMA loader (round 4: 128-byte LDS rows = whole cache lines of the source, the
    chunk swizzle reaching the row block's parity): every such tile on every layer class with whole 64-channel chunks,
    with residual, ReLU and channel-slice operands whose pixels are only 16-byte aligned."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(hash((k, stride, cin, cout)) % 1000)
    ih, iw = (23, 20) if cin <= 64 else (9, 10)
    x = rnd(torch.randn(3, ih, iw, cin, generator=g), td)
    w = rnd(torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5), td)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    if isinstance(padding, str):
        pads = (tf_pads(ih, k[0], stride, padding), tf_pads(iw, k[1], stride, padding))
    else:
        pads = (padding[0], padding[2])
    ref0 = oracle_conv(x, w, stride, padding, scale, shift, False)
    res = rnd(torch.randn(ref0.shape, generator=g), td)
    ref = oracle_conv(x, w, stride, padding, scale, shift, True, residual=res)
    for tile_i in (16, 17, 18, 19, 20, 21, 24):        # (every tile on the same operands and the same oracle result)
        y = run_conv(x, w, stride, pads, ref.shape[1:3], scale, shift, True, ty, residual=res, tile=dma_tiles()[tile_i],
                     x_ld=cin + 16, x_off=8, y_ld=cout + 24, y_off=16)
        close(y, ref.numpy(), ulp)


def ws_tiles():
    """The wave-specialised kernel's tile configurations (csrc/conv_ws.hip) follow the 25 LDS-DMA tiles."""
    return dma_tiles()[25:]


WS_COMBOS = [
    ((1, 7), "SAME", 128, 128, (12, 12)), ((7, 1), "SAME", 160, 192, (12, 12)), ((7, 1), "SAME", 192, 200, (17, 17)),
    ((3, 3), "SAME", 64, 96, (25, 25)), ((3, 3), "SAME", 96, 48, (9, 10)), ((5, 5), "SAME", 64, 64, (13, 11)),
    ((1, 3), "SAME", 384, 384, (5, 5)), ((3, 1), "SAME", 448, 384, (8, 8)), ((1, 1), "SAME", 192, 48, (23, 20)),
    ((1, 1), "SAME", 768, 704, (12, 12)), ((3, 3), (2, 2, 2, 2), 64, 32, (21, 37)), ((1, 1), "SAME", 128, 40, (3, 7)),
]


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("k,padding,cin,cout,hw", WS_COMBOS)
def test_lp_ws_tiles_vs_oracle(k, padding, cin, cout, hw, ty):
    """The wave-specialised kernel (loader waves + MFMA consumer waves; tap re-use from one LDS strip per channel chunk):
    every tile on the stride-1 same-grid layer classes of Inception-v3 (1x7, 7x1, 3x3, 5x5, 1x3, 3x1) and on 1x1 GEMMs —
    image rows narrower and wider than a wave's 32 pixels, tiles that span several images, ragged M and cout, padding
    larger than SAME (a data gradient's), residual + ReLU, channel-slice operands whose pixels are only 16-byte aligned."""
    code, td, ulp = TYPES[ty]
    assert len(ws_tiles()) == 11
    g = torch.Generator().manual_seed(hash((k, cin, cout, hw)) % 1000)
    ih, iw = hw
    nb = 5 if ih * iw < 200 else 3
    x = rnd(torch.randn(nb, ih, iw, cin, generator=g), td)
    w = rnd(torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5), td)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    if isinstance(padding, str):
        pads = (tf_pads(ih, k[0], 1, padding), tf_pads(iw, k[1], 1, padding))
    else:
        pads = (padding[0], padding[2])
    ref0 = oracle_conv(x, w, 1, padding, scale, shift, False)
    same_grid = tuple(ref0.shape[1:3]) == (ih, iw)
    res = rnd(torch.randn(ref0.shape, generator=g), td)
    ref = oracle_conv(x, w, 1, padding, scale, shift, True, residual=res)
    ran_k64 = False
    for tile in ws_tiles():
        if not same_grid:                       # (a full-padding 3x3 grows the map: not this kernel's class)
            run_conv(x, w, 1, pads, ref.shape[1:3], scale, shift, True, ty, residual=res, tile=tile, expect=_lib.GV_E_UNSUPPORTED)
            continue
        if tile in ws_tiles()[5:]:              # may decline: 64-channel k-steps, two-per-CU workgroups (80 KB), register strips: cin % 64 == 0, and the strips / a 1x1's ring must fit LDS
            try:
                y = run_conv(x, w, 1, pads, ref.shape[1:3], scale, shift, True, ty, residual=res, tile=tile,
                             x_ld=cin + 16, x_off=8, y_ld=cout + 24, y_off=16)
            except _lib.GvError:            # (declined: channel count, or strips + ring past 160 KB for this map width)
                continue
            ran_k64 = True
        else:
            y = run_conv(x, w, 1, pads, ref.shape[1:3], scale, shift, True, ty, residual=res, tile=tile,
                         x_ld=cin + 16, x_off=8, y_ld=cout + 24, y_off=16)
        try:
            close(y, ref.numpy(), ulp)
        except AssertionError as e:
            raise AssertionError("wave-specialised tile %d: %s" % (ws_tiles().index(tile), str(e)[:400]))
    if same_grid and cin % 64 == 0 and k != (1, 1) and iw * (k[0] - 1) < 96:
        assert ran_k64


WS_FAST_COMBOS = [
    # (k, cin, cout, (ih, iw), nb): interior AND ragged tiles (M = nb*ih*iw is never a multiple of 256), cout % BN != 0
    ((1, 7), 128, 192, (17, 17), 5), ((7, 1), 192, 200, (12, 12), 9), ((3, 3), 64, 96, (25, 25), 3),
    ((1, 1), 192, 320, (12, 12), 7), ((5, 5), 64, 72, (13, 11), 6),
]


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("k,cin,cout,hw,nb", WS_FAST_COMBOS)
def test_lp_ws_tiles_plain_destination_fast_epilogue(k, cin, cout, hw, nb, ty, relu):
    """The pipelined straight-line epilogue of conv_ws.hip (`ws_epilogue_fast`): what every interior tile of a plain
    conv + BN (+ ReLU) launch takes — no residual, ONE destination — i.e. the tuned c3 / c5 plans' path.  Every
    wave-specialised tile, ReLU on and off, M with interior and ragged tiles, cout not a multiple of the tile's width:
    against the oracle, and bit for bit against the general staged epilogue (debug bit 1048576 forces it)."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(hash((k, cin, cout, hw)) % 1000)
    ih, iw = hw
    x = rnd(torch.randn(nb, ih, iw, cin, generator=g), td)
    w = rnd(torch.randn(k[0], k[1], cin, cout, generator=g) * (1.0 / (k[0] * k[1] * cin) ** 0.5), td)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    pads = (tf_pads(ih, k[0], 1, "SAME"), tf_pads(iw, k[1], 1, "SAME"))
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, relu)
    assert nb * ih * iw > 512 and (nb * ih * iw) % 256 != 0     # at least one interior 256- / 512-row tile and a ragged one
    ran = 0
    for tile in ws_tiles():
        kw = dict(tile=tile, x_ld=cin + 16, x_off=8, y_ld=cout + 24, y_off=16)
        try:
            y = run_conv(x, w, 1, pads, (ih, iw), scale, shift, relu, ty, **kw)
        except _lib.GvError:                    # (declined: 64-channel k-steps need cin % 64 == 0; LDS share)
            assert tile in ws_tiles()[5:]
            continue
        ran += 1
        try:
            close(y, ref.numpy(), ulp)
        except AssertionError as e:
            raise AssertionError("wave-specialised tile %d (fast epilogue): %s" % (ws_tiles().index(tile), str(e)[:400]))
        lib().gv_conv2d_set_debug(1048576)      # the general staged epilogue on the same accumulators
        try:
            y_gen = run_conv(x, w, 1, pads, (ih, iw), scale, shift, relu, ty, **kw)
        finally:
            lib().gv_conv2d_set_debug(0)
        assert np.array_equal(y, y_gen), "tile %d: fast and general epilogue differ" % ws_tiles().index(tile)
        # the consumers' tap-table form (round 6, debug bit 2097152; an A/B variant, not the product): the same fragments from
        # the same LDS bytes in the same order — bit for bit
        lib().gv_conv2d_set_debug(2097152)
        try:
            y_tab = run_conv(x, w, 1, pads, (ih, iw), scale, shift, relu, ty, **kw)
        finally:
            lib().gv_conv2d_set_debug(0)
        assert np.array_equal(y, y_tab), "tile %d: tap-table consumers differ" % ws_tiles().index(tile)
    assert ran >= 5


@pytest.mark.parametrize("ty", ["bf16", "f16"])
def test_lp_ws_tiles_split_and_dual_outputs(ty):
    """A fused sibling GEMM's two destinations (split on a chunk boundary, partial ReLU) and a second activation of the
    same values through every wave-specialised tile — the lean two-destination epilogue and the full one."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(11)
    x = rnd(torch.randn(4, 12, 12, 192, generator=g), td)
    w = rnd(torch.randn(1, 1, 192, 224, generator=g) * 0.07, td)
    scale, shift = torch.rand(224, generator=g) + 0.5, torch.randn(224, generator=g) * 0.1
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, True)
    sc2, sh2 = torch.rand(224, generator=g) + 0.5, torch.randn(224, generator=g) * 0.1
    ref2 = torch.relu(oracle_conv(x, w, 1, "SAME", scale, shift, False) * sc2 + sh2)
    for tile in ws_tiles():
        try:
            y, y2 = run_conv(x, w, 1, (0, 0), (12, 12), scale, shift, True, ty, split=64, y_ld=256, y_off=0, tile=tile)
        except _lib.GvError:                    # (a 1x1's ring past the workgroup's LDS share)
            assert tile in ws_tiles()[5:]
            continue
        close(y, ref.numpy()[..., :64], ulp)
        close(y2, ref.numpy()[..., 64:], ulp)
        y, y2 = run_conv(x, w, 1, (0, 0), (12, 12), scale, shift, False, ty, second=(sc2, sh2), tile=tile)
        close(y, oracle_conv(x, w, 1, "SAME", scale, shift, False).numpy(), ulp)
        close(y2, ref2.numpy(), ulp)


def test_lp_ws_tiles_decline_other_classes():
    """Strided, map-changing (VALID) and cin % 32 != 0 layers are not the wave-specialised kernel's: GV_E_UNSUPPORTED."""
    g = torch.Generator().manual_seed(5)
    U = _lib.GV_E_UNSUPPORTED
    x = rnd(torch.randn(2, 9, 10, 96, generator=g), torch.bfloat16)
    w = rnd(torch.randn(3, 3, 96, 64, generator=g) * 0.05, torch.bfloat16)
    run_conv(x, w, 2, (0, 0), (4, 4), torch.ones(64), torch.zeros(64), False, "bf16", tile=ws_tiles()[0], expect=U)
    run_conv(x, w, 1, (0, 0), (7, 8), torch.ones(64), torch.zeros(64), False, "bf16", tile=ws_tiles()[1], expect=U)
    x = rnd(torch.randn(2, 9, 10, 48, generator=g), torch.bfloat16)
    w = rnd(torch.randn(5, 5, 48, 64, generator=g) * 0.05, torch.bfloat16)
    run_conv(x, w, 1, (2, 2), (9, 10), torch.ones(64), torch.zeros(64), False, "bf16", tile=ws_tiles()[2], expect=U)


def test_lp_dma_k64_tiles_decline_other_channel_counts():
    """cin % 64 != 0: the 64-deep tiles return GV_E_UNSUPPORTED (the autotuner skips them), nothing is written."""
    g = torch.Generator().manual_seed(3)
    x = rnd(torch.randn(2, 9, 10, 96, generator=g), torch.bfloat16)
    w = rnd(torch.randn(3, 3, 96, 64, generator=g) * 0.05, torch.bfloat16)
    for tile_i in (16, 21):
        run_conv(x, w, 1, (1, 1), (9, 10), torch.ones(64), torch.zeros(64), False, "bf16", tile=dma_tiles()[tile_i],
                 expect=_lib.GV_E_UNSUPPORTED)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("cout,pad,hw,cin", [(32, 0, (23, 41), 32), (64, 1, (23, 41), 32), (64, 1, (8, 32), 32), (48, 0, (5, 70), 32),
                                             (32, 0, (36, 70), 32), (24, 1, (17, 33), 32),          # two rows per wave
                                             (32, 2, (21, 37), 64), (32, 1, (9, 40), 64), (24, 0, (12, 66), 64)])   # 64 -> 32 (data gradient of Conv2d_2b)
def test_lp_halo_stem_kernel(ty, cout, pad, hw, cin):
    """The halo-tiled 3x3 kernel of Conv2d_2a/2b (the last tile configuration): ragged strips (width not a multiple of
    32, height not a multiple of 4 / 8), VALID, SAME and the full padding of a data gradient, 32 and 64 input channels,
    with a residual, into a channel slice — equal to the implicit-GEMM kernel's result up to fp32 summation order."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(cout + pad + cin)
    ih, iw = hw
    x = rnd(torch.randn(3, ih, iw, cin, generator=g), td)
    w = rnd(torch.randn(3, 3, cin, cout, generator=g) * 0.06, td)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    oh, ow = ih + 2 * pad - 2, iw + 2 * pad - 2
    res = rnd(torch.randn(3, oh, ow, cout, generator=g), td)
    ref = oracle_conv(x, w, 1, (pad, pad, pad, pad) if pad == 2 else ("SAME" if pad else "VALID"), scale, shift, True, residual=res)
    y = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, ty, residual=res, tile=special_tile(), y_ld=cout + 16, y_off=8)
    close(y, ref.numpy(), ulp)
    y0 = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, ty, residual=res, tile=0, y_ld=cout + 16, y_off=8)
    close(y, y0, ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("k,pad,cout,hw", [(3, 0, 32, (47, 75)), (7, 3, 64, (47, 75)), (7, 3, 64, (224, 64)), (3, 0, 24, (9, 131))])
def test_lp_stem_strip_kernel(ty, k, pad, cout, hw):
    """The strip kernel of the 3-channel stems (Conv2d_1a 3x3/2 VALID, ResNet conv1 7x7/2 with explicit pad 3) reading
    the fp32 images: ragged strips and heights, equal to the gather kernel (same rounding of the inputs)."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(k + cout)
    ih, iw = hw
    x = torch.rand(2, ih, iw, 3, generator=g) - 0.5
    w = rnd(torch.randn(k, k, 3, cout, generator=g) * 0.2, td)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    oh, ow = (ih + 2 * pad - k) // 2 + 1, (iw + 2 * pad - k) // 2 + 1
    y = run_conv(x, w, 2, (pad, pad), (oh, ow), scale, shift, True, ty, x_f32=True, tile=special_tile())
    y0 = run_conv(x, w, 2, (pad, pad), (oh, ow), scale, shift, True, ty, x_f32=True, tile=2)
    close(y, y0, 2 * ulp)            # two roundings of sums in different order: 1 ulp, 2 across a binade boundary
    xr = rnd(x, td)
    ref = OB.conv2d(torch.nn.functional.pad(xr, (0, 0, pad, pad, pad, pad)) if pad else xr, w, 2, "VALID") * scale + shift
    close(y, torch.relu(ref).numpy(), 2 * ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
def test_lp_conv_fp32_network_input(ty):
    """GV_CONV_X_F32: the stem reads the fp32 images and rounds them in its loader (== casting first)."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 31, 29, 3, generator=g) - 0.5
    w = rnd(torch.randn(3, 3, 3, 32, generator=g) * 0.2, td)
    scale, shift = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    y = run_conv(x, w, 2, (0, 0), (15, 14), scale, shift, True, ty, x_f32=True)
    y_cast = run_conv(rnd(x, td), w, 2, (0, 0), (15, 14), scale, shift, True, ty)
    assert np.array_equal(y, y_cast)
    close(y, oracle_conv(rnd(x, td), w, 2, "VALID", scale, shift, True).numpy(), ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
def test_lp_conv_residual_dual_and_slices(ty):
    """ResNet unit tail (resnet_v2.py:87-91 + next preact :75) and concat-slice input/output."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(9)
    x = rnd(torch.randn(2, 7, 9, 64, generator=g), td)
    w = rnd(torch.randn(1, 1, 64, 256, generator=g) * 0.12, td)
    res = rnd(torch.randn(2, 7, 9, 256, generator=g), td)
    bias = torch.randn(256, generator=g) * 0.1
    sc2, sh2 = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    y, y2 = run_conv(x, w, 1, (0, 0), (7, 9), torch.ones(256), bias, False, ty, residual=res, second=(sc2, sh2))
    ref = oracle_conv(x, w, 1, "SAME", torch.ones(256), bias, False, residual=res)
    close(y, ref.numpy(), ulp)
    close(y2, torch.relu(ref * sc2 + sh2).numpy(), ulp, extra=3e-3)    # y2 is computed from the unrounded y
    # channel-slice in and out (16-byte aligned slice offsets)
    y = run_conv(x, w[:, :, :, :96], 1, (0, 0), (7, 9), torch.ones(96), bias[:96], True, ty, x_ld=80, x_off=8,
                 y_ld=128, y_off=24)
    close(y, oracle_conv(x, w[:, :, :, :96], 1, "SAME", torch.ones(96), bias[:96], True).numpy(), ulp)
    # misaligned slice: the gather path
    y = run_conv(x, w[:, :, :, :96], 1, (0, 0), (7, 9), torch.ones(96), bias[:96], True, ty, x_ld=70, x_off=3,
                 y_ld=101, y_off=5)
    close(y, oracle_conv(x, w[:, :, :, :96], 1, "SAME", torch.ones(96), bias[:96], True).numpy(), ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("tile_cfg", [0, 1, 2, 7, 8, 9])
@pytest.mark.parametrize("cin,cout,stride,hw", [(256, 64, 1, (13, 21)), (1024, 256, 1, (7, 7)), (72, 40, 2, (15, 9)),
                                                (8, 200, 1, (5, 31))])
def test_lp_conv_preactivation_on_load(ty, tile_cfg, cin, cout, stride, hw):
    """gv_conv2d_fwd_xpre: conv1 of a ResNet-v2 unit (resnet_v2.py:83) over relu(bn(x)) (resnet_v2.py:75) with the
    pre-activation applied by the loader.  Reference: the oracle's conv over the pre-activation computed in fp32 and
    rounded to the storage type (what the unit before it used to store); and the kernel's own stored-input form on that
    tensor, which it must match except where one fp32 fused-multiply-add rounds the other way (counted)."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(cin + cout + stride)
    x = rnd(torch.randn(3, hw[0], hw[1], cin, generator=g), td)
    w = rnd(torch.randn(1, 1, cin, cout, generator=g) * (cin ** -0.5), td)
    xs, xh = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    pre = rnd(torch.relu((x.double() * xs.double() + xh.double()).float()), td)
    oh, ow = (hw[0] - 1) // stride + 1, (hw[1] - 1) // stride + 1
    y = run_conv(x, w, stride, (0, 0), (oh, ow), scale, shift, True, ty, xpre=(xs, xh), tile_cfg=tile_cfg)
    close(y, oracle_conv(pre, w, stride, "VALID", scale, shift, True).numpy(), ulp, extra=1e-3)
    stored = run_conv(pre, w, stride, (0, 0), (oh, ow), scale, shift, True, ty, tile=max(tile_cfg - 1, 0))
    assert (y != stored).mean() < 2e-3, (y != stored).mean()
    # with the residual and in / out channel slices (the shortcut conv of a unit would read the same tensor)
    if tile_cfg == 0:
        res = rnd(torch.randn(3, oh, ow, cout, generator=g), td)
        y = run_conv(x, w, stride, (0, 0), (oh, ow), scale, shift, False, ty, residual=res, xpre=(xs, xh), x_ld=cin + 8,
                     x_off=8, y_ld=cout + 16, y_off=8)
        close(y, oracle_conv(pre, w, stride, "VALID", scale, shift, False, residual=res).numpy(), ulp, extra=1e-3)


def test_lp_conv_preactivation_on_load_rejects_what_it_cannot_do():
    g = torch.Generator().manual_seed(1)
    x, one = torch.randn(1, 6, 6, 64, generator=g), torch.ones(64)
    w3, w1 = torch.randn(3, 3, 64, 64, generator=g), torch.randn(1, 1, 64, 64, generator=g)
    bad = dict(xpre=(one, one), expect=-2)
    assert run_conv(x, w3, 1, (1, 1), (6, 6), one, one, True, "bf16", **bad) is None          # 3x3 window
    assert run_conv(x, w1, 1, (1, 1), (7, 7), one, one, True, "bf16", **bad) is None          # padded 1x1
    assert run_conv(x, w1, 1, (0, 0), (6, 6), one, one, True, "bf16", tile_cfg=4, **bad) is None   # no such instantiation
    assert run_conv(x, w1, 1, (0, 0), (6, 6), one, one, True, "bf16", tile_cfg=14, **bad) is None  # an LDS-DMA tile
    assert run_conv(x, w1, 1, (0, 0), (6, 6), one, one, True, "bf16", x_ld=70, x_off=3, **bad) is None   # gather path
    d = _lib.ConvDesc(1, 6, 6, 64, 64, 1, 1, 1, 0, 0, 6, 6, 64, 64, 0, 0, 0, _lib.GV_BF16, 0, 0, 0, 0)
    assert lib().gv_conv2d_fwd_xpre(C.byref(d), 16, None, None, 16, 16, 16, None, 16, None, None, None, st()) == -1
    d32 = _lib.ConvDesc(1, 6, 6, 64, 64, 1, 1, 1, 0, 0, 6, 6, 64, 64, 0, 0, 0, _lib.GV_F32, 0, 0, 0, 0)
    buf = torch.zeros(1 << 16, device=DEV)
    p = buf.data_ptr()
    assert lib().gv_conv2d_fwd_xpre(C.byref(d32), p, p, p, p, p, p, None, p, None, None, None, st()) == -2


@pytest.mark.parametrize("ty", ["bf16", "f16"])
def test_lp_conv_split_output(ty):
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(11)
    x = rnd(torch.randn(2, 12, 12, 192, generator=g), td)
    w = rnd(torch.randn(1, 1, 192, 64 + 48 + 64, generator=g) * 0.07, td)
    scale, shift = torch.rand(176, generator=g) + 0.5, torch.randn(176, generator=g) * 0.1
    y, y2 = run_conv(x, w, 1, (0, 0), (12, 12), scale, shift, True, ty, split=64, y_ld=256, y_off=0)
    ref = oracle_conv(x, w, 1, "SAME", scale, shift, True).numpy()
    close(y, ref[..., :64], ulp)
    close(y2, ref[..., 64:], ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("pad,hw,nb", [(1, (109, 109), 2), (1, (147, 147), 1), (1, (19, 23), 3), (0, (5, 5), 2), (1, (3, 3), 2),
                                       (0, (12, 37), 3), (1, (8, 64), 2), (1, (16, 31), 2), (0, (11, 95), 1), (1, (9, 33), 2)])
def test_lp_conv_maxpool_one_launch(ty, pad, hw, nb):
    """GV_CONV_MAXPOOL3S2 (Conv2d_2b_3x3 -> MaxPool_3a_3x3, nets/inception_v3.py:111-113, as ONE launch that writes only the
    pooled tensor): bit for bit the two launches — the same halo kernel followed by gv_pool2d — and, through them, the
    oracle's conv -> max_pool2d; maps whose pooled width is / is not a multiple of the 15 pooled columns of a strip and whose
    height ends inside a tile, VALID and SAME, into a channel slice of a wider buffer (nothing else is written)."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(hw[0] * 1000 + hw[1] + pad)
    ih, iw = hw
    x = rnd(torch.randn(nb, ih, iw, 32, generator=g), td)
    w = rnd(torch.randn(3, 3, 32, 64, generator=g) * 0.06, td)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    oh, ow = ih + 2 * pad - 2, iw + 2 * pad - 2
    ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
    y = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, ty, tile=special_tile())
    two = run_pool(torch.from_numpy(y), 3, 2, (0, 0), (ph, pw), _lib.GV_POOL_MAX, ty)
    ref = OB.max_pool2d(oracle_conv(x, w, 1, "SAME" if pad else "VALID", scale, shift, True), 3, 2, "VALID")
    close(two, ref.numpy(), ulp)
    for tile_cfg, tile in ((0, None), (special_tile() + 1, None), (0, special_tile())):     # heuristic, plan's choice, override
        one = run_conv(x, w, 1, (pad, pad), (oh, ow), scale, shift, True, ty, tile=tile, tile_cfg=tile_cfg, y_ld=64 + 16, y_off=8,
                       pooled=(ph, pw))
        assert np.array_equal(one, two), (tile_cfg, tile, np.abs(one - two).max())


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("k,pad,hw,same,relu", [(7, 3, (64, 64), True, False), (7, 3, (224, 120), True, False), (3, 0, (65, 129), True, True),
                                                (3, 0, (47, 75), False, True), (7, 3, (50, 66), False, False), (7, 3, (12, 8), True, False),
                                                (7, 3, (16, 128), True, False)])
def test_lp_stem_conv_maxpool_one_launch(ty, k, pad, hw, same, relu):
    """The strip kernel of the 3-channel stems with the max pool behind it in the same launch (ResNet-v2: conv1 7x7 / 2 with
    a bias and no activation -> pool1 3x3 / 2 SAME, nets/resnet_v2.py:178-181; signed values, the last window clipped):
    bit for bit the two launches, and through them the oracle; VALID pools and a ReLU in front of the pool as well."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(k * 100 + hw[0] + hw[1])
    ih, iw = hw
    x = torch.randn(2, ih, iw, 3, generator=g)
    w = rnd(torch.randn(k, k, 3, 64, generator=g) * 0.1, td)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    oh, ow = (ih + 2 * pad - k) // 2 + 1, (iw + 2 * pad - k) // 2 + 1
    ph, pw = (oh // 2, ow // 2) if same else ((oh - 3) // 2 + 1, (ow - 3) // 2 + 1)
    y = run_conv(x, w, 2, (pad, pad), (oh, ow), scale, shift, relu, ty, tile=special_tile(), x_f32=True)
    assert relu or float(y.min()) < 0                                       # signed values reach the pool
    two = run_pool(torch.from_numpy(y), 3, 2, (0, 0), (ph, pw), _lib.GV_POOL_MAX, ty)
    ref = OB.max_pool2d(oracle_conv(rnd(x, td), w, 2, (pad, pad, pad, pad), scale, shift, relu), 3, 2, "SAME" if same else "VALID")
    close(two, ref.numpy(), ulp)
    for tile_cfg, tile in ((0, None), (special_tile() + 1, None), (0, special_tile())):
        one = run_conv(x, w, 2, (pad, pad), (oh, ow), scale, shift, relu, ty, tile=tile, tile_cfg=tile_cfg, x_f32=True,
                       y_ld=64 + 16, y_off=8, pooled=(ph, pw), pooled_same=same)
        assert np.array_equal(one, two), (tile_cfg, tile, np.abs(one - two).max())
    # GV_CONV_POOL_ACT2: the pooled tensor leaves through a second BatchNorm (+ ReLU) — ResNet-v2's first `preact`
    # (nets/resnet_v2.py:75), whose only input is pool1: bit for bit gv_scale_shift_act of the pooled tensor
    s2, h2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    for relu2 in (True, False):
        td_two = torch.from_numpy(two).to(td).to(DEV).contiguous()
        three = torch.empty_like(td_two)
        s2d, h2d = s2.to(DEV), h2.to(DEV)
        _lib.check(lib().gv_scale_shift_act(td_two.data_ptr(), td_two.numel() // 64, 64, 64, s2d.data_ptr(), h2d.data_ptr(),
                                            int(relu2), three.data_ptr(), 64, code, st()), "ssa")
        three = three.float().cpu().numpy()
        v = torch.from_numpy(two) * s2 + h2
        close(three, (torch.relu(v) if relu2 else v).numpy(), ulp)
        assert relu2 or float(three.min()) < 0
        one = run_conv(x, w, 2, (pad, pad), (oh, ow), scale, shift, relu, ty, tile_cfg=special_tile() + 1, x_f32=True,
                       y_ld=64 + 16, y_off=8, pooled=(ph, pw), pooled_same=same, pool_act=(s2, h2, relu2))
        assert np.array_equal(one, three), (relu2, np.abs(one - three).max())


def test_lp_conv_maxpool_declines_what_it_does_not_serve():
    """Outside the halo kernel's 32 -> 64 channel ReLU class the flag is refused and nothing is written (the plan builder
    then issues the two launches): another tile, no ReLU, a residual, other channel counts, a 2 x 2 map, fp32 storage."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 12, 14, 32, generator=g)
    w = torch.randn(3, 3, 32, 64, generator=g) * 0.06
    sc, sh = torch.ones(64), torch.zeros(64)
    U = _lib.GV_E_UNSUPPORTED
    run_conv(x, w, 1, (1, 1), (12, 14), sc, sh, True, "bf16", tile=0, pooled=(5, 6), expect=U)
    run_conv(x, w, 1, (1, 1), (12, 14), sc, sh, True, "bf16", tile=dma_tiles()[0], pooled=(5, 6), expect=U)
    run_conv(x, w, 1, (1, 1), (12, 14), sc, sh, False, "bf16", pooled=(5, 6), expect=U)
    run_conv(x, w, 1, (1, 1), (12, 14), sc, sh, True, "bf16", residual=torch.zeros(2, 12, 14, 64), pooled=(5, 6), expect=U)
    run_conv(x, w[..., :32], 1, (1, 1), (12, 14), sc[:32], sh[:32], True, "bf16", pooled=(5, 6), expect=U)
    x64 = torch.randn(2, 12, 14, 64, generator=g)
    run_conv(x64, torch.randn(3, 3, 64, 64, generator=g) * 0.05, 1, (1, 1), (12, 14), sc, sh, True, "bf16", pooled=(5, 6), expect=U)
    run_conv(x[:, :4, :4], w, 1, (0, 0), (2, 2), sc, sh, True, "bf16", pooled=(1, 1), expect=U)
    # the SAME form: the stem strip kernel only, even maps only
    run_conv(x, w, 1, (1, 1), (12, 14), sc, sh, True, "bf16", pooled=(6, 7), pooled_same=True, expect=U)
    xi = torch.randn(2, 33, 37, 3, generator=g)
    w7 = torch.randn(7, 7, 3, 64, generator=g) * 0.1
    run_conv(xi, w7, 2, (3, 3), (17, 19), sc, sh, False, "bf16", x_f32=True, pooled=(8, 9), pooled_same=True, expect=U)
    run_conv(xi, w7[..., :32], 2, (3, 3), (17, 19), sc[:32], sh[:32], False, "bf16", x_f32=True, pooled=(8, 9), expect=U)
    # GV_CONV_POOL_ACT2: with a fused pool only, its vectors present, the stem strip kernel only
    B = _lib.GV_E_BADARG
    run_conv(x, w, 1, (1, 1), (12, 14), sc, sh, True, "bf16", pool_act=(sc, sh, True), expect=B)
    run_conv(xi, w7, 2, (3, 3), (17, 19), sc, sh, False, "bf16", x_f32=True, pooled=(8, 9), pool_act=(sc, None, True), expect=B)
    run_conv(x, w, 1, (1, 1), (12, 14), sc, sh, True, "bf16", pooled=(5, 6), pool_act=(sc, sh, True), expect=U)


@pytest.mark.parametrize("ty,size,nb", [("bf16", 107, 6), ("f16", 75, 4), ("bf16", 224, 24), ("f16", 299, 10)])
def test_lp_inception_plan_with_and_without_the_fused_max_pool(ty, size, nb):
    """The 16-bit Inception plan issues Conv2d_2b_3x3 -> MaxPool_3a_3x3 as one launch (one op fewer, the un-pooled tensor
    does not exist); with `fuse_maxpool=False` and the same kernel for Conv2d_2b the two launches give the same bits at
    MaxPool_3a_3x3 and at Mixed_7c.  A tapped Conv2d_2b_3x3 keeps the two launches."""
    from gvcnn_tf_amd import backbones
    x = (torch.rand(nb, size, size, 3, generator=torch.Generator().manual_seed(size)) - 0.5).to(DEV)
    outs, nops = [], []
    for fuse in (True, False):
        plan = backbones.make_plan("inception_v3", nb, size, size, DEV, dtype=ty, lanes=False, fuse_maxpool=fuse,
                                   raw_tap="MaxPool_3a_3x3")          # (a kept tensor: its buffer is not recycled)
        plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
        names = [op["name"] for op in plan.ops]
        assert ("MaxPool_3a_3x3" in names) == (not fuse)
        if not fuse:
            plan.apply_tiles({"InceptionV3/Conv2d_2b_3x3": special_tile()})
        plan.run(x)
        torch.cuda.synchronize()
        outs.append({k: plan.view(plan.end_points[k]).clone() for k in ("MaxPool_3a_3x3", "Mixed_7c")})
        nops.append(len(plan.ops))
        assert ("Conv2d_2b_3x3" in plan.end_points) == (not fuse)
    assert nops[0] == nops[1] - 1
    for k in outs[0]:
        assert outs[0][k].shape == outs[1][k].shape and torch.equal(outs[0][k], outs[1][k]), k
    assert float(outs[0]["Mixed_7c"].float().abs().max()) > 1e-3
    tapped = backbones.make_plan("inception_v3", nb, size, size, DEV, dtype=ty, lanes=False, raw_tap="Conv2d_2b_3x3")
    assert "MaxPool_3a_3x3" in [op["name"] for op in tapped.ops]


@pytest.mark.parametrize("ty,size,nb", [("bf16", 64, 6), ("f16", 224, 12), ("bf16", 97, 4)])
def test_lp_resnet_plan_with_and_without_the_fused_max_pool(ty, size, nb):
    """The 16-bit ResNet-v2-50 plan issues conv1 -> pool1 -> the first unit's preact as one launch where conv1's map is even
    (64, 224: two ops fewer — GV_CONV_MAXPOOL3S2_SAME | GV_CONV_POOL_ACT2); at 97 (a 49 x 49 map: TF's SAME pads (1, 1)
    there) it keeps the three.  Same bits at block3 / block4 either way."""
    from gvcnn_tf_amd import backbones
    x = (torch.rand(nb, size, size, 3, generator=torch.Generator().manual_seed(size)) - 0.5).to(DEV)
    outs, nops = [], []
    for fuse in (True, False):
        plan = backbones.make_plan("resnet_v2_50", nb, size, size, DEV, dtype=ty, lanes=False, fuse_maxpool=fuse)
        plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
        if any(op["name"].endswith("/pool1") for op in plan.ops):          # two launches: conv1 on the same kernel
            plan.apply_tiles({"resnet_v2_50/conv1": special_tile()})
        plan.run(x)
        torch.cuda.synchronize()
        outs.append({k: plan.view(plan.end_points[k]).clone() for k in ("resnet_v2_50/block3", "resnet_v2_50/block4")})
        nops.append(len(plan.ops))
    assert nops[0] == nops[1] - (2 if size != 97 else 0)
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert float(outs[0]["resnet_v2_50/block4"].float().abs().max()) > 1e-3


def run_pool(x, k, stride, pads, out_hw, mode, ty, x_ld=None, y_ld=None, y_off=0):
    code, td, _ = TYPES[ty]
    nb, ih, iw, c = x.shape
    x_ld = x_ld or c
    y_ld = y_ld or c
    xb = torch.full((nb, ih, iw, x_ld), 7.0)
    xb[..., :c] = x
    xd = xb.to(td).to(DEV)
    yd = torch.full((nb, out_hw[0], out_hw[1], y_ld), -77.0, dtype=td, device=DEV)
    d = _lib.PoolDesc(nb, ih, iw, c, x_ld, k, k, stride, pads[0], pads[1], out_hw[0], out_hw[1], y_ld, mode, code)
    _lib.check(lib().gv_pool2d_fwd(C.byref(d), xd.data_ptr(), yd.data_ptr() + 2 * y_off, st()), "pool")
    torch.cuda.synchronize()
    return yd.float().cpu().numpy()[..., y_off:y_off + c]


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("c", [64, 20])
def test_lp_pools(ty, c):
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(c)
    x = rnd(torch.randn(2, 15, 14, c, generator=g), td)
    # max 3x3/2 VALID (inception_v3.py:112) and SAME with TF's (0,1) pad (resnet_v2.py:181): exact
    y = run_pool(x, 3, 2, (0, 0), (7, 6), _lib.GV_POOL_MAX, ty, y_ld=c + 16, y_off=8)
    assert np.array_equal(y, OB.max_pool2d(x, 3, 2, "VALID").numpy())
    ref = OB.max_pool2d(x, 3, 2, "SAME")
    pt, pl = OB.same_pads(15, 3, 2)[0], OB.same_pads(14, 3, 2)[0]
    y = run_pool(x, 3, 2, (pt, pl), ref.shape[1:3], _lib.GV_POOL_MAX, ty)
    assert np.array_equal(y, ref.numpy())
    # subsample (resnet_utils.py:64-67)
    y = run_pool(x, 1, 2, (0, 0), (8, 7), _lib.GV_POOL_MAX, ty)
    assert np.array_equal(y, x[:, ::2, ::2].numpy())
    # avg 3x3/1 SAME, valid-tap divisor (inception_v3.py:152)
    y = run_pool(x, 3, 1, (1, 1), (15, 14), _lib.GV_POOL_AVG, ty)
    close(y, OB.avg_pool2d_same3(x).numpy(), ulp)
    # non-finite inputs (an f16 activation that overflowed): a window with +inf averages to +inf, one with -inf to -inf,
    # one with a NaN (or with both infinities) to NaN — what the division gives; the reciprocal-and-correction form must
    # not turn an infinity into a NaN
    xi = x.clone()
    xi[0, 4, 4, 0] = float("inf")
    xi[0, 10, 9, 1] = float("-inf")
    xi[1, 7, 7, 2] = float("nan")
    yi = run_pool(xi, 3, 1, (1, 1), (15, 14), _lib.GV_POOL_AVG, ty)
    ri = OB.avg_pool2d_same3(xi).numpy()
    assert np.array_equal(np.isnan(yi), np.isnan(ri))
    assert np.array_equal(np.isposinf(yi), np.isposinf(ri)) and np.array_equal(np.isneginf(yi), np.isneginf(ri))
    assert np.isposinf(yi[0, 3:6, 3:6, 0]).all() and np.isneginf(yi[0, 9:12, 8:11, 1]).all() and np.isnan(yi[1, 6:9, 6:9, 2]).all()
    fin = np.isfinite(ri)
    close(np.where(fin, yi, 0.0), np.where(fin, ri, 0.0), ulp)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("hw", [(3, 3), (9, 11), (10, 9), (17, 21), (35, 35), (12, 7)])
def test_lp_multi_row_pool_forms(ty, hw):
    """The multi-row form of the 3x3 / stride-2 max pool (round 4: four vertically adjacent outputs per thread) on maps
    whose height and width are and are not multiples of four, into a channel slice of a wider buffer: bit for bit against
    the oracle and against the one-output form; the average pool (row-of-4 form) beside it within one rounding of the oracle
    (nets/inception_v3.py:112,152)."""
    code, td, ulp = TYPES[ty]
    h, w = hw
    g = torch.Generator().manual_seed(h * 100 + w)
    x = rnd(torch.randn(3, h, w, 72, generator=g), td)
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    outs = []
    for rows in (1, 0):
        lib().gv_pool2d_set_rows(rows)
        try:
            ym = run_pool(x, 3, 2, (0, 0), (oh, ow), _lib.GV_POOL_MAX, ty, y_ld=72 + 16, y_off=8)
            ya = run_pool(x, 3, 1, (1, 1), (h, w), _lib.GV_POOL_AVG, ty, y_ld=72 + 8, y_off=8)
        finally:
            lib().gv_pool2d_set_rows(1)
        assert np.array_equal(ym, OB.max_pool2d(x, 3, 2, "VALID").numpy())
        close(ya, OB.avg_pool2d_same3(x).numpy(), ulp)
        outs.append((ym, ya))
    assert np.array_equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("ty", ["bf16", "f16"])
def test_lp_scale_shift_gap_and_scorer(ty):
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(3)
    N, V, h, w, c = 2, 3, 4, 5, 72
    x = rnd(torch.randn(N * V, h, w, c, generator=g), td)
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    xd = x.to(td).to(DEV)
    yd = torch.empty_like(xd)
    scd, shd = scale.to(DEV), shift.to(DEV)
    _lib.check(lib().gv_scale_shift_act(xd.data_ptr(), N * V * h * w, c, c, scd.data_ptr(), shd.data_ptr(), 1,
                                        yd.data_ptr(), c, code, st()), "ssa")
    close(yd.float().cpu().numpy(), torch.relu(x * scale + shift).numpy(), ulp)
    gap = torch.empty(N * V, c, device=DEV)
    _lib.check(lib().gv_global_avg_pool(xd.data_ptr(), N * V, h * w, c, c, gap.data_ptr(), code, st()), "gap")
    np.testing.assert_allclose(gap.cpu().numpy(), x.mean(dim=(1, 2)).numpy(), rtol=1e-5, atol=1e-6)
    k = torch.randn(V, c, generator=g) * 0.3
    b = torch.randn(V, generator=g)
    kd, bd = k.to(DEV), b.to(DEV)
    r = torch.empty(N * V, device=DEV)
    _lib.check(lib().gv_view_score_partial(xd.data_ptr(), N * V, h * w, c, c, kd.data_ptr(), bd.data_ptr(), V,
                                           _lib.GV_ORDER_SHAPE_MAJOR, r.data_ptr(), code, st()), "score")
    want = [float(x[i].mean(dim=(0, 1)) @ k[i % V] + b[i % V]) for i in range(N * V)]
    np.testing.assert_allclose(r.cpu().numpy(), want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("E", [2 * 2 * 64, 3 * 5 * 7])
def test_lp_view_pool_fuse(ty, E):
    """model.py:44-102 on 16-bit descriptors: D (max) is exact, S within one rounding."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(E)
    V, N, G = 6, 3, 5
    F = rnd(torch.randn(N, V, E, generator=g), td)
    scheme = OG.group_scheme([[0.05, 0.31, 0.32, 0.47, 0.07, 0.39]], G, V)
    weight = OG.group_weight(scheme)
    Fd = F.to(td).to(DEV)
    D = torch.empty(G, N, E, dtype=td, device=DEV)
    S = torch.empty(N, E, dtype=td, device=DEV)
    sd = torch.from_numpy(scheme.astype(np.int32)).to(DEV)
    wd = torch.from_numpy(weight).to(DEV)
    _lib.check(lib().gv_view_pool_fuse_fwd(Fd.data_ptr(), V, N, E, E, V * E, sd.data_ptr(), G, wd.data_ptr(),
                                           _lib.GV_VIEWPOOL_MAX, 1.0, D.data_ptr(), S.data_ptr(), code, st()), "fuse")
    descs = [F[:, v].reshape(N, 1, 1, E).numpy() for v in range(V)]
    oD = OG.view_pooling(descs, scheme)
    oS = OG.group_fusion(oD, weight)
    for gi in range(G):
        assert np.array_equal(D[gi].float().cpu().numpy(), oD[gi].reshape(N, E))
    close(S.float().cpu().numpy(), oS.reshape(N, E), ulp)


# ------------------------------------------------------------------------------------------------
# end to end: the whole path on 16-bit storage against the fp32 oracle
# ------------------------------------------------------------------------------------------------
def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("backbone,size,ty", [("inception_v3", 75, "bf16"), ("resnet_v2_50", 64, "bf16"),
                                              ("inception_v3", 75, "f16"), ("resnet_v2_50", 64, "f16")])
def test_lp_gvcnn_vs_fp32_oracle(backbone, size, ty):
    """Config c3/c4/c5 dtype path.  Against the fp32 oracle the descriptors carry the storage rounding of
    every layer (~depth^0.5 * 2^-9 for bf16): bound = 3e-2 (bf16) / 4e-3 (fp16) in relative L2 and
    8x that per element relative to the tensor's scale; the integer group assignment must be identical
    (the synthetic scorer keeps scores away from bin edges)."""
    code, td, ulp = TYPES[ty]
    N, V, C_, G = 2, 6, 10, 10
    eng = gv.GVCNN(backbone, N, V, size, size, C_, G, device=DEV, storage=ty)
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C_, seed=3, spread_scores=True)
    eng.plan.bind(P)
    eng.set_head(Hd)
    x = torch.rand(N, V, size, size, 3, generator=torch.Generator().manual_seed(1)) - 0.5
    scores, S, logits = eng.forward(x.to(DEV))
    o_scores, o_S, o_logits, o_scheme, o_weight = OM.gvcnn(x, C_, P, Hd, G, backbone)
    bound = 3e-2 if ty == "bf16" else 4e-3
    fin = eng.final_view_descriptors().float().cpu().numpy()
    assert np.isfinite(fin).all()
    assert S.dtype == td
    np.testing.assert_allclose(scores.cpu().numpy(), np.array(o_scores), rtol=bound, atol=bound * 0.1)
    assert eng.scheme.cpu().numpy().tolist() == o_scheme.tolist()
    assert eng.weight.cpu().numpy().tolist() == o_weight.tolist()
    Sn = S.float().cpu().numpy()
    assert rel_l2(Sn, o_S) < bound, rel_l2(Sn, o_S)
    assert np.abs(Sn - o_S).max() < 8 * bound * np.abs(o_S).max()
    assert rel_l2(logits.cpu().numpy(), o_logits) < bound, rel_l2(logits.cpu().numpy(), o_logits)
    # MVCNN baseline (model.py:169-206)
    Sb, lb = eng.forward_basic(x.to(DEV))
    _, ob_logits = OM.basic(x, C_, P, Hd, backbone)
    assert rel_l2(lb.cpu().numpy(), ob_logits) < bound


@pytest.mark.parametrize("ty", ["bf16", "f16"])
def test_lp_pooled_branch_commutes_with_the_1x1_conv(ty):
    """relu_cols + GV_POOL_AVG_RELU on 16-bit storage (see the fp32 test of the same name): against the
    reference order the only extra error is the rounding of the un-pooled 1x1 output: 2 ulp."""
    code, td, ulp = TYPES[ty]
    g = torch.Generator().manual_seed(21)
    nb, h, wd, cin, couts = 2, 9, 11, 64, (32, 48, 40)
    total = sum(couts)
    x = rnd(torch.randn(nb, h, wd, cin, generator=g), td)
    wcat = rnd(torch.randn(1, 1, cin, total, generator=g) * 0.1, td)
    scale = torch.rand(total, generator=g) + 0.5
    shift = torch.randn(total, generator=g) * 0.1
    n_relu = couts[0] + couts[1]
    ref_pooled = oracle_conv(OB.avg_pool2d_same3(x), wcat[..., n_relu:], 1, "SAME", scale[n_relu:], shift[n_relu:],
                             True).numpy()
    xd, wp, sc, sh = x.to(td).to(DEV), pack(wcat, code), scale.to(DEV), shift.to(DEV)
    yd = torch.empty(nb, h, wd, couts[0], dtype=td, device=DEV)
    y2d = torch.empty(nb, h, wd, total - couts[0], dtype=td, device=DEV)
    d = _lib.ConvDesc(nb, h, wd, cin, cin, 1, 1, 1, 0, 0, h, wd, total, couts[0], 0, total - couts[0],
                      _lib.GV_CONV_RELU | _lib.GV_CONV_SPLIT, code, couts[0], 0, 0, 0, n_relu)
    _lib.check(lib().gv_conv2d_fwd(C.byref(d), xd.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                   None, yd.data_ptr(), y2d.data_ptr(), None, None, st()), "conv")
    out = torch.empty(nb, h, wd, couts[2], dtype=td, device=DEV)
    pd = _lib.PoolDesc(nb, h, wd, couts[2], total - couts[0], 3, 3, 1, 1, 1, h, wd, couts[2], _lib.GV_POOL_AVG_RELU, code)
    _lib.check(lib().gv_pool2d_fwd(C.byref(pd), y2d.data_ptr() + 2 * couts[1], out.data_ptr(), st()), "pool")
    torch.cuda.synchronize()
    assert (y2d.float().cpu().numpy()[..., couts[1]:] < 0).any()
    assert (y2d.float().cpu().numpy()[..., :couts[1]] >= 0).all()
    close(out.float().cpu().numpy(), ref_pooled, 2 * ulp, extra=2 * ulp)


@pytest.mark.parametrize("backbone,V,size,G,ty", [("resnet_v2_50", 12, 224, 10, "bf16"),      # configs[3]
                                                  ("inception_v3", 20, 299, 10, "f16"),      # configs[4]
                                                  ("inception_v3", 12, 224, 7, "bf16")])     # configs[2] (forward)
def test_lp_baseline_configs_at_full_size(backbone, V, size, G, ty):
    """BASELINE.json configs[2..4] at their real geometry and dtype: one image of the batch through the fp32 CPU
    oracle backbone (bound as in test_lp_gvcnn_vs_fp32_oracle), the oracle grouping head on the device
    descriptors (the head adds one rounding of S), and the size-independent weight identity."""
    code, td, ulp = TYPES[ty]
    N, C_ = 2, 40
    # num_bins = G: with the reference's literal 10 bins a score >= G/10 is an IndexError (model.py:23)
    eng = gv.GVCNN(backbone, N, V, size, size, C_, G, device=DEV, storage=ty, num_bins=G)
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C_, seed=3, spread_scores=True)
    eng.plan.bind(P)
    eng.set_head(Hd)
    x = torch.rand(N, V, size, size, 3, generator=torch.Generator().manual_seed(5)) - 0.5
    scores, S, logits = eng.forward(x.to(DEV))
    F = eng.final_view_descriptors().clone()
    assert tuple(F.shape[2:]) == ((8, 8, 2048) if size == 299 else ((7, 7, 2048) if backbone == "resnet_v2_50" else (5, 5, 2048)))
    assert float(eng.weight.sum()) == G + V and torch.isfinite(logits).all()
    b = 7
    ep = OM.run_backbone(backbone, x.reshape(N * V, size, size, 3)[b:b + 1], P)
    bound = 3e-2 if ty == "bf16" else 4e-3
    got = F.reshape(N * V, *F.shape[2:])[b].float().cpu().numpy()
    assert rel_l2(got, ep[eng.plan.final_tap][0].numpy()) < bound
    oS, oL = OG.grouping_head([F[:, v].float().cpu().numpy() for v in range(V)], eng.scheme.cpu().numpy(),
                              eng.weight.cpu().numpy(), Hd["dense_%d/kernel" % V].numpy(),
                              Hd["dense_%d/bias" % V].numpy())
    close(S.float().cpu().numpy(), oS, ulp)
    assert rel_l2(logits.cpu().numpy(), oL) < 2 * ulp
