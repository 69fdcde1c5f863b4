"""The deterministic filter gradient (gv_conv2d_wgrad_ws) and the reproducible training step built on it.

The reference's CPU path computes the same gradients on every run (utils/train_utils.py:217-259: tf.gradients on
the host kernels).  The plain gv_conv2d_wgrad combines its pixel slices with fp32 atomics — the order they arrive in
decides the last bits, a randomly initialised train-mode-BN network amplifies them, and two runs of one engine used to
end 150 steps later at different accuracies.  The `_ws` form stores each slice's partial image of dW into a workspace
and adds the slices in slice order.  Checked here, for every kernel form behind the entry point (LDS-DMA tiles,
register-staged tiles, strips, the 3-channel stem rows, the fp32-MFMA kernels, the direct stem kernel):

  * against torch autograd through the oracle's convolution (same tolerance as the plain form's tests);
  * bitwise equal over repeated launches, also while other work perturbs the dispatch order, and for a workspace so
    small that the number of slices is cut down (and for one that holds a single image: the un-split launch);
  * the whole step: two engines, same seeds -> every gradient, the loss and every updated variable bitwise equal after
    several Momentum steps (both backbones, fp32 and bf16 storage, fused and parity-class paths included).
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gvcnn_tf_amd import _lib                       # noqa: E402
from gvcnn_tf_amd.training import TrainGVCNN        # noqa: E402
from oracle import backbone as OB                   # noqa: E402

DEV = "cuda:0"
TDT = {_lib.GV_F32: torch.float32, _lib.GV_BF16: torch.bfloat16, _lib.GV_F16: torch.float16}


def lib():
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def close(a, d, tol):
    a, d = np.asarray(a, dtype=np.float64), np.asarray(d, dtype=np.float64)
    scale = max(float(np.abs(d).max()), 1e-30)
    err = float(np.abs(a - d).max())
    assert err <= tol * scale, "max|diff| %.3e vs scale %.3e (%.2e rel)" % (err, scale, err / scale)


def case(dt, k, stride, padding, cin, cout, nb, ih, iw, seed=0):
    tdt = TDT[dt]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(nb, ih, iw, cin, generator=g).to(tdt).float().requires_grad_(True)
    w = (torch.randn(k[0], k[1], cin, cout, generator=g) * 0.1).requires_grad_(True)
    z = OB.conv2d(x.to(DEV), w.to(DEV), stride, padding)
    dz = torch.randn(*z.shape, generator=g).to(tdt).float()
    z.backward(dz.to(DEV))
    oh, ow = z.shape[1:3]
    pt = OB.same_pads(ih, k[0], stride)[0] if padding == "SAME" else 0
    pl = OB.same_pads(iw, k[1], stride)[0] if padding == "SAME" else 0
    xd, dzd = x.detach().to(tdt).to(DEV).contiguous(), dz.to(tdt).to(DEV).contiguous()
    return xd, dzd, w.grad, (oh, ow, pt, pl)


def run_ws(dt, k, stride, cin, cout, nb, ih, iw, geo, xd, dzd, cfg, ws, fill=0.5):
    oh, ow, pt, pl = geo
    dw = torch.full((k[0], k[1], cin, cout), fill, device=DEV)
    d = _lib.ConvDesc(nb, ih, iw, cin, cin, k[0], k[1], stride, pt, pl, oh, ow, cout, cout, 0, 0, 0, dt, 0, cfg, 0, 0)
    rc = lib().gv_conv2d_wgrad_ws(C.byref(d), xd.data_ptr(), dzd.data_ptr(), cout, dw.data_ptr(), ws.data_ptr(),
                                  ws.numel(), st())
    return rc, dw


# (dtype, kernel, stride, padding, cin, cout, nb, ih, iw, tile_cfgs): one row per kernel form behind the entry point
FORMS = [
    # LDS-DMA tiles (default for 16-bit), register-staged tiles (1..27), strips (28..30), a few of each
    (_lib.GV_BF16, (3, 3), 1, "SAME", 64, 96, 12, 35, 35, [0, 1, 5, 14, 27, 28, 30, 31, 38, 47, 55, 60, 64, 74, 77, 86]),
    (_lib.GV_F16, (1, 7), 1, "SAME", 128, 192, 12, 17, 17, [0, 9, 31, 44, 58]),
    (_lib.GV_BF16, (1, 1), 1, "SAME", 288, 448, 24, 12, 12, [0, 27, 34, 64, 66, 73, 77, 78, 91]),
    (_lib.GV_BF16, (3, 3), 2, "VALID", 96, 96, 8, 25, 25, [0, 31]),
    (_lib.GV_BF16, (3, 3), 1, "VALID", 32, 32, 8, 55, 55, [0, 28, 92, 96]),   # the 32-channel stem layers: strips, deep strips
    (_lib.GV_F16, (3, 3), 1, "SAME", 32, 64, 6, 41, 70, [93]),                # ... at 64 output channels
    (_lib.GV_BF16, (3, 3), 2, "VALID", 3, 32, 8, 96, 96, [0]),               # Conv2d_1a: stem rows
    (_lib.GV_F16, (7, 7), 2, "SAME", 3, 64, 4, 64, 64, [0]),                 # ResNet conv1 (explicit pad 3 == SAME here)
    (_lib.GV_BF16, (1, 1), 1, "SAME", 20, 36, 6, 9, 9, [0]),                 # channels not a multiple of 8: fp32-MFMA kernel on typed loads
    (_lib.GV_F32, (3, 3), 1, "SAME", 64, 96, 6, 17, 17, [0]),                # fp32 storage: conv_wgrad2_f32
    (_lib.GV_F32, (3, 3), 2, "VALID", 3, 32, 8, 321, 321, [0]),              # fp32 storage, stem: the direct kernel (M >= 200000)
    (_lib.GV_F32, (1, 1), 1, "SAME", 7, 5, 3, 6, 6, [0]),                    # scalar reduce path (elements not a multiple of 4)
]


@pytest.mark.parametrize("dt,k,stride,padding,cin,cout,nb,ih,iw,cfgs", FORMS)
def test_ws_form_matches_autograd_and_is_bitwise_reproducible(dt, k, stride, padding, cin, cout, nb, ih, iw, cfgs):
    xd, dzd, ref, geo = case(dt, k, stride, padding, cin, cout, nb, ih, iw, seed=cin + cout)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    tol = 2e-4 if nb * ih * iw > 100000 else 3e-5          # (fp32 accumulation over the pixels: summation order only)
    image = k[0] * k[1] * cin * cout * 4
    noise = torch.randn(1 << 22, device=DEV)
    side = torch.cuda.Stream()
    for cfg in cfgs:
        rc, dw0 = run_ws(dt, k, stride, cin, cout, nb, ih, iw, geo, xd, dzd, cfg, ws)
        if rc == _lib.GV_E_UNSUPPORTED and cfg:
            continue                                       # (a tile configuration this geometry does not take)
        _lib.check(rc, "wgrad_ws cfg %d" % cfg)
        close(dw0.cpu() - 0.5, ref.cpu(), tol)
        for rep in range(3):
            ws.fill_(0xA5 if rep == 1 else 0)              # the workspace is scratch: its contents must not matter
            if rep == 2:                                   # competing work on another stream: a different dispatch order
                with torch.cuda.stream(side):
                    for _ in range(8):
                        noise.mul_(1.0001)
            rc, dw = run_ws(dt, k, stride, cin, cout, nb, ih, iw, geo, xd, dzd, cfg, ws)
            assert rc == 0 and torch.equal(dw, dw0), "cfg %d, repeat %d: not bitwise reproducible" % (cfg, rep)
        torch.cuda.synchronize()
    # a workspace for three images only (fewer slices than the launch wants), and for one (the un-split launch)
    for nimg in (3, 1):
        small = torch.empty(nimg * ((image + 15) // 16 * 16) + 16, dtype=torch.uint8, device=DEV)
        rc, a = run_ws(dt, k, stride, cin, cout, nb, ih, iw, geo, xd, dzd, 0, small)
        _lib.check(rc, "wgrad_ws, %d-image workspace" % nimg)
        close(a.cpu() - 0.5, ref.cpu(), tol)
        rc, b = run_ws(dt, k, stride, cin, cout, nb, ih, iw, geo, xd, dzd, 0, small)
        assert torch.equal(a, b)


def test_ws_form_rejects_bad_workspaces():
    xd, dzd, ref, geo = case(_lib.GV_BF16, (1, 1), 1, "SAME", 32, 32, 2, 8, 8)
    oh, ow, pt, pl = geo
    dw = torch.zeros(1, 1, 32, 32, device=DEV)
    d = _lib.ConvDesc(2, 8, 8, 32, 32, 1, 1, 1, 0, 0, oh, ow, 32, 32, 0, 0, 0, _lib.GV_BF16, 0, 0, 0, 0)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    assert lib().gv_conv2d_wgrad_ws(C.byref(d), xd.data_ptr(), dzd.data_ptr(), 32, dw.data_ptr(), None, 1 << 20, st()) == _lib.GV_E_BADARG
    assert lib().gv_conv2d_wgrad_ws(C.byref(d), xd.data_ptr(), dzd.data_ptr(), 32, dw.data_ptr(), ws.data_ptr() + 4, 1 << 19, st()) == _lib.GV_E_BADARG
    assert lib().gv_conv2d_wgrad_ws(C.byref(d), xd.data_ptr(), dzd.data_ptr(), 32, dw.data_ptr(), ws.data_ptr(), -1, st()) == _lib.GV_E_BADARG


def _steps(backbone, storage, S, steps, autotune=False):
    N, V, C_, G = 4, 3, 5, 5
    eng = TrainGVCNN(backbone, N, V, S, S, C_, G, device=DEV, num_bins=G, storage=storage, seed=3)
    g = torch.Generator(device=DEV).manual_seed(9)
    losses = []
    for it in range(steps):
        x = torch.rand(N, V, S, S, 3, generator=g, device=DEV) - 0.5
        y = torch.randint(0, C_, (N,), generator=g, device=DEV)
        eng.forward(x, y, check=False)
        eng.backward()
        if it == 0:
            first = eng._flat_g.clone()
        eng.update_moving_averages(decay=0.9)
        eng.apply_momentum(0.01, 0.9, 1e-4)
        losses.append(eng.loss.clone())
    moving = torch.cat([v.reshape(-1) for k, v in sorted(eng.params.items()) if "moving_" in k])
    return first, eng._flat_g.clone(), eng._flat_p.clone(), torch.cat(losses), moving


@pytest.mark.parametrize("backbone,S", [("inception_v3", 96), ("resnet_v2_50", 64)])
@pytest.mark.parametrize("storage", ["bf16", "f32"])
def test_training_steps_are_bitwise_reproducible(backbone, S, storage):
    """Two engines, same seeds, four Momentum steps: first-step gradients, last-step gradients, variables, losses and
    moving statistics are the SAME BITS (folded BatchNorm sums, parity-class data gradients, pool -> BN pairs and the
    slice-ordered filter gradients all included)."""
    a = _steps(backbone, storage, S, 4)
    b = _steps(backbone, storage, S, 4)
    for name, u, v in zip(("first-step gradients", "last-step gradients", "variables", "losses", "moving statistics"), a, b):
        assert torch.equal(u, v), "%s %s: %s differ between two runs (max |d| %.3e)" % (
            backbone, storage, name, float((u - v).abs().max()))
    assert bool(torch.isfinite(a[3]).all())


@pytest.mark.parametrize("storage", ["bf16", "f32"])
def test_residual_gradient_aliasing_is_bitwise_neutral(storage):
    """ResNet-v2's `shortcut + residual` (nets/resnet_v2.py:91) sends dy to both addends.  TrainGVCNN lets the shortcut's
    gradient SHARE dy's buffer where the unit's conv3 is its first contributor (alias_residual_grad) instead of copying
    dy into a buffer of its own: 0 + dy is exact, so every gradient must keep its bits."""
    N, V, S, C_, G = 3, 2, 96, 5, 5
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.rand(N, V, S, S, 3, generator=g, device=DEV) - 0.5
    y = torch.randint(0, C_, (N,), generator=g, device=DEV)
    outs = []
    for alias in (True, False):
        eng = TrainGVCNN("resnet_v2_50", N, V, S, S, C_, G, device=DEV, num_bins=G, storage=storage, seed=3)
        eng.alias_residual_grad = alias
        for _ in range(2):                                  # (the second pass re-uses whatever the first one aliased)
            eng.forward(x, y, check=False)
            eng.backward()
        outs.append(eng._flat_g.clone())
        shared = sum(1 for op in eng.plan.ops if op["kind"] == "conv" and op.get("res") is not None and
                     eng.grad[op["res"].vbuf] is eng.grad[op["y"].vbuf])
        assert shared == (16 if alias else 0)               # every unit of the four blocks
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0
