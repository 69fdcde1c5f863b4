"""BatchNorm -> ReLU -> max pool 3x3/2 as pool -> BatchNorm -> ReLU (gv_bn_bwd_coeffs_t, gv_pool2d_bwd_argmax_bn; the
engine's fuse_bn_pool).  Reference: the stem of nets/inception_v3.py:107-128 (Conv2d_2b -> MaxPool_3a, Conv2d_4a ->
MaxPool_5a) under the arg scope of nets/inception_utils.py:36-70 (train-mode batch_norm without gamma, ReLU).

With a positive scale BN + ReLU are monotone, so relu(bn(max z)) = max relu(bn(z)) element for element, and only a
window's winner carries a gradient: the backward sums are sums over the pooled tensors.  Checked here against a float64
restatement of the un-fused composition, and on the whole engine against the un-fused step.
"""
import ctypes as C

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
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("ty", ["bf16", "f16"])
@pytest.mark.parametrize("nb,h,w,c,V", [(6, 21, 21, 64, 3), (4, 12, 17, 192, 2), (5, 9, 9, 32, 5), (8, 35, 34, 64, 4)])
def test_pool_then_batchnorm_equals_batchnorm_then_pool(ty, nb, h, w, c, V):
    code, td = TYPES[ty]
    g = torch.Generator().manual_seed(nb * 100 + c)
    z = (torch.randn(nb, h, w, c, generator=g) * 1.5 + 0.2).to(td).to(DEV)
    beta = (torch.randn(c, generator=g) * 0.3).to(DEV)
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    dy = torch.randn(nb, oh, ow, c, generator=g).to(td).to(DEV)
    L = lib()
    eps = 1e-3
    # ---- forward sums over the whole z, then pool z, then finalize + apply on the pooled tensor
    acc = torch.zeros(V * c * 2, dtype=torch.float64, device=DEV)
    counts = torch.tensor([(nb - gi + V - 1) // V * h * w for gi in range(V)], dtype=torch.int32, device=DEV)
    _lib.check(L.gv_bn_sums_grouped_t(z.data_ptr(), nb, h * w, c, c, V, acc.data_ptr(), code, st()), "sums")
    d = _lib.PoolDesc(nb, h, w, c, c, 3, 3, 2, 0, 0, oh, ow, c, _lib.GV_POOL_MAX, code)
    pz = torch.empty(nb, oh, ow, c, dtype=td, device=DEV)
    arg = torch.empty(nb, oh, ow, c, dtype=torch.uint8, device=DEV)
    _lib.check(L.gv_pool2d_fwd_argmax(C.byref(d), z.data_ptr(), pz.data_ptr(), arg.data_ptr(), st()), "pool z")
    stat = {k: torch.empty(V * c, device=DEV) for k in ("mean", "var", "inv", "scale", "shift")}
    y = torch.empty_like(pz)
    _lib.check(L.gv_bn_finalize_apply_grouped_t(acc.data_ptr(), counts.data_ptr(), None, beta.data_ptr(), eps, pz.data_ptr(),
                                                nb, oh * ow, c, c, V, 1, y.data_ptr(), c, stat["mean"].data_ptr(),
                                                stat["var"].data_ptr(), stat["inv"].data_ptr(), stat["scale"].data_ptr(),
                                                stat["shift"].data_ptr(), code, st()), "finalize + apply (pooled)")
    # the un-fused order with the same statistics: normalise all of z, then pool
    acc2 = torch.zeros_like(acc)
    _lib.check(L.gv_bn_sums_grouped_t(z.data_ptr(), nb, h * w, c, c, V, acc2.data_ptr(), code, st()), "sums")
    a_full, y_ref = torch.empty_like(z), torch.empty_like(pz)
    stat2 = {k: torch.empty(V * c, device=DEV) for k in stat}
    _lib.check(L.gv_bn_finalize_apply_grouped_t(acc2.data_ptr(), counts.data_ptr(), None, beta.data_ptr(), eps, z.data_ptr(),
                                                nb, h * w, c, c, V, 1, a_full.data_ptr(), c, stat2["mean"].data_ptr(),
                                                stat2["var"].data_ptr(), stat2["inv"].data_ptr(), stat2["scale"].data_ptr(),
                                                stat2["shift"].data_ptr(), code, st()), "finalize + apply")
    _lib.check(L.gv_pool2d_fwd(C.byref(d), a_full.data_ptr(), y_ref.data_ptr(), st()), "pool a")
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref), "relu(bn(max z)) != max relu(bn(z))"
    for k in stat:
        assert torch.equal(stat[k], stat2[k]), k
    assert float(stat["scale"].min()) > 0

    # ---- backward: sums over the pooled tensors -> coefficients -> gather + mask + dz in the pool's backward kernel
    accb = torch.zeros(V * c * 2, dtype=torch.float64, device=DEV)
    _lib.check(L.gv_bn_relu_bwd_sums_grouped_t(dy.data_ptr(), c, None, c, pz.data_ptr(), c, stat["mean"].data_ptr(),
                                               stat["inv"].data_ptr(), nb, oh * ow, c, V, accb.data_ptr(),
                                               stat["scale"].data_ptr(), stat["shift"].data_ptr(), code, st()), "bwd sums (pooled)")
    ca, cb, cc = (torch.empty(V * c, device=DEV) for _ in range(3))
    dbeta = torch.zeros(c, device=DEV)
    _lib.check(L.gv_bn_bwd_coeffs_t(accb.data_ptr(), counts.data_ptr(), stat["mean"].data_ptr(), stat["inv"].data_ptr(), None,
                                    c, V, 0, ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), dbeta.data_ptr(), None, st()), "coeffs")
    dz = torch.full_like(z, 7.0)
    _lib.check(L.gv_pool2d_bwd_argmax_bn(C.byref(d), arg.data_ptr(), dy.data_ptr(), c, z.data_ptr(), c, V, ca.data_ptr(),
                                         cb.data_ptr(), cc.data_ptr(), stat["scale"].data_ptr(), stat["shift"].data_ptr(),
                                         dz.data_ptr(), c, st()), "pool_bwd + bn_bwd")
    torch.cuda.synchronize()

    # float64 restatement of the un-fused backward (MaxPoolGrad by the recorded winner -> ReLU mask -> batch_norm gradient
    # with per-view statistics: image n belongs to view n % V, model.py:153-158)
    zd, dyd = z.double(), dy.double()
    gfull = torch.zeros_like(zd)
    for t in range(9):
        kh, kw = divmod(t, 3)
        gfull[:, kh:kh + 2 * oh:2, kw:kw + 2 * ow:2, :] += dyd * (arg == t)
    grp = torch.arange(nb, device=DEV) % V
    scale = stat["scale"].double().reshape(V, c)[grp][:, None, None, :]
    shift = stat["shift"].double().reshape(V, c)[grp][:, None, None, :]
    mean = stat["mean"].double().reshape(V, c)[grp][:, None, None, :]
    inv = stat["inv"].double().reshape(V, c)[grp][:, None, None, :]
    gm = gfull * ((zd * scale + shift).float() > 0)
    zh = (zd - mean) * inv
    want = torch.empty_like(zd)
    for gi in range(V):
        sel = grp == gi
        n = float(counts[gi])
        s0, s1 = gm[sel].sum((0, 1, 2)), (gm[sel] * zh[sel]).sum((0, 1, 2))
        want[sel] = inv[sel] * (gm[sel] - s0 / n - zh[sel] * s1 / n)
    err = float((dz.double() - want).abs().max())
    tol = (2.0 ** -8 if ty == "bf16" else 2.0 ** -11) * float(want.abs().max()) * 1.5
    assert err <= tol, (err, tol)
    db_want = gm.sum((0, 1, 2))
    assert float((dbeta.double() - db_want).abs().max()) <= 1e-5 * float(db_want.abs().max()) + 1e-6
    # the sums over the pooled tensors ARE the sums over the full tensors (every other element has g = 0)
    s0 = torch.stack([gm[grp == gi].sum((0, 1, 2)) for gi in range(V)])
    got = accb.reshape(V, c, 2)[..., 0]
    assert float((got - s0).abs().max()) <= 1e-6 * float(s0.abs().max()) + 1e-9


@pytest.mark.parametrize("size", [139, 171])
def test_engine_with_pooled_batchnorm_reproduces_the_unfused_step(size):
    """Whole bf16 Inception step, fuse_bn_pool on / off.  The forward pass is bit-identical (hence the loss and every
    activation gradient that does not pass the two pairs); below the pairs the gradient routing may differ where two
    DIFFERENT z round to the same bf16 activation (the un-fused pool sees a tie and elects the first tap; the fused pool
    elects the larger z, as the fp32 reference would)."""
    N, V = 4, 3
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(N, V, size, size, 3, generator=g) - 0.5).to(DEV)
    labels = torch.randint(0, 10, (N,), generator=g).to(DEV)
    res = {}
    for fuse in (False, True):
        eng = TrainGVCNN("inception_v3", N, V, size, size, 10, 5, device=DEV, num_bins=5, storage="bf16", seed=4)
        eng.fuse_bn_pool = fuse
        eng.forward(x, labels, check=False)
        eng.backward()
        torch.cuda.synchronize()
        pairs = [op for op in eng.plan.ops if op.get("pool_after") is not None]
        assert len(pairs) == 2, [op["name"] for op in pairs]
        pools = [op["pool_after"] for op in pairs]
        res[fuse] = dict(loss=float(eng.loss), flat=eng._flat_g.clone(), eng=eng,
                         pooled=[eng.view(p["y"]).clone() for p in pools],
                         dz=[eng.view(b["x"], grad=True).clone() for b in pairs],
                         stats=[{k: v.clone() for k, v in b["stat"].items()} for b in pairs])
    a, b = res[False], res[True]
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    for pa, pb in zip(a["pooled"], b["pooled"]):
        assert torch.equal(pa, pb)
    for sa, sb in zip(a["stats"], b["stats"]):
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    # gradient of the second pair's input: everything above it is identical, so the two forms differ by ties only
    da, db = a["dz"][1].double(), b["dz"][1].double()
    cos = float((da * db).sum() / (da.norm() * db.norm()))
    assert cos > 0.98, cos
    ga, gb = a["flat"].double(), b["flat"].double()
    cosw = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    print("pool->bn vs bn->pool: loss %.6f, cosine dz(Conv2d_4a) %.5f, all filter gradients %.5f" % (a["loss"], cos, cosw))
    assert cosw > 0.95, cosw


def test_pool_backward_with_the_batchnorm_tail_rejects_what_it_cannot_do():
    """gv_pool2d_bwd_argmax_bn exists for 3x3 / stride 2 / VALID windows on 16-bit storage with 16-byte aligned coefficient
    tables: anything else is refused (GV_E_UNSUPPORTED / GV_E_BADARG) and nothing is written."""
    code, td = TYPES["bf16"]
    nb, h, w, c, V = 2, 9, 9, 32, 2
    oh = ow = 4
    z = torch.randn(nb, h, w, c).to(td).to(DEV)
    dy = torch.randn(nb, oh, ow, c).to(td).to(DEV)
    arg = torch.zeros(nb, oh, ow, c, dtype=torch.uint8, device=DEV)
    tab = torch.ones(5 * V * c + 4, device=DEV)
    t = [tab[i * V * c:(i + 1) * V * c] for i in range(5)]
    dz = torch.full_like(z, 3.0)
    L = lib()

    def call(d, tables=t):
        return L.gv_pool2d_bwd_argmax_bn(C.byref(d), arg.data_ptr(), dy.data_ptr(), c, z.data_ptr(), c, V, tables[0].data_ptr(),
                                         tables[1].data_ptr(), tables[2].data_ptr(), tables[3].data_ptr(), tables[4].data_ptr(),
                                         dz.data_ptr(), c, st())
    good = _lib.PoolDesc(nb, h, w, c, c, 3, 3, 2, 0, 0, oh, ow, c, _lib.GV_POOL_MAX, code)
    assert call(good) == 0
    torch.cuda.synchronize()
    dz.fill_(3.0)
    stride1 = _lib.PoolDesc(nb, h, w, c, c, 3, 3, 1, 1, 1, h, w, c, _lib.GV_POOL_MAX, code)
    assert call(stride1) != 0
    f32 = _lib.PoolDesc(nb, h, w, c, c, 3, 3, 2, 0, 0, oh, ow, c, _lib.GV_POOL_MAX, _lib.GV_F32)
    assert call(f32) != 0
    misaligned = [tab[1 + i * V * c:1 + (i + 1) * V * c] for i in range(5)]      # 4-byte, not 16-byte aligned
    assert call(good, misaligned) != 0
    torch.cuda.synchronize()
    assert float(dz.float().min()) == 3.0 and float(dz.float().max()) == 3.0, "a refused call wrote to dz"
