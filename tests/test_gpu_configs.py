"""One oracle-parity test per BASELINE.json config AT THE BATCH bench.py RUNS (32 shapes per GPU on one GPU; the
per-GPU share of the 8-GPU configs: 48 views for configs[3], 80 views for configs[4]).

Tile choice, the XCD grid order, the > 64-images-per-tile paths of the folded BatchNorm sums and every 32-bit index
in the kernels depend on the batch; the other test files run the configs at 2 - 8 shapes.  The CPU oracle cannot push
384 - 640 images through a backbone in seconds, so at full size:

  * the HIP path runs the whole batch; SAMPLED images (first, one in the middle, last) go through the oracle backbone
    (folded == per-view in inference mode, oracle/model.py) and both taps are compared at the path's tolerance: 1e-3
    for fp32 (north_star), the storage-rounding bound of tests/test_gpu_lowp.py for 16-bit storage;
  * the oracle's grouping head (nets/model.py:44-102,163-164) runs on the device's descriptors of the WHOLE batch, the
    oracle's group_scheme / group_weight (nets/model.py:16-41) on the device's scores: integers bit-exact;
  * configs[2] as written (bf16 forward + backward): with BatchNorm frozen on its moving statistics the step is
    per-shape independent, so ONE sampled shape's contribution to the filter gradients is isolated (every other row of
    dlogits zeroed) and compared with the oracle's autograd through that shape alone under the batch's scheme.
configs[0] (the reference's own CPU-runnable case, batch 2) is tests/test_gpu_model.py::test_config_c1_plumbing_case.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvcnn_tf_amd as gv                       # noqa: E402
from gvcnn_tf_amd.training import TrainGVCNN     # noqa: E402
from oracle import grouping as OG                # noqa: E402
from oracle import model as OM                   # noqa: E402
from oracle import train as OT                   # noqa: E402

DEV = "cuda:0"


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_close(actual, desired, rtol=1e-3, atol_rel=1e-5):
    actual, desired = np.asarray(actual), np.asarray(desired)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol_rel * max(float(np.abs(desired).max()), 1e-30))


def engine(backbone, N, V, size, C, G, **kw):
    eng = gv.GVCNN(backbone, N, V, size, size, C, G, device=DEV, num_bins=G, **kw)
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C, seed=3, spread_scores=True)
    eng.plan.bind(P)
    eng.set_head(Hd)
    return eng, P, Hd


def check_forward(eng, P, Hd, x, backbone, V, G, fp32, ulp=0.0):
    """Sampled images against the oracle backbone; the oracle head and the oracle's integer grouping on device values."""
    N, size = x.shape[0], x.shape[2]
    scores, S, logits = eng.forward(x.to(DEV))
    torch.cuda.synchronize()
    F = eng.final_view_descriptors()
    R = eng.raw_view_descriptors()
    Ff, Rf = F.reshape(N * V, *F.shape[2:]), R.reshape(N * V, *R.shape[2:])
    flat = x.reshape(N * V, size, size, 3)
    for b in (0, (N * V) // 2 - 1, N * V - 1):
        ep = OM.run_backbone(backbone, flat[b:b + 1], P)
        fo, ro = ep[eng.plan.final_tap][0].numpy(), ep[eng.plan.raw_tap][0].numpy()
        if fp32:
            assert_close(Ff[b].float().cpu().numpy(), fo)
            assert_close(Rf[b].float().cpu().numpy(), ro)
        else:
            bound = 3e-2 if ulp > 2.0 ** -10 else 4e-3
            assert rel_l2(Ff[b].float().cpu().numpy(), fo) < bound, (b, rel_l2(Ff[b].float().cpu().numpy(), fo))
            assert rel_l2(Rf[b].float().cpu().numpy(), ro) < bound
    # integers from the device's scores through the ORACLE's group_scheme / group_weight: bit-exact
    sc = scores.float().cpu().numpy().astype(np.float32)
    o_scheme = OG.group_scheme([sc], G, V, G)
    o_weight = OG.group_weight(o_scheme)
    assert eng.scheme.cpu().numpy().tolist() == o_scheme.tolist()
    assert eng.weight.cpu().numpy().tolist() == o_weight.tolist()
    assert float(eng.weight.sum()) == G + V
    # the oracle's head on the device descriptors of the whole batch
    oS, oL = OG.grouping_head([F[:, v].float().cpu().numpy() for v in range(V)], o_scheme, o_weight,
                              Hd["dense_%d/kernel" % V].numpy(), Hd["dense_%d/bias" % V].numpy())
    Sn, Ln = S.float().cpu().numpy(), logits.float().cpu().numpy()
    if fp32:
        np.testing.assert_allclose(Sn, oS, rtol=1e-6, atol=1e-6 * float(np.abs(oS).max()))
        assert_close(Ln, oL)
    else:
        np.testing.assert_allclose(Sn, oS, rtol=1.01 * ulp, atol=1e-4 * float(np.abs(oS).max()))   # one rounding of S
        assert rel_l2(Ln, oL) < 2 * ulp
    assert np.isfinite(Ln).all()
    return scores, S, logits


def views(N, V, size, seed):
    return torch.rand(N, V, size, size, 3, generator=torch.Generator().manual_seed(seed)) - 0.5


def test_c2_fp32_forward_at_the_bench_batch():
    """configs[1]: 32 shapes x 12 views x 224^2, Inception-v3, G = 7, fp32 storage on the bf16x3 arithmetic, tiles
    autotuned — the step bench.py times — at 1e-3 of the oracle."""
    N, V, size, C, G = 32, 12, 224, 10, 7
    eng, P, Hd = engine("inception_v3", N, V, size, C, G, math="bf16x3")
    x = views(N, V, size, seed=4)
    eng.plan.autotune(views(N, V, size, seed=5).view(N * V, size, size, 3).to(DEV), iters=1)
    _, S, L = check_forward(eng, P, Hd, x, "inception_v3", V, G, fp32=True)
    S1, L1 = S.clone(), L.clone()
    _, S2, L2 = eng.forward(x.to(DEV))
    assert torch.equal(S1, S2) and torch.equal(L1, L2)                          # bitwise repeatable at this size too


@pytest.mark.parametrize("name,backbone,V,size,G,ty,N", [
    ("c3 forward", "inception_v3", 12, 224, 7, "bf16", 32),
    ("c4", "resnet_v2_50", 12, 224, 10, "bf16", 32),
    ("c4, one GPU's 48 views of the 8-GPU job", "resnet_v2_50", 12, 224, 10, "bf16", 4),
    ("c5", "inception_v3", 20, 299, 10, "f16", 32),
    ("c5, one GPU's 80 views of the 8-GPU job", "inception_v3", 20, 299, 10, "f16", 4)])
def test_16bit_configs_forward_at_the_bench_batch(name, backbone, V, size, G, ty, N):
    """configs[2] (forward), [3], [4] on their storage types at 32 shapes per GPU and at the per-GPU share of the 8-GPU
    jobs; bound: the storage rounding of every layer (3e-2 bf16 / 4e-3 fp16 in relative L2 against the fp32 oracle)."""
    eng, P, Hd = engine(backbone, N, V, size, 40, G, storage=ty)
    x = views(N, V, size, seed=6)
    if N == 32 and backbone == "inception_v3" and V == 12:
        # the c3 forward case runs on AUTOTUNED tiles, as the bench does: the default heuristic never picks a
        # wave-specialised tile, so only a tuned plan reaches conv_ws.hip (and its pipelined epilogue) inside the network
        chosen = eng.plan.autotune(views(N, V, size, seed=5).view(N * V, size, size, 3).to(DEV), iters=1)
        assert len(chosen) > 40
    check_forward(eng, P, Hd, x, backbone, V, G, fp32=False, ulp=2.0 ** -8 if ty == "bf16" else 2.0 ** -11)


def _backward_at_the_bench_batch(storage, cls_tol, filt_tol, filt_cos, stem_tol, stem_cos):
    """configs[2] as written, at 32 shapes: bf16 forward + backward of the whole batch with frozen statistics; the
    contribution of ONE shape to the gradients (the other 31 rows of dlogits zeroed: the mean CE loss is a sum over
    shapes and, with BatchNorm on its moving statistics, so is every gradient) against the oracle's autograd through that
    shape's 12 views under the batch's scheme and weights, divided by the batch size.  Bounds as in
    test_bf16_step_with_frozen_statistics_tracks_the_fp32_step (8 mantissa bits per stored tensor, ~100 roundings deep):
    classifier 2e-2, sampled filters 1.5e-1 in relative L2 with cosine >= 0.98 (the stem's: 3.5e-1 / 0.93)."""
    backbone, N, V, size, C, G = "inception_v3", 32, 12, 224, 40, 7
    probe = TrainGVCNN(backbone, 1, V, size, size, C, G, device=DEV, num_bins=G)
    P = gv.params.init_backbone_params(probe.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, probe.raw.c, probe.final.c, C, seed=3, spread_scores=True)
    del probe
    x = views(N, V, size, seed=7)
    labels = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(8))
    eng = TrainGVCNN(backbone, N, V, size, size, C, G, backbone_params=P, head_params=Hd, device=DEV, num_bins=G,
                     storage=storage, frozen_bn=True)
    n = 19
    _, _, logits, loss = eng.forward(x.to(DEV), labels)
    assert bool(torch.isfinite(loss).all())
    keep = eng.dlogits[n].clone()
    eng.dlogits.zero_()
    eng.dlogits[n] = keep
    grads = {k: v.clone() for k, v in eng.backward().items()}
    again = eng.backward()                                                      # same bits on a second pass
    assert all(torch.equal(grads[k], again[k]) for k in grads)
    scheme, weight = eng.scheme.cpu().numpy(), eng.weight.cpu().numpy()
    o = OT.loss_and_grads(x[n:n + 1], labels[n:n + 1].numpy(), P, Hd, G, backbone, num_bins=G, frozen_bn=True,
                          scheme=scheme, weight=weight)
    kn, bn = "dense_%d/kernel" % V, "dense_%d/bias" % V
    assert rel_l2(logits[n].cpu().numpy(), o["logits"][0]) < cls_tol
    for k in (kn, bn):
        assert rel_l2(grads[k].cpu().numpy() * N, o["grads"][k].numpy()) < cls_tol, k
    sampled = ["InceptionV3/Mixed_7c/Branch_0/Conv2d_0a_1x1/weights", "InceptionV3/Mixed_7a/Branch_1/Conv2d_0c_7x1/weights",
               "InceptionV3/Mixed_6c/Branch_2/Conv2d_0d_7x1/weights", "InceptionV3/Mixed_6a/Branch_0/Conv2d_1a_1x1/weights",
               "InceptionV3/Mixed_5c/Branch_1/Conv_1_0c_5x5/weights", "InceptionV3/Mixed_5b/Branch_3/Conv2d_0b_1x1/weights",
               "InceptionV3/Conv2d_4a_3x3/weights", "InceptionV3/Conv2d_2b_3x3/weights", "InceptionV3/Conv2d_1a_3x3/weights"]
    sampled = [k for k in sampled if k in o["grads"]] or sorted(k for k in o["grads"] if k.endswith("/weights"))[::11]
    assert len(sampled) >= 5
    report = []
    for k in sampled:
        a, d = grads[k].double().cpu().flatten() * N, o["grads"][k].double().flatten()
        cos = float((a @ d) / (a.norm() * d.norm()))
        rel = float((a - d).norm() / d.norm())
        report.append("%s: relative L2 %.3f, cosine %.4f" % (k.split("/", 1)[1], rel, cos))
        # the stem's filters sit ~100 roundings deep in the backward pass and carry the largest share of the error
        # (test_bf16_step_with_frozen_statistics_tracks_the_fp32_step: 37 - 42 % of the total): their own bound
        stem = k.split("/")[1].startswith("Conv2d_")
        assert rel < (stem_tol if stem else filt_tol) and cos > (stem_cos if stem else filt_cos), report[-1]
    print("c3 backward at 32 shapes (%s storage), shape %d against the oracle:\n  " % (storage, n) + "\n  ".join(report))


def test_c3_bf16_backward_at_the_bench_batch():
    """configs[2] as written (bf16 storage): the bounds of 8 mantissa bits per stored tensor, ~100 roundings deep."""
    _backward_at_the_bench_batch("bf16", cls_tol=2e-2, filt_tol=1.5e-1, filt_cos=0.98, stem_tol=3.5e-1, stem_cos=0.93)


def test_c3_geometry_backward_at_the_bench_batch_on_fp32_storage():
    """The same geometry, batch, tiles-per-launch, image counts per tile and 32-bit index ranges on fp32 STORAGE
    (bf16x3 products, fp32 accumulation): every sampled filter gradient INCLUDING the stem's within 1e-2 in relative L2
    of the oracle — so what the bf16 test above tolerates at the stem (35 %) is storage rounding, not a kernel that
    misbehaves at this batch (reference: train.py:145,166-187 computes these gradients in fp32)."""
    _backward_at_the_bench_batch("f32", cls_tol=2e-3, filt_tol=1e-2, filt_cos=0.9999, stem_tol=1e-2, stem_cos=0.9999)
