"""End-to-end parity of the HIP path against the CPU oracle (reference-shaped: V sequential
backbone calls at batch N, host numpy grouping), for both backbones, plus size-independent
properties at BASELINE.json's full sizes.

Tolerance: 1e-3 on descriptors / logits (north_star, fp32); group indices bit-exact given the
scores (the synthetic scorer keeps scores >= 1e-3 away from bin edges, SURVEY hard part (v))."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvcnn_tf_amd as gv                       # noqa: E402
from oracle import backbone as OB                # noqa: E402
from oracle import grouping as OG                # noqa: E402
from oracle import model as OM                   # noqa: E402

DEV = "cuda:0"


def assert_close(actual, desired, rtol=1e-3, atol_rel=1e-5):
    """North_star tolerance 1e-3 (fp32), stated relative to the tensor's scale: an element passes
    when |a-d| <= rtol*|d| + atol_rel*max|d|.  (ResNet-v2 activations with random weights reach
    |x| ~ 1e3 in the residual stream; fp32 summation-order noise there is ~1e-5 of the scale, so a
    fixed absolute 1e-3 would test the weights' scale, not the kernels.)"""
    actual, desired = np.asarray(actual), np.asarray(desired)
    scale = max(float(np.abs(desired).max()), 1e-30)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol_rel * scale)


def make_engine(backbone, N, V, H, W, C, G, **kw):
    eng = gv.GVCNN(backbone, N, V, H, W, C, G, device=DEV, **kw)
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C, seed=3, spread_scores=True)
    eng.plan.bind(P)
    eng.set_head(Hd)
    return eng, P, Hd


def views(N, V, H, W, seed=0):
    return torch.rand(N, V, H, W, 3, generator=torch.Generator().manual_seed(seed)) - 0.5   # train_data.py:101


@pytest.mark.parametrize("backbone,size", [("inception_v3", 75), ("inception_v3", 107), ("resnet_v2_50", 64),
                                           ("resnet_v2_50", 97)])
def test_backbone_endpoints_vs_oracle(backbone, size):
    """Every kept tap of the folded backbone equals the oracle's per-image result."""
    N, V = 2, 3
    eng, P, _ = make_engine(backbone, N, V, size, size, 10, 10)
    x = views(N, V, size, size)
    eng.run_backbone(x.to(DEV))
    torch.cuda.synchronize()
    ep = OM.folded_backbone(x, P, backbone)
    raw_o, fin_o = ep[eng.plan.raw_tap].numpy(), ep[eng.plan.final_tap].numpy()
    raw = eng.raw_view_descriptors().reshape(raw_o.shape).cpu().numpy()
    fin = eng.final_view_descriptors().reshape(fin_o.shape).cpu().numpy()
    assert np.isfinite(fin).all() and np.abs(fin_o).max() > 1e-3
    assert_close(raw, raw_o)
    assert_close(fin, fin_o)


@pytest.mark.parametrize("backbone,size,G", [("resnet_v2_50", 64, 10), ("inception_v3", 75, 10)])
def test_gvcnn_fused_vs_reference_shaped_oracle(backbone, size, G):
    N, V, C = 2, 6, 10
    eng, P, Hd = make_engine(backbone, N, V, size, size, C, G)
    x = views(N, V, size, size, seed=1)
    scores, S, logits = eng.forward(x.to(DEV))
    o_scores, o_S, o_logits, o_scheme, o_weight = OM.gvcnn(x, C, P, Hd, G, backbone)
    np.testing.assert_allclose(scores.cpu().numpy(), np.array(o_scores), rtol=1e-3, atol=1e-4)
    assert eng.scheme.cpu().numpy().tolist() == o_scheme.tolist()           # integer path: exact
    assert eng.weight.cpu().numpy().tolist() == o_weight.tolist()
    assert len(set(eng.gidx.cpu().tolist())) > 1                            # scores spread over bins
    assert_close(S.cpu().numpy(), o_S)
    assert_close(logits.cpu().numpy(), o_logits)

    # the two-phase protocol of train.py:264-288 with the reference-shaped host functions
    sc = eng.forward_phase1(x.to(DEV))
    host_scores = [np.array(sc.cpu().numpy())]
    g_scheme = gv.group_scheme(host_scores, G, V)
    g_weight = gv.group_weight(g_scheme)
    assert g_scheme.tolist() == o_scheme.tolist() and g_weight.tolist() == o_weight.tolist()
    S2, logits2 = eng.forward_phase2(g_scheme, g_weight)
    np.testing.assert_array_equal(S2.cpu().numpy(), S.cpu().numpy())
    np.testing.assert_array_equal(logits2.cpu().numpy(), logits.cpu().numpy())


def test_functional_surface_and_basic():
    """nets/model.py-shaped calls: gvcnn(inputs, num_classes, group_scheme, group_weight, ...), basic()."""
    N, V, size, C, G = 2, 4, 64, 5, 10
    eng, P, Hd = make_engine("resnet_v2_50", N, V, size, size, C, G)
    gv.configure(backbone="resnet_v2_50", backbone_params=P, head_params=Hd)
    x = views(N, V, size, size, seed=2)
    o_scores, o_S, o_logits, o_scheme, o_weight = OM.gvcnn(x, C, P, Hd, G, "resnet_v2_50")
    scores, S, logits = gv.gvcnn(x.to(DEV), C, o_scheme, o_weight, is_training=False)
    assert isinstance(scores, list) and len(scores) == V and scores[0].dim() == 0
    assert_close(S.cpu().numpy(), o_S)
    assert_close(logits.cpu().numpy(), o_logits)
    Sb, Lb = gv.basic(x.to(DEV), C, is_training=False)
    oSb, oLb = OM.basic(x, C, P, Hd, "resnet_v2_50")
    assert_close(Sb.cpu().numpy(), oSb)
    assert_close(Lb.cpu().numpy(), oLb)
    s3, S3, L3 = gv.gvcnn_fused(x.to(DEV), C, G)
    assert_close(S3.cpu().numpy(), o_S)
    # is_training defaults to True (model.py:105): batch-statistics BN per view, like the reference graph
    tr = OM.gvcnn(x, C, P, Hd, G, "resnet_v2_50", is_training=True)
    ts, tS, tL = gv.gvcnn(x.to(DEV), C, tr[3], tr[4])
    assert_close(tS.cpu().numpy(), tr[1], rtol=5e-3, atol_rel=1e-3)
    assert_close(np.array([float(v) for v in ts]), np.array(tr[0]), rtol=5e-3, atol_rel=1e-3)
    with pytest.raises(IndexError):                                      # G=5 with scores up to 0.9 (D6)
        gv.gvcnn_fused(x.to(DEV), C, 5)


def test_config_c1_plumbing_case():
    """BASELINE.json configs[0]: ModelNet10, 6 views, 224x224, Inception-v3, batch 2 (CPU-runnable
    reference case).  num_groups=5 with the literal x10 binning overflows, so the bins
    here are num_bins=G=5 sub-ranges on both sides."""
    N, V, size, C, G = 2, 6, 224, 10, 5
    eng, P, Hd = make_engine("inception_v3", N, V, size, size, C, G, num_bins=G)
    x = views(N, V, size, size, seed=3)
    scores, S, logits = eng.forward(x.to(DEV))
    o_scores, o_S, o_logits, o_scheme, _ = OM.gvcnn(x, C, P, Hd, G, "inception_v3", num_bins=G)
    assert eng.scheme.cpu().numpy().tolist() == o_scheme.tolist()
    assert_close(S.cpu().numpy(), o_S)
    assert_close(logits.cpu().numpy(), o_logits)
    assert tuple(S.shape) == (N, 5, 5, 2048)


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_full_size_properties_config_c2(math):
    """configs[1] (12 views, 224, Inception-v3, G=7) at a size the oracle cannot finish in seconds:
    size-independent properties instead of element-wise comparison — in the exact fp32 chain and in the mode bench.py
    times (bf16x3 with three-plane intermediates, autotuned tiles), whose descriptors are also held against the exact
    chain's at this full size."""
    N, V, size, C, G = 8, 12, 224, 10, 7
    eng, P, Hd = make_engine("inception_v3", N, V, size, size, C, G, num_bins=G, math=math)
    if math == "bf16x3":
        eng.plan.autotune(views(N, V, size, size, seed=5).view(N * V, size, size, 3).to(DEV))
    x = views(N, V, size, size, seed=4).to(DEV)
    scores, S, logits = eng.forward(x)
    S1, L1 = S.clone(), logits.clone()
    F = eng.final_view_descriptors().clone()                   # [N,V,5,5,2048]
    # (1) determinism / idempotence: same input, bitwise same output
    scores2, S2, L2 = eng.forward(x)
    assert torch.equal(S1, S2) and torch.equal(L1, L2)
    # (2) batch independence of the backbone: image b alone gives the same descriptor rows
    eng1, _, _ = make_engine("inception_v3", 1, V, size, size, C, G, num_bins=G, math=math)
    eng1.run_backbone(x[3:4].contiguous())
    # (the two engines run different tiles — the one-shape plan is not tuned, the tuned one runs the wave-specialised kernels,
    # whose k-steps add the six plane products in another order: fp32 rounding through 94 layers, 2e-5 absolute at most)
    np.testing.assert_allclose(eng1.final_view_descriptors()[0].cpu().numpy(), F[3].cpu().numpy(),
                               rtol=1e-4, atol=5e-5)
    # (3) fusion bounds: S lies between min(min_v F, 1) and max(max_v F, 1) (convex combination of
    #     group maxima and the all-ones dummy of model.py:63)
    lo = torch.minimum(F.min(dim=1).values, torch.ones_like(S1))
    hi = torch.maximum(F.max(dim=1).values, torch.ones_like(S1))
    assert bool((S1 >= lo - 1e-5).all()) and bool((S1 <= hi + 1e-5).all())
    # (4) Σw = G + V  (model.py:33-39)
    assert float(eng.weight.sum()) == G + V
    # (5) permuting the shapes permutes the outputs (grouping is per batch via the mean score, so use
    #     the SAME scheme through phase 2)
    perm = torch.tensor([7, 6, 5, 4, 3, 2, 1, 0], device=DEV)
    Sp, Lp = eng.pool_fuse_classify(eng.scheme, eng.weight, F=F[perm])
    np.testing.assert_array_equal(Sp.cpu().numpy(), S1[perm].cpu().numpy())
    # (6) the oracle's grouping head on the device descriptors (cheap on CPU) matches
    oS, oL = OG.grouping_head([F[:, v].cpu().numpy() for v in range(V)], eng.scheme.cpu().numpy(),
                              eng.weight.cpu().numpy(), Hd["dense_%d/kernel" % V].numpy(),
                              Hd["dense_%d/bias" % V].numpy())
    np.testing.assert_allclose(S1.cpu().numpy(), oS, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(L1.cpu().numpy(), oL, rtol=1e-4, atol=1e-4)
    if math == "bf16x3":
        # (7) the benched arithmetic against the exact fp32 MFMA chain on the same full-size input: 1e-4 of the
        #     descriptor scale after 47 layers (the stated tolerance is 1e-3)
        e32, _, _ = make_engine("inception_v3", N, V, size, size, C, G, num_bins=G, math="f32")
        e32.run_backbone(x)
        F32 = e32.final_view_descriptors()
        assert float((F - F32).abs().max()) <= 1e-4 * float(F32.abs().max())


@pytest.mark.parametrize("backbone,size", [("inception_v3", 75), ("resnet_v2_50", 64)])
def test_bf16x3_math_matches_oracle_at_fp32_tolerance(backbone, size):
    """GV_MATH_BF16X3 (fp32 split into three bf16 planes on the bf16 MFMA) keeps fp32-level accuracy
    end to end: same oracle, same tolerance as the exact fp32 MFMA path."""
    N, V, C, G = 2, 6, 10, 10
    eng, P, Hd = make_engine(backbone, N, V, size, size, C, G, math="bf16x3")
    x = views(N, V, size, size, seed=1)
    scores, S, logits = eng.forward(x.to(DEV))
    o_scores, o_S, o_logits, o_scheme, o_weight = OM.gvcnn(x, C, P, Hd, G, backbone)
    assert eng.scheme.cpu().numpy().tolist() == o_scheme.tolist()
    assert_close(S.cpu().numpy(), o_S)
    assert_close(logits.cpu().numpy(), o_logits)
    eng32, _, _ = make_engine(backbone, N, V, size, size, C, G)
    _, S32, _ = eng32.forward(x.to(DEV))
    err_x3 = float(np.abs(S.cpu().numpy() - o_S).max())
    err_32 = float(np.abs(S32.cpu().numpy() - o_S).max())
    assert err_x3 < 4 * err_32 + 1e-6 * float(np.abs(o_S).max())      # same error class as exact fp32


@pytest.mark.parametrize("backbone,V,size,G,math", [("inception_v3", 20, 299, 10, "bf16x3"),
                                                     ("resnet_v2_50", 12, 224, 10, "bf16x3"),
                                                     ("resnet_v2_50", 12, 224, 10, "f32")])
def test_other_baseline_configs_properties(backbone, V, size, G, math):
    """BASELINE.json configs[3]/[4] geometries (ResNet-v2-50 12x224; Inception 20 views x 299, G=10) in
    fp32: size-independent checks, plus the oracle backbone on ONE image of the batch."""
    N, C = 2, 40
    eng, P, Hd = make_engine(backbone, N, V, size, size, C, G, math=math)
    x = views(N, V, size, size, seed=5)
    scores, S, logits = eng.forward(x.to(DEV))
    F = eng.final_view_descriptors().clone()
    assert tuple(F.shape[2:]) == ((8, 8, 2048) if size == 299 else (7, 7, 2048))
    assert float(eng.weight.sum()) == G + V and torch.isfinite(logits).all()
    # one image through the CPU oracle backbone (folded == per-view in inference mode)
    b = 7
    ep = OM.run_backbone(backbone, x.reshape(N * V, size, size, 3)[b:b + 1], P)
    assert_close(F.reshape(N * V, *F.shape[2:])[b].cpu().numpy(), ep[eng.plan.final_tap][0].numpy())
    assert_close(eng.raw_view_descriptors().reshape(N * V, *eng.raw_view_descriptors().shape[2:])[b].cpu().numpy(),
                 ep[eng.plan.raw_tap][0].numpy())
    # oracle grouping head on the device descriptors
    oS, oL = OG.grouping_head([F[:, v].cpu().numpy() for v in range(V)], eng.scheme.cpu().numpy(),
                              eng.weight.cpu().numpy(), Hd["dense_%d/kernel" % V].numpy(),
                              Hd["dense_%d/bias" % V].numpy())
    np.testing.assert_allclose(S.cpu().numpy(), oS, rtol=1e-6, atol=1e-6 * float(np.abs(oS).max()))
    assert_close(logits.cpu().numpy(), oL)


@pytest.mark.parametrize("weight_mode", ["count", "mean_score"])
@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_per_shape_grouping_vs_oracle(weight_mode, storage):
    """per-shape scores / schemes / weights / fusion.  Integer results bit-exact given the scores;
    the kernels are checked on the DEVICE's descriptors and responses (the backbone has its own tests)."""
    import ctypes as C_
    from gvcnn_tf_amd import _lib
    N, V, G, C = 3, 6, 10, 10
    eng, P, Hd = make_engine("resnet_v2_50", N, V, 64, 64, C, G, storage=storage, per_shape=True, weight_mode=weight_mode)
    x = views(N, V, 64, 64, seed=3)
    scores, S, logits = eng.forward(x.to(DEV))
    assert tuple(scores.shape) == (N, V)
    F = eng.final_view_descriptors().float().cpu().numpy()
    r = eng.r_img.cpu().numpy().reshape(N, V)
    o_scores, o_schemes, o_weights, o_S = OG.per_shape_grouping(F, r, G, weight_mode=weight_mode)
    np.testing.assert_allclose(scores.cpu().numpy(), o_scores, rtol=1e-5, atol=1e-6)
    # bins from the DEVICE scores (identical fp32 inputs => identical integers)
    dev_scores = scores.cpu().numpy()
    for n in range(N):
        sch = OG.group_scheme([dev_scores[n]], G, V)
        assert eng.scheme_ps[n].cpu().numpy().tolist() == sch.tolist()
        w = OG.group_weight(sch) if weight_mode == "count" else OG.group_weight_mean_score(sch, dev_scores[n])
        np.testing.assert_allclose(eng.weight_ps[n].cpu().numpy(), w, rtol=1e-6)
    tol = 1e-5 if storage == "f32" else 2.0 ** -7
    np.testing.assert_allclose(S.float().cpu().numpy(), o_S, rtol=tol, atol=tol * float(np.abs(o_S).max()))
    oL = OG.dense(OG.global_average_pool(S.float().cpu().numpy()), Hd["dense_%d/kernel" % V].numpy(),
                  Hd["dense_%d/bias" % V].numpy())
    np.testing.assert_allclose(logits.cpu().numpy(), oL, rtol=1e-4, atol=1e-4 * float(np.abs(oL).max()))


def test_bench_two_rank_control_flow_on_one_device():
    """`python bench.py --gpus 2` exactly as the driver types it (NO torch.distributed.run in front): the process
    launches its own two ranks, relays rank 0's single JSON line and returns their exit code.  The ranks share
    cuda:0 over a gloo group (control flow of the N>1 path: shape sharding, score + descriptor exchange,
    max-over-ranks timing) — RCCL itself needs two devices and is exercised by the driver's multi-GPU run."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--same-device", "--steps", "2", "--warmup", "1", "--shapes", "2", "--no-roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                      # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_views"] == 2 * 2 * 12 and j["scaling"] == "weak"
    assert j["config"]["world_size"] == 2                       # what the ranks saw
    assert j["value"] > 0 and j["config"]["exchange"] == "allgather"          # the north_star form is the default
    desc = 2 * 12 * 5 * 5 * 2048 * 4 + 2 * 12 * 4
    assert j["config"]["exchange_bytes_per_step"] == {"sent_per_rank": desc, "received_per_rank": desc}
    assert j["other_exchange"]["exchange"] == "scores" and j["other_exchange"]["value"] > 0


@pytest.mark.parametrize("preset,V,hw,esz", [("c4", 12, 7, 2), ("c5", 20, 8, 2)])
def test_bench_eight_rank_control_flow_of_the_8gpu_configs(preset, V, hw, esz):
    """BASELINE.json configs[3] / configs[4] as the driver's scaling run types them, `python bench.py --gpus 8`, at their
    per-GPU share (4 shapes = 48 / 80 views per rank): eight self-launched ranks over a gloo group on cuda:0 (control
    flow only: RCCL needs eight devices).  The line must carry what the ranks saw (world size 8), weak scaling over
    8 x 4 shapes, the exchange volume of the north_star form — every rank sends its scorer responses and final
    descriptors: 9.6 MB (c4) / 21 MB (c5) — and both ways of moving it (one collective, point-to-point sends)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # c4 with the tile tuning the driver's run has (ONE tuning, by rank 0, its table broadcast); c5 with default tiles
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--same-device", "--preset",
           preset, "--steps", "1", "--warmup", "1", "--shapes", "4", "--no-roofline", "--no-lanes"] + \
          ([] if preset == "c4" else ["--no-tune"])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["world_size"] == 8 and j["scaling"] == "weak"
    assert j["config"]["views_per_gpu"] == 4 * V and j["config"]["global_views"] == 8 * 4 * V and j["value"] > 0
    sent = 4 * V * 4 + 4 * V * hw * hw * 2048 * esz
    assert j["config"]["exchange"] == "allgather"
    assert j["config"]["exchange_bytes_per_step"] == {"sent_per_rank": sent, "received_per_rank": 7 * sent}
    assert abs(sent - (9.6e6 if preset == "c4" else 21e6)) < 0.05 * sent          # SURVEY §8e's message sizes
    assert j["other_exchange"]["exchange"] == "scores" and j["other_exchange"]["value"] > 0
    assert j["config"]["gather"].startswith("collective") and j["other_gather"]["gather"] == "direct"
    assert j["other_gather"]["value"] > 0
    # what the ranks ran on / with: eight ranks, one world size, ONE tile table; the call-by-call re-run is on the line
    facts = j["config"]["ranks"]
    assert [f["rank"] for f in facts] == list(range(8)) and all(f["world_size"] == 8 for f in facts)
    assert len({f["tile_table"] for f in facts}) == 1
    assert j["config"]["tiles_tuned_by"].startswith("rank 0" if preset == "c4" else "--no-tune")
    assert j["other_overlap"]["value"] > 0 and j["other_overlap"]["exchange_overlap"].startswith("none")


def test_bench_external_launcher_form_still_works():
    """The other form the contract names: the driver's own `python -m torch.distributed.run ... bench.py --gpus N`."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--same-device", "--steps", "2", "--warmup", "1", "--shapes", "2", "--no-roofline", "--exchange", "scores",
           "--no-other-exchange"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["exchange"] == "scores" and "other_exchange" not in j


@pytest.mark.parametrize("dp", ["views", "shapes", "hybrid"])
def test_two_rank_training_bench_on_bf16_storage(dp):
    """bench.py --train --preset c3 as two ranks on one device (gloo): the sharded training step on bf16 storage —
    gathers, gradient all-reduce, per-layer BN sum all-reduce (shapes) / the gathered moving-average update (views)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--same-device", "--train", "--preset", "c3", "--dp", dp, "--steps", "2", "--warmup", "1", "--shapes", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dtype"] == "bf16" and j["value"] > 0
    assert j["scaling"] == ("strong" if dp == "views" else "weak")


def test_eval_harness_metrics_and_protocols():
    """SURVEY §8 f4 (eval.py:94-99,146-215): argmax / confusion matrix / correct count against numpy (ties -> first
    maximum, out-of-range label ignored), mean-of-batch accuracy, and both calling protocols give the same result."""
    import ctypes as C_
    from gvcnn_tf_amd import _lib
    from gvcnn_tf_amd.evaluate import Evaluator
    rng = np.random.RandomState(3)
    n, c = 37, 10
    logits = rng.randn(n, c).astype(np.float32)
    logits[5, 2] = logits[5, 7] = logits[5].max() + 1.0           # tie: first maximum (index 2)
    labels = rng.randint(0, c, size=n).astype(np.int64)
    labels[9] = 12                                                  # out of range: counted nowhere
    ld, lab = torch.from_numpy(logits).to(DEV), torch.from_numpy(labels).to(DEV)
    pred = torch.empty(n, dtype=torch.int64, device=DEV)
    conf = torch.zeros(c, c, dtype=torch.int32, device=DEV)
    corr = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(_lib.load().gv_eval_metrics(ld.data_ptr(), lab.data_ptr(), n, c, pred.data_ptr(), conf.data_ptr(),
                                           corr.data_ptr(), torch.cuda.current_stream().cuda_stream), "eval")
    want_pred = logits.argmax(axis=1)
    assert pred.cpu().numpy().tolist() == want_pred.tolist() and want_pred[5] == 2
    want_conf = np.zeros((c, c), np.int32)
    for l, p in zip(labels, want_pred):
        if 0 <= l < c:
            want_conf[l, p] += 1
    assert conf.cpu().numpy().tolist() == want_conf.tolist()
    assert int(corr.item()) == int(((labels == want_pred) & (labels < c)).sum())
    # the harness: two batches through both protocols
    N, V, G = 2, 6, 10
    eng, P, Hd = make_engine("resnet_v2_50", N, V, 64, 64, c, G)
    accs = []
    for fused in (True, False):
        ev = Evaluator(eng)
        for seed in (1, 2):
            ev.add_batch(views(N, V, 64, 64, seed=seed).to(DEV), torch.tensor([seed, 3]), fused=fused)
        acc, cm, count = ev.result()
        assert count == 4 and cm.sum() == 4 and abs(acc - sum(ev.batch_accuracies) / 2) < 1e-12
        # a padded last batch (ViewBatcher(remainder="pad")): label -1 is ignored, the batch counts its real shapes
        a3 = ev.add_batch(views(N, V, 64, 64, seed=0).to(DEV), torch.tensor([0, -1]), fused=fused, valid=1)
        assert ev.num_shapes == 5 and int(ev.confusion.sum()) == 5 and a3 in (0.0, 1.0)
        accs.append((acc, cm.tolist()))
    assert accs[0] == accs[1]


def test_input_pipeline_preprocess_and_batcher(tmp_path):
    """gv_preprocess_views (legacy bilinear resize, flips, brightness, x/255 - 0.5) against the numpy
    oracle, and a GZIP TFRecord of PNG views through ViewBatcher into the engine's input layout."""
    import os
    from gvcnn_tf_amd import _lib, records as R
    from oracle import preprocess as OP
    rng = np.random.RandomState(2)
    nimg, h0, w0, H, W = 5, 37, 53, 64, 48
    src = rng.randint(0, 256, size=(nimg, h0, w0, 3)).astype(np.uint8)
    flip = np.array([0, 1, 2, 3, 0], np.int32)
    delta = rng.uniform(-1.1, 1.1, size=nimg).astype(np.float32)
    sd, fd, dd = torch.from_numpy(src).to(DEV), torch.from_numpy(flip).to(DEV), torch.from_numpy(delta).to(DEV)
    out = torch.empty(nimg, H, W, 3, device=DEV)
    st_ = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.load().gv_preprocess_views(sd.data_ptr(), nimg, h0, w0, H, W, fd.data_ptr(), dd.data_ptr(),
                                               out.data_ptr(), st_), "preprocess")
    np.testing.assert_allclose(out.cpu().numpy(), OP.preprocess_views(src, H, W, flip, delta), rtol=0, atol=2e-6)
    _lib.check(_lib.load().gv_preprocess_views(sd.data_ptr(), nimg, h0, w0, h0, w0, None, None,
                                               torch.empty(nimg, h0, w0, 3, device=DEV).data_ptr(), st_), "preprocess")
    # records -> batches
    N, V = 2, 3
    shapes = [([rng.randint(0, 256, size=(20, 24, 3)).astype(np.uint8) for _ in range(V)], 7 - i) for i in range(4)]
    path = os.path.join(tmp_path, "train.record")
    R.write_tfrecords(path, [R.make_example([R.encode_png(v) for v in vs], lab) for vs, lab in shapes])
    batches = list(R.ViewBatcher(path, V, 32, 32, N, DEV))
    assert len(batches) == 2
    for b, (x, labels) in enumerate(batches):
        assert tuple(x.shape) == (N, V, 32, 32, 3) and labels.tolist() == [shapes[2 * b][1], shapes[2 * b + 1][1]]
        want = OP.preprocess_views(np.stack([v for k in (2 * b, 2 * b + 1) for v in shapes[k][0]]), 32, 32)
        np.testing.assert_allclose(x.cpu().numpy().reshape(N * V, 32, 32, 3), want, rtol=0, atol=2e-6)
        assert float(x.min()) >= -0.5 and float(x.max()) <= 0.5 + 1e-6     # 255 * fp32(1/255) - 0.5 = 0.50000006
    # the trailing N_total % N shapes: dropped WITH a warning by default, padded on request (labels -1, last_valid)
    bt = R.ViewBatcher(path, V, 32, 32, 3, DEV)
    with pytest.warns(UserWarning, match="dropped"):
        assert len(list(bt)) == 1
    assert bt.dropped == 1
    bp = R.ViewBatcher(path, V, 32, 32, 3, DEV, remainder="pad")
    got = list(bp)
    assert len(got) == 2 and bp.last_valid == 1 and got[1][1].tolist() == [shapes[3][1], -1, -1]
    assert torch.equal(got[1][0][1], got[1][0][0]) and torch.equal(got[1][0][2], got[1][0][0])
    with pytest.raises(ValueError):
        list(R.ViewBatcher(path, V, 32, 32, 3, DEV, remainder="error"))


@pytest.mark.parametrize("lanes", [False, True])
def test_hipgraph_replay_matches_eager(lanes):
    """GVCNN.capture: the whole fused forward (plan with or without branch lanes, scorer, grouping, classifier) as one
    hipGraph; a replay on new input data written INTO the captured buffer equals the eager forward bit for bit."""
    N, V, C, G = 1, 6, 10, 10
    eng, P, Hd = make_engine("inception_v3", N, V, 75, 75, C, G, lanes=lanes)
    x = views(N, V, 75, 75, seed=4).to(DEV)
    replay = eng.capture(x)
    x2 = views(N, V, 75, 75, seed=9).to(DEV)
    s_e, S_e, l_e = [t.clone() for t in eng.forward(x2)]
    x.copy_(x2)
    s_g, S_g, l_g = replay()
    torch.cuda.synchronize()
    assert torch.equal(s_g, s_e) and torch.equal(S_g, S_e) and torch.equal(l_g, l_e)
    eng.check_status()


def test_hipgraph_replay_of_the_16bit_resnet_plan_with_bottleneck_launches():
    """The same for the 16-bit ResNet-v2-50 plan of round 6 — one launch per bottleneck unit in blocks 1 and 2
    (gv_bottleneck_unit_fwd: inline-asm LDS-DMA, a zero page, > 64 KB of dynamic LDS) — captured into a hipGraph."""
    N, V, C, G = 2, 3, 10, 10
    eng, P, Hd = make_engine("resnet_v2_50", N, V, 64, 64, C, G, storage="bf16")
    assert sum(1 for op in eng.plan.ops if op.get("chain")) == 5
    x = views(N, V, 64, 64, seed=4).to(DEV)
    replay = eng.capture(x)
    x2 = views(N, V, 64, 64, seed=9).to(DEV)
    s_e, S_e, l_e = [t.clone() for t in eng.forward(x2)]
    x.copy_(x2)
    s_g, S_g, l_g = replay()
    torch.cuda.synchronize()
    assert torch.equal(s_g, s_e) and torch.equal(S_g, S_e) and torch.equal(l_g, l_e)
    eng.check_status()


def test_bench_line_contract():
    """bench.py prints exactly ONE JSON line with every key the driver and the judge read (metric/value/unit/n_gpus/
    steps/warmup/ms_per_step/higher_is_better/scaling/vs_baseline/dtype/data/config + roofline + cpu_baseline)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--shapes", "2",
                          "--cpu-seconds", "2"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(j[k], t), (k, j[k])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["vs_baseline"] is None
    assert j["metric"] == "views/sec" and j["unit"] == "views/s" and j["scaling"] == "weak" and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 2 * 12 / (j["ms_per_step"] * 1e-3)) < 0.01 * j["value"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)


def test_roofline_timing_is_in_sequence_and_the_training_line_has_a_roofline():
    """gv_plan_time_each (what bench.py's `roofline` is built from): one duration per op of the plan, measured over whole
    passes in launch order; their sum is the single-lane step time.  And `bench.py --train --preset c3` carries a
    `roofline` object with the MFMA side (conv / wgrad) and the HBM side (bn / pool), every fraction = achieved / peak."""
    import json
    import os
    import subprocess
    import sys
    eng = gv.GVCNN("inception_v3", 2, 3, 139, 139, 10, 5, device=DEV, num_bins=5, lanes=False)
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    eng.plan.bind(P)
    x = (torch.rand(6, 139, 139, 3, generator=torch.Generator().manual_seed(0)) - 0.5).to(DEV)
    each = eng.plan.time_each(x, 3)
    assert len(each) == len(eng.plan.ops) and all(t > 0 for t in each)
    whole = eng.plan.time_range(x, 0, len(eng.plan.ops), 3)
    assert 0.5 * whole <= sum(each) <= 3.0 * whole + 1.0, (sum(each), whole)     # (event pairs add a little per op)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--train", "--preset", "c3", "--steps", "2",
                          "--warmup", "1", "--shapes", "2", "--no-traffic"], capture_output=True, text=True, timeout=900,
                         cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    r = j["roofline"]
    assert j["dtype"] == "bf16" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and "traffic" in r
    for key, bound in (("conv", "mfma"), ("wgrad", "mfma"), ("bn", "hbm"), ("pool", "hbm")):
        assert r[key]["bound"] == bound and abs(r[key]["frac"] - r[key]["achieved"] / r[key]["peak"]) < 1e-3, key
    assert r["bn"]["folded_sums"]["batchnorm_layers"] == 94
