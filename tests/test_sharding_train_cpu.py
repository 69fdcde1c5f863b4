"""View-sharded data-parallel TRAINING step on CPU: two gloo ranks run the algorithm of
gvcnn-tf_amd/sharding.py::ShardedTrainGVCNN — each rank the train-mode backbone of ITS views (per-view
BatchNorm statistics stay local), all-gather of scorer responses and final descriptors along the view
axis, the grouping head on the gathered data, the local slice of dF back through the local backbone,
bucketed all-reduce (sum) of the shared variables' gradients — with the oracle as the compute, and the
result must equal the single-process oracle step on the whole batch."""
import importlib.util
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import backbone as OB
from oracle import grouping as OG
from oracle import model as OM
from oracle import train as OT

N, V, WORLD, G, C, SIZE = 2, 4, 2, 10, 5, 48
BACKBONE = "resnet_v2_50"


def _setup():
    shapes = OB.trace_param_shapes(BACKBONE, SIZE, SIZE)
    P = OB.init_params(shapes, seed=2)
    raw_c, fin_c = 1024, 2048
    H = OM.init_head_params(V, raw_c, fin_c, C, seed=3)
    x = torch.rand(N, V, SIZE, SIZE, 3, generator=torch.Generator().manual_seed(0)) - 0.5
    labels = np.array([1, 3])
    return P, H, x, labels


def _load_sharding():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)                                       # host logic only: no HIP library needed
    return sh


def _worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    sh = _load_sharding()
    P, H, x, labels = _setup()
    raw_tap, final_tap = OM.TAPS[BACKBONE]
    lo, hi = sh.view_shard_range(V, WORLD, rank)
    Pg = {k: v.clone().requires_grad_(not k.endswith(("moving_mean", "moving_variance"))) for k, v in P.items()}
    finals, r_loc = [], []
    for v in range(lo, hi):                                           # this rank's views, train-mode BN per view
        ep = OM.run_backbone(BACKBONE, x[:, v], Pg, OB._BNMode(True, groups=[0] * N))
        nm = "dense" if v == 0 else "dense_%d" % v
        gap = ep[raw_tap].detach().mean(dim=(1, 2))
        r_loc.append(gap @ H[nm + "/kernel"].reshape(-1) + H[nm + "/bias"][0])
        finals.append(ep[final_tap])
    r_all = sh.gather_views(torch.stack(r_loc, dim=1))                # [N, V]
    F_all = sh.gather_views(torch.stack([f.detach() for f in finals], dim=1)).requires_grad_(True)
    scores = OG.score_from_r(r_all.numpy().astype(np.float32).mean(axis=0))
    scheme = OG.group_scheme([np.asarray(scores, np.float32)], G, V)
    weight = OG.group_weight(scheme)
    kn, bn = "dense_%d/kernel" % V, "dense_%d/bias" % V
    Wc, bc = H[kn].clone().requires_grad_(True), H[bn].clone().requires_grad_(True)
    loss, _, _ = OT.head(F_all.permute(1, 0, 2, 3, 4), scheme, weight, Wc, bc, labels)
    loss.backward()
    dF = F_all.grad                                                    # [N, V, h, w, C]
    torch.autograd.backward(finals, [dF[:, v] for v in range(lo, hi)])
    names = sorted(k for k, v in Pg.items() if v.requires_grad)
    grads = [Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k]) for k in names]
    nb = sh.allreduce_sum_bucketed(grads, bucket_bytes=8 << 20)
    assert nb > 1                                                      # several buckets were exercised
    # compare with the single-process step here (shipping 94 MB of gradients through the manager is slow)
    ref = OT.loss_and_grads(x, labels, P, H, G, BACKBONE)
    gmax = max(float(v.abs().max()) for v in ref["grads"].values())
    worst, worst_k, noise = 0.0, None, 0.0
    for k, g in zip(names, grads):
        r = ref["grads"][k] if k in ref["grads"] else torch.zeros_like(g)
        scale = float(r.abs().max())
        err = float((g - r).abs().max())
        if scale < 1e-5 * gmax:          # bias in front of a train-mode BatchNorm: zero gradient, rounding noise
            noise = max(noise, err / gmax)
        elif err / scale > worst:
            worst, worst_k = err / scale, k
    digest = float(sum(float(g.double().sum()) for g in grads))
    ret[rank] = dict(loss=float(loss.detach()), ref_loss=ref["loss"], scheme=scheme.tolist(),
                     ref_scheme=ref["scheme"].tolist(), worst=worst, worst_k=worst_k, noise=noise, digest=digest,
                     cls_err=max(float((Wc.grad - ref["grads"][kn]).abs().max()),
                                 float((bc.grad - ref["grads"][bn]).abs().max())),
                     cls_scale=float(ref["grads"][kn].abs().max()))
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharded_training_step_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(port, ret), nprocs=WORLD, join=True)
    for rank in range(WORLD):
        got = ret[rank]
        assert got["scheme"] == got["ref_scheme"]
        assert abs(got["loss"] - got["ref_loss"]) < 1e-5 * max(1.0, abs(got["ref_loss"]))
        assert got["cls_err"] < 1e-4 * got["cls_scale"]
        # same graph, fp32 CPU; only the cross-rank summation order differs
        assert got["worst"] < 2e-3, (got["worst"], got["worst_k"])
        assert got["noise"] < 1e-5
    assert ret[0]["digest"] == ret[1]["digest"]            # both ranks hold the same reduced gradients
