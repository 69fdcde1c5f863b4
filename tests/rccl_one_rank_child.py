"""Child of tests/test_gpu_rccl_one_rank.py — a FRESH process (started before it makes any GPU call) that is the only
rank of an RCCL process group and drives the N > 1 code of gvcnn-tf_amd/sharding.py through real RCCL collectives
(GV_FORCE_COLLECTIVES=1: a one-rank group still calls all_gather_into_tensor / all_reduce / the point-to-point form).
SURVEY §8(e); the reference's only collective call site is utils/_train_helper.py:17-30 (dead code there).
Prints ONE JSON line; every check is made here and reported, the parent asserts on the report."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
                      GV_FORCE_COLLECTIVES="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import gvcnn_tf_amd as gv
    from gvcnn_tf_amd import sharding
    from gvcnn_tf_amd.sharding import ShardedGVCNN, ShardedTrainGVCNN
    from gvcnn_tf_amd.training import TrainGVCNN
    rep = {"backend": dist.get_backend(), "world": dist.get_world_size(),
           "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "forced": sharding._FORCE}

    # a bare collective of each kind first: the communicator exists and moves bytes
    t = torch.arange(1024, dtype=torch.float32, device=dev)
    u = torch.empty_like(t)
    dist.all_gather_into_tensor(u, t)
    v = t.clone()
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    rep["bare_all_gather_ok"] = bool(torch.equal(u, t))
    rep["bare_all_reduce_ok"] = bool(torch.equal(v, t))

    # ---- inference: ShardedGVCNN against the unsharded engine, bit for bit ----
    N, V, H, W, C, G = 2, 3, 64, 64, 10, 10
    eng = gv.GVCNN("resnet_v2_50", N, V, H, W, C, G, device=dev)
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C, seed=3, spread_scores=True)
    eng.plan.bind(P)
    eng.set_head(Hd)
    x = (torch.rand(N, V, H, W, 3, generator=torch.Generator().manual_seed(0)) - 0.5).to(dev)
    ref = [r.clone() for r in eng.forward(x)]
    ref_scheme = eng.scheme.clone()

    def same(got):
        return bool(all(torch.equal(a, b) for a, b in zip(got, ref)) and torch.equal(eng.scheme, ref_scheme))
    for mode in ("collective", "direct"):
        sh = ShardedGVCNN(eng, exchange="allgather", gather_mode=mode)
        assert sh.exchanging
        rep["infer_%s_bitwise" % mode] = same([r.clone() for r in sh.forward(x)])
    sh = ShardedGVCNN(eng, exchange="allgather", overlap=True, gather_mode="collective")
    assert sh.overlap
    first = sh.forward(x)
    second = [r.clone() for r in sh.forward(x)]                  # the results of the first call, one call later
    last = [r.clone() for r in sh.flush()]
    rep["infer_overlap_bitwise"] = first is None and same(second) and same(last)
    sh = ShardedGVCNN(eng, exchange="scores")
    rep["infer_scores_exchange_bitwise"] = same([r.clone() for r in sh.forward(x)])

    # ---- training: ShardedTrainGVCNN one step against the engine's own step ----
    def plain_step(e, xx, ll):
        e.forward(xx, ll, check=False)
        e.backward()
        e.update_moving_averages()
        e.apply_momentum(1e-3, 0.9, 1e-4)
        return e.loss
    labels = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(1))
    for mode, storage in (("shapes", "bf16"), ("views", "f32")):
        kw = dict(device=dev, num_bins=G, storage=storage)
        a = TrainGVCNN("resnet_v2_50", N, V, H, W, C, G, **kw)
        b = TrainGVCNN("resnet_v2_50", N, V, H, W, C, G, **kw)
        la = float(plain_step(a, x, labels))
        shb = ShardedTrainGVCNN(b, mode=mode)
        assert not shb.solo
        lb = float(shb.train_step(x, labels, lr=1e-3, mu=0.9, weight_decay=1e-4))
        torch.cuda.synchronize()
        gmax = max(float(g.abs().max()) for g in a.grads.values())
        worst = max(float((a.grads[k].float() - b.grads[k].float()).abs().max()) for k in a.grads) / gmax
        pw = max(float((a.params[k].float() - b.params[k].float()).abs().max()) for k in a.params)
        rep["train_%s" % mode] = {"loss_plain": la, "loss_sharded": lb, "loss_bitwise": la == lb,
                                  "grads_bitwise": bool(all(torch.equal(a.grads[k], b.grads[k]) for k in a.grads)),
                                  "grad_worst_rel": worst, "param_worst_abs": pw,
                                  "scheme_equal": bool(torch.equal(a.scheme, b.scheme))}
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL1 " + json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
