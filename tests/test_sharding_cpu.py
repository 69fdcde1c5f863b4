"""N>1 path on CPU: two gloo ranks exchange scorer responses and view descriptors exactly as
gvcnn-tf_amd/sharding.py does on RCCL, and the grouping computed from the gathered data (oracle as
checker) equals the single-process result on the whole batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import grouping as OG

V, N_L, WORLD, G = 6, 2, 2, 10
SHAPE = (3, 3, 16)


def _data():
    rng = np.random.RandomState(0)
    F = rng.randn(WORLD * N_L, V, *SHAPE).astype(np.float32)        # [N_g, V, h, w, C]
    r = rng.uniform(0.05, 8.0, size=(WORLD * N_L, V)).astype(np.float32)
    return F, r


def _scores(r_all):                                                  # model.py:146-147, n ascending
    s = np.zeros(V, np.float32)
    for n in range(r_all.shape[0]):
        s = (s + r_all[n]).astype(np.float32)
    return OG.score_from_r(s / np.float32(r_all.shape[0]))


def _worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)                                       # host logic only: no HIP library needed
    F, r = _data()
    lo, hi = sh.shard_range(WORLD * N_L, WORLD, rank)
    r_all = sh.gather_scores(torch.from_numpy(r[lo:hi].reshape(-1).copy()))
    F_all = sh.gather_descriptors(torch.from_numpy(F[lo:hi].copy()))
    ret[rank] = (r_all.numpy().copy(), F_all.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(port, ret), nprocs=WORLD, join=True)
    F, r = _data()
    ref_scores = _scores(r)
    scheme = OG.group_scheme([ref_scores], G, V)
    weight = OG.group_weight(scheme)
    S_ref = OG.group_fusion(OG.view_pooling([F[:, v] for v in range(V)], scheme), weight)
    for rank in range(WORLD):
        r_all, F_all = ret[rank]
        np.testing.assert_array_equal(r_all.reshape(WORLD * N_L, V), r)          # global shape-major order
        np.testing.assert_array_equal(F_all, F)
        sc = _scores(r_all.reshape(WORLD * N_L, V))
        np.testing.assert_array_equal(sc, ref_scores)                            # identical on every rank
        sch = OG.group_scheme([sc], G, V)
        assert sch.tolist() == scheme.tolist()
        S = OG.group_fusion(OG.view_pooling([F_all[:, v] for v in range(V)], sch), OG.group_weight(sch))
        np.testing.assert_array_equal(S, S_ref)
        # exchange="scores": pooling only the owned shapes gives the owned rows of the global result
        lo, hi = rank * N_L, (rank + 1) * N_L
        S_loc = OG.group_fusion(OG.view_pooling([F[lo:hi, v] for v in range(V)], sch), weight)
        np.testing.assert_array_equal(S_loc, S_ref[lo:hi])


def test_shard_range():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding2", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    assert [sh.shard_range(32, 8, p) for p in (0, 7)] == [(0, 4), (28, 32)]
    import pytest
    with pytest.raises(ValueError):
        sh.shard_range(30, 8, 0)


def _uneven_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=3)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    full = torch.arange(2 * 7 * 3, dtype=torch.float32).reshape(2, 7, 3)          # [N, V = 7, E]
    lo, hi = sh.view_shard_range(7, 3, rank)
    ret[rank] = ((lo, hi), sh.gather_views(full[:, lo:hi].contiguous(), num_views=7).numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_view_shards_gather_in_view_order():
    """7 views over 3 ranks (3, 2, 2): padded all-gather, trimmed, global view order on every rank."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_uneven_worker, args=(port, ret), nprocs=3, join=True)
    full = np.arange(2 * 7 * 3, dtype=np.float32).reshape(2, 7, 3)
    assert [ret[r][0] for r in range(3)] == [(0, 3), (3, 5), (5, 7)]
    for r in range(3):
        assert np.array_equal(ret[r][1], full)


def _allreduce_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    t = torch.arange(10, dtype=torch.float64) * (rank + 1)          # the BatchNorm sums are fp64
    sh.allreduce_sum_(t)
    ret[rank] = t.numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


def test_bn_sum_allreduce_helper():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_allreduce_worker, args=(port, ret), nprocs=2, join=True)
    for r in range(2):
        assert np.array_equal(ret[r], np.arange(10, dtype=np.float64) * 3)
