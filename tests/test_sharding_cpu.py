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


# ------------------------------------------------------------------------------------------------
# hybrid (view groups x shape shards) layout of the training step: grid, sub-groups, exchanges (world size 4 = 2 x 2)
# ------------------------------------------------------------------------------------------------
def test_hybrid_grid_gives_every_rank_the_same_work():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding_h", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    assert sh.hybrid_grid(12, 8) == (4, 2)          # 3 views of half the shapes each: 1/8 of the work per rank
    assert sh.hybrid_grid(12, 4) == (4, 1) and sh.hybrid_grid(12, 2) == (2, 1) and sh.hybrid_grid(12, 1) == (1, 1)
    assert sh.hybrid_grid(20, 8) == (4, 2) and sh.hybrid_grid(6, 4) == (2, 2) and sh.hybrid_grid(12, 5) == (1, 5)
    for V, P in ((12, 8), (20, 8), (6, 4), (12, 6)):
        vg, s = sh.hybrid_grid(V, P)
        assert vg * s == P and V % vg == 0
        cells = {sh.hybrid_coords(V, P, r) for r in range(P)}
        assert cells == {(g, k) for g in range(vg) for k in range(s)}       # a bijection rank <-> (view group, shape shard)
    # view sharding alone on 8 ranks: the most loaded rank owns 2 of 12 views -> efficiency 12 / (8 * 2) = 0.75
    assert max(hi - lo for lo, hi in (sh.view_shard_range(12, 8, r) for r in range(8))) == 2


def _hybrid_worker(rank, port, ret):
    import importlib.util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=4)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding_h", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    V, N, W = 6, 4, 4
    vg, s = sh.hybrid_grid(V, W)                      # (2, 2)
    gi, si = sh.hybrid_coords(V, W, rank)
    shape_group, view_group = sh.hybrid_groups(V, W, rank)
    v_l, n_l = V // vg, N // s
    full = torch.arange(N * V * 3, dtype=torch.float32).view(N, V, 3)         # a [N, V, feature] tensor everybody knows
    mine = full[si * n_l:(si + 1) * n_l, gi * v_l:(gi + 1) * v_l].contiguous()
    views = sh.gather_views(mine, view_group, V)                               # [N_l, V, 3]: all views of my shapes
    allsh = sh._all_gather_flat(views, shape_group)                           # [N, V, 3]: every shape, shape order
    # BatchNorm-style sum inside the shape group (same views, all shapes) and gradient-style sums
    bn = sh.allreduce_sum_(mine.sum(dim=0).clone(), shape_group)              # [v_l, 3] over all N shapes
    world_sum = mine.sum().reshape(1).clone()
    sh.allreduce_sum_bucketed([world_sum], group=None)
    ret[rank] = dict(views_ok=bool(torch.equal(views, full[si * n_l:(si + 1) * n_l])), all_ok=bool(torch.equal(allsh, full)),
                     bn_ok=bool(torch.equal(bn, full[:, gi * v_l:(gi + 1) * v_l].sum(dim=0))),
                     world_ok=float(world_sum) == float(full.sum()), coords=(gi, si))
    dist.barrier()
    dist.destroy_process_group()


def test_hybrid_exchanges_on_four_gloo_ranks():
    """2 view groups x 2 shape shards: descriptors gathered inside the view group give every rank all views of ITS
    shapes, the scorer responses gathered over both axes give everybody the whole [N, V] array in shape order, a sum
    inside the shape group covers all N shapes of the rank's views (BatchNorm statistics), a world sum covers everything
    (shared variables' gradients)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hybrid_worker, args=(port, ret), nprocs=4, join=True)
    assert {tuple(ret[r]["coords"]) for r in range(4)} == {(0, 0), (0, 1), (1, 0), (1, 1)}
    for r in range(4):
        assert ret[r]["views_ok"] and ret[r]["all_ok"] and ret[r]["bn_ok"] and ret[r]["world_ok"], (r, dict(ret[r]))


# ------------------------------------------------------------------------------------------------
# ShardedGVCNN(overlap=True): the descriptor all-gather of step k rides under the backbone of step k+1
# ------------------------------------------------------------------------------------------------
class _FakeEngine:
    """The engine surface ShardedGVCNN drives, with the CPU oracle's grouping functions as compute: per step the
    'backbone' produces descriptors and scorer responses that depend on the step's input."""
    per_shape = False

    def __init__(self, world, rank):
        self.N, self.V, self.G = N_L, V, G
        self.world, self.rank = world, rank
        self.r_img = torch.zeros(N_L * V)
        self.scores = torch.zeros(V)
        self.scheme = torch.zeros(G, V, dtype=torch.int32)
        self.weight = torch.zeros(G)
        self._F = None

    def run_backbone(self, x):                         # x [N_l, V, *SHAPE]: "descriptors" = 2x + 1, responses = mean |x|
        self._F = x * 2.0 + 1.0
        self._r = x.abs().mean(dim=(2, 3, 4)) * 4.0 + 0.05

    def compute_scores(self):
        self.r_img = self._r.reshape(-1).clone()

    def finalize_scores(self, r_all, num_shapes):
        self.scores = torch.from_numpy(_scores(r_all.numpy().reshape(num_shapes, V)))

    def assign_groups(self, check=False):
        sch = OG.group_scheme([self.scores.numpy()], G, V)
        self.scheme = torch.from_numpy(sch.astype(np.int32))
        self.weight = torch.from_numpy(OG.group_weight(sch))

    def final_view_descriptors(self):
        return self._F

    def pool_fuse_classify(self, scheme, weight, F=None):
        F = self._F if F is None else F
        S = OG.group_fusion(OG.view_pooling([F.numpy()[:, v] for v in range(V)], scheme.numpy()), weight.numpy())
        return torch.from_numpy(S), torch.from_numpy(S.mean(axis=(1, 2)))

    def check_status(self):
        pass


def _overlap_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding_o", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    rng = np.random.RandomState(100 + rank)
    xs = [torch.from_numpy(rng.randn(N_L, V, *SHAPE).astype(np.float32)) for _ in range(4)]
    plain = sh.ShardedGVCNN(_FakeEngine(WORLD, rank), exchange="allgather")
    want = [tuple(t.clone() for t in plain.forward(x, check=False)) for x in xs]
    piped = sh.ShardedGVCNN(_FakeEngine(WORLD, rank), exchange="allgather", overlap=True)
    assert piped.overlap
    got = []
    for x in xs:
        out = piped.forward(x, check=False)            # results of the PREVIOUS call
        if out is not None:
            got.append(tuple(t.clone() for t in out))
    got.append(tuple(t.clone() for t in piped.flush()))
    assert piped.flush() is None
    ok = len(got) == len(want) and all(all(torch.equal(a, b) for a, b in zip(g, w)) for g, w in zip(got, want))
    ret[rank] = (ok, [float(w[2].sum()) for w in want])
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_returns_the_same_results_one_call_later():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_overlap_worker, args=(port, ret), nprocs=WORLD, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == ret[1][1]                      # allgather form: every rank holds all logits, identical


# ------------------------------------------------------------------------------------------------
# OverlappedFlatAllReduce: the filter gradients are reduced slice by slice while they become final
# ------------------------------------------------------------------------------------------------
def _flat_reduce_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding_f", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    n, hi = 100003, 90001                                   # the tail [hi, n) is NOT part of the reduced region
    base = torch.arange(n, dtype=torch.float32) * (rank + 1)
    flat = base.clone()
    red = sh.OverlappedFlatAllReduce(flat, hi, bucket_bytes=4 * 7000)
    # the backward pass finalises the region from its end: decreasing, irregular frontiers (and a repeated one)
    for lo in (88000, 87990, 70000, 69000, 69000, 41000, 40000, 12345, 3):
        flat[lo:hi] += 0.0                                   # (stands for "these values are final now")
        red.progress(lo)
        assert red.sent_lo >= lo
    launches = red.finish()
    want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(WORLD))
    ok = torch.equal(flat[:hi], want[:hi]) and torch.equal(flat[hi:], base[hi:]) and red.sent_lo == 0
    ret[rank] = (ok, launches)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_flat_all_reduce_sums_every_element_exactly_once():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_flat_reduce_worker, args=(port, ret), nprocs=WORLD, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == ret[1][1] and 4 <= ret[0][1] <= 9    # several slices, fewer than the progress calls


# ------------------------------------------------------------------------------------------------
# the node's real world size: 8 ranks — hybrid 4 x 2 grid (12 views), ShardedGVCNN with the direct (point-to-point)
# gather beside the collective one, uneven view shards (12 views on 8 ranks: 2,2,2,2,1,1,1,1)
# ------------------------------------------------------------------------------------------------
def _load_sharding(tag):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(tag, os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    return sh


def _world8_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=8)
    torch.set_num_threads(1)
    sh = _load_sharding("gv_sharding_w8")
    out = {}
    # (1) hybrid grid 4 view groups x 2 shape shards at V = 12: the layout bench.py --train uses on a whole node
    V12, N, W = 12, 4, 8
    vg, s = sh.hybrid_grid(V12, W)
    gi, si = sh.hybrid_coords(V12, W, rank)
    shape_group, view_group = sh.hybrid_groups(V12, W, rank)
    v_l, n_l = V12 // vg, N // s
    full = torch.arange(N * V12 * 3, dtype=torch.float32).view(N, V12, 3)
    mine = full[si * n_l:(si + 1) * n_l, gi * v_l:(gi + 1) * v_l].contiguous()
    for mode in ("collective", "direct"):
        sh.set_gather_mode(mode)
        views = sh.gather_views(mine, view_group, V12)                          # all 12 views of my 2 shapes
        allsh = sh._all_gather_flat(views, shape_group)                         # every shape, shape order
        out["hybrid_" + mode] = bool(torch.equal(views, full[si * n_l:(si + 1) * n_l])) and bool(torch.equal(allsh, full))
    sh.set_gather_mode("collective")
    bn = sh.allreduce_sum_(mine.double().sum(dim=0).clone(), shape_group)       # statistics: the 2 ranks that share my views
    out["bn"] = bool(torch.equal(bn, full[:, gi * v_l:(gi + 1) * v_l].double().sum(dim=0)))
    out["grid"] = (vg, s, gi, si)
    # (2) uneven view shards, both gather forms: 12 views on 8 ranks
    lo, hi = sh.view_shard_range(V12, W, rank)
    for mode in ("collective", "direct"):
        sh.set_gather_mode(mode)
        got = sh.gather_views(full[:, lo:hi].contiguous(), num_views=V12)
        out["uneven_" + mode] = bool(torch.equal(got, full))
    out["views"] = (lo, hi)
    # (3) ShardedGVCNN at world 8, exchange = allgather, in both gather forms and overlapped: identical results
    rng = np.random.RandomState(200 + rank)
    xs = [torch.from_numpy(rng.randn(N_L, V, *SHAPE).astype(np.float32)) for _ in range(3)]
    res = {}
    for name, kw in (("collective", dict(gather_mode="collective")), ("direct", dict(gather_mode="direct")),
                     ("direct_overlap", dict(gather_mode="direct", overlap=True))):
        eng = sh.ShardedGVCNN(_FakeEngine(8, rank), exchange="allgather", **kw)
        got = []
        for x in xs:
            o = eng.forward(x, check=False)
            if o is not None:
                got.append(tuple(t.clone() for t in o))
        if eng.overlap:
            got.append(tuple(t.clone() for t in eng.flush()))
        res[name] = got
    same = all(len(res[k]) == 3 and all(all(torch.equal(a, b) for a, b in zip(g, w))
                                        for g, w in zip(res[k], res["collective"])) for k in res)
    out["sharded_same"] = same
    out["logits"] = [float(g[2].sum()) for g in res["collective"]]
    out["num_logits"] = int(res["collective"][0][2].numel())
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_eight_gloo_ranks_hybrid_grid_direct_gather_and_sharded_forward():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_world8_worker, args=(port, ret), nprocs=8, join=True)
    cells = set()
    for r in range(8):
        o = ret[r]
        assert o["hybrid_collective"] and o["hybrid_direct"] and o["bn"], (r, o)
        assert o["uneven_collective"] and o["uneven_direct"], (r, o)
        assert o["sharded_same"], r
        assert o["num_logits"] == 8 * N_L * SHAPE[2]        # the all-gather form: every rank holds all N_g shapes' rows
        assert o["logits"] == ret[0]["logits"]              # ... identical everywhere
        vg, s_, gi, si = o["grid"]
        assert (vg, s_) == (4, 2)
        cells.add((gi, si))
    assert cells == {(g, k) for g in range(4) for k in range(2)}
    assert [ret[r]["views"] for r in range(8)] == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 9), (9, 10), (10, 11), (11, 12)]


def test_moving_average_count_is_the_global_sample_count():
    """Hybrid / shape-sharded training: a view's batch statistics are reduced over the ranks that share it, so the sample
    count of the moving-variance update (n / (n - 1)) is local shapes x shape shards x pixels (advisor finding of round 2:
    the local count was used)."""
    sh = _load_sharding("gv_sharding_mv")
    assert sh.moving_average_count(16, 1, 49) == 16 * 49
    assert sh.moving_average_count(16, 2, 49) == 32 * 49              # 8 ranks = 4 view groups x 2 shape shards
    vg, s = sh.hybrid_grid(12, 8)
    n_global = 32
    assert sh.moving_average_count(n_global // s, s, 144) == n_global * 144    # what the unsharded engine counts


def _forced_one_rank_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gv_sharding", os.path.join(root, "gvcnn-tf_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    g = torch.Generator().manual_seed(0)
    r = torch.rand(12, generator=g)
    F = torch.randn(2, 6, 3, 3, 8, generator=g)
    out = {}
    # not forced: a one-rank group exchanges nothing (the very same tensors come back)
    out["solo_identity"] = sh.gather_scores(r) is r and sh.gather_descriptors(F) is F and sh.allreduce_sum_(r) is r
    sh.set_force_collectives(True)
    r2, F2, Fv = sh.gather_scores(r), sh.gather_descriptors(F), sh.gather_views(F, None, 6)
    out["forced_new_tensors"] = r2 is not r and F2 is not F
    out["forced_same_values"] = bool(torch.equal(r2, r) and torch.equal(F2, F) and torch.equal(Fv, F))
    d = torch.empty_like(F)
    sh.direct_all_gather(d, F)                                   # no peers: the local copy only
    out["direct"] = bool(torch.equal(d, F))
    gs = [torch.randn(5, generator=g), torch.randn(3, 4, generator=g)]
    keep = [t.clone() for t in gs]
    out["buckets"] = sh.allreduce_sum_bucketed(gs, bucket_bytes=16)
    out["bucket_values"] = bool(all(torch.equal(a, b) for a, b in zip(gs, keep)))
    flat = torch.randn(1000, generator=g)
    want = flat.clone()
    red = sh.OverlappedFlatAllReduce(flat, 800, bucket_bytes=4 * 256)
    for lo in (700, 300, 0):
        red.progress(lo)
    out["overlap_launches"] = red.finish()
    out["overlap_values"] = bool(torch.equal(flat, want))
    sh.set_force_collectives(False)
    ret[0] = out
    dist.destroy_process_group()


def test_forced_collectives_on_a_one_rank_group():
    """GV_FORCE_COLLECTIVES / set_force_collectives (what the one-rank RCCL test of the GPU box relies on): a one-rank group
    exchanges nothing by default; forced, every helper goes through its collective and returns the same values."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_forced_one_rank_worker, args=(port, ret), nprocs=1, join=True)
    out = ret[0]
    assert out["solo_identity"] and out["forced_new_tensors"] and out["forced_same_values"] and out["direct"]
    assert out["buckets"] >= 2 and out["bucket_values"]
    assert out["overlap_launches"] >= 2 and out["overlap_values"]
