"""One process per GPU: shard the view batch, exchange what the grouping module couples.

What couples the shards: nothing in the backbone (weights are shared, every view is
independent — nets/model.py:129-141); the grouping module couples
  (1) the N shapes of a batch through the batch-mean score of each view (model.py:146), and
  (2) the V views of one shape through view pooling (model.py:62-74).

Partitioning: the global batch of N_g shapes is cut on SHAPE boundaries, rank p owning shapes
[p*N_l, (p+1)*N_l) with all their V views (a contiguous slice of the flattened [N_g*V] view batch in
memory order).  Exchanges, both over RCCL (torch.distributed backend "nccl") on xGMI:

  * scores:      all-gather of the per-image scorer responses r_img (N_l*V floats per rank), then
                 every rank reduces the SAME array in the SAME order -> bitwise identical scores,
                 hence identical group indices on every rank (an all-reduce would leave the
                 summation order to the library);
  * descriptors: `exchange="allgather"` (BASELINE.json north_star): all-gather of the final view
                 descriptors [N_l*V, h, w, C] so every rank holds the whole [N_g, V, h, w, C] and
                 produces all N_g logits; `exchange="scores"`: each rank pools only the shapes it
                 owns — no descriptor traffic at all, logits stay sharded (what a data-parallel
                 trainer wants).

xGMI is a point-to-point mesh: one all-gather of a large contiguous buffer per step (not one per
view or per layer) keeps every link busy with a single big message.
"""
import os

import torch
import torch.distributed as dist

# A ONE-rank process group normally exchanges nothing (every `world == 1` test below returns the local data).  With
# GV_FORCE_COLLECTIVES=1 (or set_force_collectives(True)) a one-rank group still goes through every collective call —
# all_gather_into_tensor, all_reduce, the point-to-point form — so that a single-GPU box can execute the exact RCCL code
# path of the N > 1 job (tests/test_gpu_rccl_one_rank.py); the values are those of the local data by construction.
_FORCE = os.environ.get("GV_FORCE_COLLECTIVES", "0") == "1"


def set_force_collectives(on):
    global _FORCE
    _FORCE = bool(on)


def _solo(world):
    """True when there is nobody to exchange with AND the collectives are not forced."""
    return world == 1 and not _FORCE


def shard_range(num_shapes_global, world_size, rank):
    """Shapes [lo, hi) owned by `rank`; the batch must divide evenly (weak scaling: fixed N_l)."""
    if num_shapes_global % world_size != 0:
        raise ValueError("global batch %d does not divide over %d ranks" % (num_shapes_global, world_size))
    n_l = num_shapes_global // world_size
    return rank * n_l, (rank + 1) * n_l


# How a gather travels ("direct one-shot all-gather on the full mesh ... ring would be 7x that"):
#   "collective"  dist.all_gather_into_tensor — the library picks the algorithm (RCCL: ring / tree by size);
#   "direct"      every rank posts ONE send of its shard to EACH peer and one receive from each (batch_isend_irecv):
#                 on the xGMI mesh every pair of GPUs has its own link, so the P-1 transfers of a rank leave on P-1
#                 different links at once and the gather takes shard / link-rate instead of (P-1) ring steps.
# A process-wide default (set_gather_mode) that ShardedGVCNN / ShardedTrainGVCNN can override per instance; the RCCL
# collective stays the default until a node has shown the direct form to be faster.  Same bytes, same result.
_GATHER_MODE = "collective"


def set_gather_mode(mode):
    global _GATHER_MODE
    if mode not in ("collective", "direct"):
        raise ValueError(mode)
    _GATHER_MODE = mode


def direct_all_gather(out, src, group=None, async_op=False):
    """out [world * n, ...] <- every rank's src [n, ...], as point-to-point transfers to and from every peer.
    Returns the list of work handles when async_op (wait on each), else waits itself."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = src.shape[0]
    out[rank * n:(rank + 1) * n].copy_(src)
    ops = []
    for d in range(1, world):                               # peer order rotated by rank: no two ranks start on one peer
        peer = (rank + d) % world
        gpeer = dist.get_global_rank(group, peer) if group is not None else peer
        ops.append(dist.P2POp(dist.isend, src, gpeer, group))
        frm = (rank - d) % world
        gfrm = dist.get_global_rank(group, frm) if group is not None else frm
        ops.append(dist.P2POp(dist.irecv, out[frm * n:(frm + 1) * n], gfrm, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return works
    for w in works:
        w.wait()
    return None


def _all_gather_flat(t_local, group=None, mode=None):
    """All-gather along dim 0 (mode: "collective" | "direct", default the process-wide one).  RCCL takes device tensors
    directly; a gloo group (CPU tests, and the single-device control-flow check `bench.py --backend gloo`) is fed
    through host memory."""
    world = dist.get_world_size(group)
    t_local = t_local.contiguous()
    via_host = t_local.is_cuda and dist.get_backend(group) == "gloo"
    src = t_local.cpu() if via_host else t_local
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if (mode or _GATHER_MODE) == "direct":
        direct_all_gather(out, src, group)
    else:
        dist.all_gather_into_tensor(out, src, group=group)
    return out.to(t_local.device) if via_host else out


def gather_scores(r_img_local, group=None, mode=None):
    """All-gather the local scorer responses [N_l*V] -> [P*N_l*V] (global shape-major order)."""
    world = dist.get_world_size(group)
    if _solo(world):
        return r_img_local
    return _all_gather_flat(r_img_local.reshape(-1), group, mode)


def gather_descriptors(F_local, group=None, mode=None):
    """All-gather final view descriptors [N_l, V, h, w, C] -> [P*N_l, V, h, w, C]."""
    world = dist.get_world_size(group)
    if _solo(world):
        return F_local
    return _all_gather_flat(F_local, group, mode)


class ShardedGVCNN:
    """Wraps a per-rank GVCNN engine built for the LOCAL batch (N_l shapes).

    overlap=True (exchange='allgather' only): the descriptor all-gather of step k is issued asynchronously on RCCL's own
    stream and its grouping head (view pooling, fusion, classifier over all N_g shapes) runs one call LATER, after the
    backbone launches of step k+1 have been enqueued — the exchange (78.6 MB sent / 550 MB received per rank and step at
    configs[1] on 8 GPUs: about 2 ms of ring-bound xGMI against a 15 ms step) then rides under matrix work instead of
    standing between two steps.  forward() returns the results of the PREVIOUS call (None the first time); flush()
    returns the last one.  Values are exactly those of the non-overlapped form: the same collectives on the same data."""

    def __init__(self, engine, group=None, exchange="allgather", overlap=False, gather_mode=None):
        if exchange not in ("allgather", "scores"):
            raise ValueError(exchange)
        if gather_mode not in (None, "collective", "direct"):
            raise ValueError(gather_mode)
        self.eng = engine
        self.group = group
        self.exchange = exchange
        self.gather_mode = gather_mode        # None: the process-wide default (set_gather_mode)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.exchanging = dist.is_initialized() and not _solo(self.world)   # (a forced one-rank group exchanges too)
        self.overlap = bool(overlap) and exchange == "allgather" and self.exchanging
        self._pending = None                  # (work handle | None, F_all, scheme, weight, scores) of the previous step
        self._slot = 0
        self._stage = [None, None]            # double-buffered copies of the local final descriptors / gathered tensors

    def _issue_gather(self):
        """Copy this step's descriptors / grouping state aside and start their all-gather; returns the pending record."""
        eng = self.eng
        F_loc = eng.final_view_descriptors()
        k = self._slot
        self._slot ^= 1
        if self._stage[k] is None:
            self._stage[k] = (torch.empty_like(F_loc),
                              torch.empty((self.world * F_loc.shape[0],) + tuple(F_loc.shape[1:]), dtype=F_loc.dtype,
                                          device=F_loc.device))
        src, dst = self._stage[k]
        src.copy_(F_loc)                                        # the plan's tap buffer is overwritten by the next step
        if src.is_cuda and dist.get_backend(self.group) == "gloo":     # control-flow check on one device: through the host
            h = torch.empty(dst.shape, dtype=dst.dtype)
            dist.all_gather_into_tensor(h, src.cpu(), group=self.group)
            dst.copy_(h)
            work = None
        elif (self.gather_mode or _GATHER_MODE) == "direct":
            work = direct_all_gather(dst, src, self.group, async_op=True)
        else:
            work = dist.all_gather_into_tensor(dst, src, group=self.group, async_op=True)
        return (work, dst, eng.scheme.clone(), eng.weight.clone(), eng.scores.clone())

    def _finish(self, pending):
        work, F_all, scheme, weight, scores = pending
        for w in (work if isinstance(work, (list, tuple)) else ([work] if work is not None else [])):
            w.wait()                                            # the CURRENT stream waits for the transfer; no host sync
        S, logits = self.eng.pool_fuse_classify(scheme, weight, F=F_all)
        return scores, S, logits

    def flush(self):
        """Results of the last overlapped step (None when nothing is pending)."""
        if self._pending is None:
            return None
        out = self._finish(self._pending)
        self._pending = None
        return out

    def forward(self, views_local, check=True):
        """views_local [N_l, V, H, W, 3].  Returns (scores [V], shape_descriptor, logits): for
        exchange='allgather' over all N_g = P*N_l shapes (identical on every rank), for
        exchange='scores' over the local N_l shapes."""
        eng = self.eng
        if getattr(eng, "per_shape", False):                   # per-shape grouping couples nothing: no exchange
            return eng.forward_per_shape(views_local, check)
        eng.run_backbone(views_local)
        eng.compute_scores()                                   # fills eng.r_img (local) + local scores
        if self.exchanging:
            r_all = gather_scores(eng.r_img, self.group, self.gather_mode)
            eng.finalize_scores(r_all, eng.N * self.world)     # same array, same order on every rank
        eng.assign_groups(check=False)
        if self.overlap:
            now = self._issue_gather()                          # collective of THIS step: in flight under the next backbone
            prev, self._pending = self._pending, now
            if check:
                eng.check_status()
            return self._finish(prev) if prev is not None else None
        if self.exchange == "allgather" and self.exchanging:
            F_all = gather_descriptors(eng.final_view_descriptors(), self.group, self.gather_mode)
            S, logits = eng.pool_fuse_classify(eng.scheme, eng.weight, F=F_all)
        else:
            S, logits = eng.pool_fuse_classify(eng.scheme, eng.weight)
        if check:
            eng.check_status()
        return eng.scores, S, logits


# ------------------------------------------------------------------------------------------------
# training: view-sharded data parallelism
# ------------------------------------------------------------------------------------------------
# In train mode BatchNorm normalises each view's graph copy over the N images of THAT view
# (nets/model.py:129-141), so the batch is cut on VIEW boundaries: rank p runs the backbone for views
# [p*V_l, (p+1)*V_l) of every shape and its BN statistics are exactly the reference's — no per-layer
# statistics exchange.  Exchanges per step:
#   forward   all-gather of the scorer responses [N, V_l] and of the final descriptors [N, V_l, h, w, C]
#             (one large message each); every rank then runs the (tiny) grouping head on all V views;
#   backward  each rank keeps the slice of dF that belongs to its views (no traffic), runs its backbone
#             backward, and the variable gradients — shared by the V views, hence SUMMED over them
#             (utils/train_utils.py:217-259) — are all-reduced in a few large buckets: xGMI is a
#             point-to-point mesh, so few large messages beat one message per variable.
def view_shard_range(num_views, world_size, rank):
    """Views [lo, hi) owned by `rank`: as even as possible (the first num_views % world_size ranks own one more), so
    any world size up to num_views works — 12 views on 8 GPUs run as 2,2,2,2,1,1,1,1."""
    if world_size > num_views:
        raise ValueError("%d ranks for %d views: a rank would own no view" % (world_size, num_views))
    base, extra = divmod(num_views, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_views(t_local, group=None, num_views=None):
    """All-gather along the VIEW axis: [N, V_l, ...] on every rank -> [N, V, ...] (global view order).  Ranks may own
    different numbers of views (view_shard_range): pass the global `num_views`; shards are padded to the largest one
    for the collective and trimmed afterwards."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not dist.is_initialized() or _solo(world):
        return t_local
    t_local = t_local.contiguous()
    n, v_l = t_local.shape[0], t_local.shape[1]
    if num_views is None:
        num_views = v_l * world
    counts = [view_shard_range(num_views, world, r) for r in range(world)]
    v_max = max(hi - lo for lo, hi in counts)
    if v_l < v_max:
        pad = torch.zeros((n, v_max - v_l) + tuple(t_local.shape[2:]), dtype=t_local.dtype, device=t_local.device)
        t_local = torch.cat([t_local, pad], dim=1).contiguous()
    out = _all_gather_flat(t_local, group).view((world, n, v_max) + tuple(t_local.shape[2:]))
    if v_max * world == num_views:
        return out.transpose(0, 1).reshape((n, num_views) + tuple(t_local.shape[2:])).contiguous()
    return torch.cat([out[r, :, :hi - lo] for r, (lo, hi) in enumerate(counts)], dim=1).contiguous()


def allreduce_sum_bucketed(tensors, bucket_bytes=64 << 20, group=None):
    """Sum `tensors` (same dtype, modified in place) over the ranks in buckets of about bucket_bytes: the
    buckets are launched asynchronously back to back and waited for at the end."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not dist.is_initialized() or _solo(world) or not tensors:
        return 0
    buckets, cur, cur_bytes = [], [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if cur and cur_bytes + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(t)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    pending = []
    via_host = tensors[0].is_cuda and dist.get_backend(group) == "gloo"
    for b in buckets:
        flat = torch.cat([t.reshape(-1) for t in b])
        if via_host:
            flat = flat.cpu()
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for work, flat, b in pending:
        work.wait()
        off = 0
        for t in b:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return len(buckets)


class OverlappedFlatAllReduce:
    """Sum over the ranks of flat[0:hi] — a gradient buffer that becomes final from its END towards offset 0 while the
    backward pass runs (TrainGVCNN.backward_backbone(progress=...)).  progress(lo) says "flat[lo:hi] is final": whenever
    at least bucket_bytes have become final since the last launch, that slice is all-reduced IN PLACE, asynchronously
    (no staging copy: the bucket IS the slice); finish() launches what is left and waits for everything.  On RCCL the
    collectives run on the communicator's stream behind the kernels enqueued so far and overlap the rest of the backward
    pass; xGMI is point-to-point, so few large slices (64 MiB) are used, not one message per variable.  A gloo group with
    device tensors (the one-device control-flow checks) falls back to one synchronous reduction in finish()."""

    def __init__(self, flat, hi, bucket_bytes=64 << 20, group=None):
        self.flat, self.hi, self.group = flat, int(hi), group
        self.bucket = max(1, int(bucket_bytes) // flat.element_size())
        self.sent_lo = self.hi                                # flat[sent_lo:hi] has been launched
        self.pending = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.solo = not dist.is_initialized() or _solo(self.world)
        self.sync_only = not self.solo and flat.is_cuda and dist.get_backend(group) == "gloo"
        self.launches = 0

    def _launch(self, lo, hi):
        if hi > lo:
            self.pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launches += 1

    def progress(self, lo):
        lo = max(0, min(int(lo), self.sent_lo))
        if self.solo or self.sync_only:
            return
        if self.sent_lo - lo >= self.bucket:
            self._launch(lo, self.sent_lo)
            self.sent_lo = lo

    def finish(self):
        if self.solo:
            return 0
        if self.sync_only:
            h = self.flat[:self.hi].cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            self.flat[:self.hi].copy_(h)
            return 1
        self._launch(0, self.sent_lo)
        self.sent_lo = 0
        for w in self.pending:
            w.wait()
        self.pending = []
        return self.launches


def allreduce_sum_(t, group=None):
    """In-place sum over the ranks (a gloo group is fed through host memory)."""
    if not dist.is_initialized() or _solo(dist.get_world_size(group)):
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def moving_average_count(num_shapes_local, shape_world, hw):
    """Samples behind one (view, channel) batch statistic of a step: the LOCAL shapes times the ranks that share the view
    (their sums were all-reduced: TrainGVCNN.bn_sync) times the pixels — the n of the unbiased-variance factor n / (n-1)
    that the moving-variance update uses (fused batch norm, SURVEY a-note 4); the same number TrainGVCNN._count() feeds
    the normalisation."""
    return int(num_shapes_local) * int(shape_world) * int(hw)


def hybrid_grid(num_views, world_size):
    """(view_groups, shape_shards) with view_groups * shape_shards == world_size: the largest divisor of the world size
    that also divides the number of views becomes the number of view groups, the rest cuts the shapes.  12 views on 8
    ranks -> (4, 2): every rank owns 3 views of half the shapes — exactly 1/8 of the work (view sharding alone deals
    2,2,2,2,1,1,1,1 views: 0.75 efficiency by construction).  Rank r sits at (view group r // shape_shards, shape
    shard r % shape_shards): the ranks that all-reduce BatchNorm sums (same views) are neighbours."""
    vg = max(d for d in range(1, world_size + 1) if world_size % d == 0 and num_views % d == 0)
    return vg, world_size // vg


def hybrid_coords(num_views, world_size, rank):
    vg, sh = hybrid_grid(num_views, world_size)
    return rank // sh, rank % sh


def hybrid_groups(num_views, world_size, rank):
    """(shape_group, view_group) of `rank` as torch.distributed groups; every rank creates every group, in the same
    order (a requirement of new_group).  shape_group: the ranks with the SAME views (they share BatchNorm statistics);
    view_group: the ranks with the SAME shapes (they exchange view descriptors)."""
    vg, sh = hybrid_grid(num_views, world_size)
    mine_s = mine_v = None
    for g in range(vg):
        ranks = [g * sh + k for k in range(sh)]
        grp = dist.new_group(ranks)
        if rank in ranks:
            mine_s = grp
    for k in range(sh):
        ranks = [g * sh + k for g in range(vg)]
        grp = dist.new_group(ranks)
        if rank in ranks:
            mine_v = grp
    return mine_s, mine_v


class ShardedTrainGVCNN:
    """Data-parallel training step around a per-rank TrainGVCNN.

    mode='views'  (default): the engine owns views [view_offset, view_offset + V_l) of every shape (built with
                  num_views = V_l, head_views = V): BatchNorm statistics stay local, descriptors are gathered.
    mode='shapes': the engine owns N_l shapes with all their views (the usual data parallelism, any world size,
                  even work).  A view's images now live on every rank, so each train-mode BatchNorm all-reduces its
                  per-(view, channel) sums — forward (sum, sum of squares) and backward (sum g, sum g*zhat): two small
                  collectives per layer — and normalises with the GLOBAL counts: exactly the reference's statistics
                  over the whole batch.  The scorer responses are gathered like in inference (batch-mean scores);
                  pooling, classifier and loss run on the local shapes with the 1/world factor of the global mean.
    mode='hybrid': view groups x shape shards (hybrid_grid): the engine owns V / view_groups views of N / shape_shards
                  shapes, so every rank does exactly 1/world of the work for any world size that shares a factor with
                  V.  BatchNorm sums are all-reduced only inside the shape group (2 ranks for 12 views on 8 GPUs), the
                  view descriptors are gathered inside the view group, the scorer responses over the whole world;
                  variable gradients are all-reduced over the world, beta/gamma (already summed over the shapes by the
                  BatchNorm exchange) over the view group, the classifier (identical inside a view group) over the
                  shape group."""

    def __init__(self, engine, group=None, bucket_bytes=64 << 20, mode="views", overlap_grads=True):
        if mode not in ("views", "shapes", "hybrid"):
            raise ValueError(mode)
        self.eng = engine
        self.group = group
        self.mode = mode
        self.bucket_bytes = bucket_bytes
        self.overlap_grads = bool(overlap_grads)   # filter gradients are all-reduced while the backward pass still runs
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.solo = not dist.is_initialized() or _solo(self.world)
        if mode == "hybrid":
            assert group is None, "hybrid sharding builds its own sub-groups of the default group"
            self.vg, self.sh = hybrid_grid(engine.Vh, self.world)
            gi, si = hybrid_coords(engine.Vh, self.world, self.rank)
            assert engine.V * self.vg == engine.Vh and engine.view_offset == gi * engine.V, "engine built for other views"
            self.shape_group = self.view_group = None
            if self.world > 1:
                self.shape_group, self.view_group = hybrid_groups(engine.Vh, self.world, self.rank)
            if self.sh > 1:
                engine.shape_world = self.sh
                engine.bn_sync = lambda accum: allreduce_sum_(accum, self.shape_group)
        elif mode == "views":
            assert (engine.view_offset, engine.view_offset + engine.V) == view_shard_range(engine.Vh, self.world, self.rank)
        else:
            assert engine.Vh == engine.V and engine.view_offset == 0
            if not self.solo:
                engine.shape_world = self.world
                engine.bn_sync = lambda accum: allreduce_sum_(accum, self.group)

    def _filter_reducer(self, group):
        """The conv filters' gradients — the first _n_wd elements of the engine's flat gradient buffer, >95 % of the bytes —
        are reduced WHILE the backward pass runs (OverlappedFlatAllReduce); None when the engine cannot report progress
        (branch lanes) or there is nobody to reduce with."""
        eng = self.eng
        if self.solo or not self.overlap_grads or not hasattr(eng, "_flat_g") or not getattr(eng, "_g_monotone", False) \
                or getattr(eng, "_lane_streams", None) is not None:
            return None
        return OverlappedFlatAllReduce(eng._flat_g, eng._n_wd, self.bucket_bytes, group)

    def _train_step_shapes(self, views_local, labels_local, lr, mu, weight_decay, check):
        eng = self.eng
        eng.forward_backbone(views_local)
        r_all = gather_scores(eng.score_partial(), self.group)              # [P*N_l*V], global shape-major order
        # batch-mean scores over the GLOBAL batch (identical on every rank), then the local head
        _lib_scores = eng.lib.gv_view_score_finalize
        from . import _lib
        from .model import _st
        _lib.check(_lib_scores(r_all.data_ptr(), eng.N * self.world, eng.V, _lib.GV_ORDER_SHAPE_MAJOR,
                               eng.scores.data_ptr(), _st()), "score finalize (global)")
        eng.forward_head(labels_local, check=check, r_img=None, scores_ready=True)
        eng.backward_head()
        red = self._filter_reducer(self.group)
        eng.backward_backbone(progress=red.progress if red else None)
        # BN beta/gamma gradients come out of the all-reduced sums: already global.  Everything else is a local sum.
        local = [g for k, g in eng.grads.items() if not k.endswith(("/beta", "/gamma")) and not (red and k.endswith("/weights"))]
        if red:
            red.finish()
        allreduce_sum_bucketed(local, self.bucket_bytes, self.group)
        eng.update_moving_averages()
        eng.apply_momentum(lr, mu, weight_decay)
        return eng.loss

    def _train_step_hybrid(self, views_local, labels_local, lr, mu, weight_decay, check):
        """views_local [N / shape_shards, V / view_groups, H, W, 3]: this rank's views of this rank's shapes; labels_local
        [N / shape_shards]."""
        from . import _lib
        from .model import _st
        eng = self.eng
        f = eng.final
        n_l, v_l, V = eng.N, eng.V, eng.Vh
        eng.forward_backbone(views_local)                      # BatchNorm sums meet inside the shape group (bn_sync)
        r_loc = eng.score_partial().view(n_l, v_l)
        r_views = gather_views(r_loc, self.view_group, V) if self.vg > 1 else r_loc                    # [N_l, V]
        r_all = _all_gather_flat(r_views, self.shape_group) if self.sh > 1 else r_views               # [N, V], shape order
        _lib.check(eng.lib.gv_view_score_finalize(r_all.data_ptr(), n_l * self.sh, V, _lib.GV_ORDER_SHAPE_MAJOR,
                                                  eng.scores.data_ptr(), _st()), "score finalize (global)")
        F_loc = eng.view(f).view(n_l, v_l, f.h, f.w, f.c)
        F_all = gather_views(F_loc, self.view_group, V) if self.vg > 1 else F_loc
        eng.forward_head(labels_local, check=check, F=F_all if self.vg > 1 else None, r_img=r_views.reshape(-1),
                         scores_ready=True)
        if self.vg > 1:
            dF = torch.zeros_like(F_all)
            eng.backward_head(dF=dF)
            lo = eng.view_offset
            eng.final_grad().copy_(dF[:, lo:lo + v_l])
        else:
            eng.backward_head()
        red = self._filter_reducer(None)
        eng.backward_backbone(progress=red.progress if red else None)
        bn = [g for k, g in eng.grads.items() if k.endswith(("/beta", "/gamma"))]
        cls = [eng.grads[k] for k in eng.cls_names]
        shared = [g for k, g in eng.grads.items() if not k.endswith(("/beta", "/gamma")) and k not in eng.cls_names
                  and not (red and k.endswith("/weights"))]
        if red:
            red.finish()
        allreduce_sum_bucketed(shared, self.bucket_bytes, None)                    # every view, every shape
        if self.vg > 1:
            allreduce_sum_bucketed(bn, self.bucket_bytes, self.view_group)         # shapes already summed by bn_sync
        if self.sh > 1:
            allreduce_sum_bucketed(cls, self.bucket_bytes, self.shape_group)       # identical inside a view group
        if self.vg > 1:
            self.update_moving_averages_views(group=self.view_group)
        else:
            eng.update_moving_averages()
        eng.apply_momentum(lr, mu, weight_decay)
        return eng.loss

    def train_step(self, views_local, labels, lr=1e-3, mu=0.9, weight_decay=0.0, check=False):
        """mode='views': views_local [N, V_l, H, W, 3] (this rank's views of every shape), labels [N] (same on all
        ranks).  mode='shapes': views_local [N_l, V, H, W, 3], labels [N_l] (this rank's shapes)."""
        if self.solo:                                     # nothing to exchange: the engine's own step
            eng = self.eng
            eng.forward(views_local, labels, check=check)
            eng.backward()
            eng.update_moving_averages()
            eng.apply_momentum(lr, mu, weight_decay)
            return eng.loss
        if self.mode == "shapes":
            return self._train_step_shapes(views_local, labels, lr, mu, weight_decay, check)
        if self.mode == "hybrid":
            return self._train_step_hybrid(views_local, labels, lr, mu, weight_decay, check)
        eng = self.eng
        f = eng.final
        eng.forward_backbone(views_local)
        r_all = gather_views(eng.score_partial().view(eng.N, eng.V), self.group, eng.Vh)
        F_all = gather_views(eng.view(f).view(eng.N, eng.V, f.h, f.w, f.c), self.group, eng.Vh)
        eng.forward_head(labels, check=check, F=F_all, r_img=r_all.reshape(-1))
        dF = torch.zeros_like(F_all)
        eng.backward_head(dF=dF)
        lo = eng.view_offset
        eng.final_grad().copy_(dF[:, lo:lo + eng.V])
        red = self._filter_reducer(self.group)
        eng.backward_backbone(progress=red.progress if red else None)
        # classifier gradients are identical on every rank (the head ran on the gathered data): not reduced
        shared = [g for k, g in eng.grads.items() if k not in eng.cls_names and not (red and k.endswith("/weights"))]
        if red:
            red.finish()
        allreduce_sum_bucketed(shared, self.bucket_bytes, self.group)
        self.update_moving_averages_views()
        eng.apply_momentum(lr, mu, weight_decay)
        return eng.loss

    def update_moving_averages_views(self, decay=None, group=None):
        """BN moving averages in view-sharded mode: the reference applies one update per view graph copy, so every
        rank needs the batch statistics of ALL views — one all-gather of every layer's [V_l, c] means and variances
        (packed into a single message), then the V sequential updates run identically on every rank."""
        from . import _lib
        from .model import _st
        eng = self.eng
        if self.solo:                                     # all views are local: the engine's own (one-launch) update
            return eng.update_moving_averages(decay)
        if decay is None:
            decay = 0.9997 if eng.backbone == "inception_v3" else 0.997
        bns = [op for op in eng.plan.ops if op["kind"] == "bn"]
        packed = torch.cat([torch.cat([op["stat"]["mean"], op["stat"]["var"]], dim=1) for op in bns], dim=1)  # [V_l, sum 2c]
        grp = group if group is not None else self.group
        full = gather_views(packed.unsqueeze(0), grp, eng.Vh)[0]                                               # [V, sum 2c]
        # ONE launch over all layers: the jobs point straight into the gathered message (row stride = its width)
        full = full.contiguous()
        ld = full.shape[1]
        if getattr(self, "_mv_counts", None) is None:
            self._mv_counts = {}
        jobs, blocks, off = [], [], 0
        for op in bns:
            c, hw = op["x"].c, op["x"].h * op["x"].w
            # samples per (view, channel) statistic: the sums were reduced over the rank's shape group (bn_sync), so the
            # count — and the unbiased-variance factor n/(n-1) of the update — is the GLOBAL one, as eng._count() uses
            key = (hw, eng.shape_world)
            if key not in self._mv_counts:
                self._mv_counts[key] = torch.full((eng.Vh,), moving_average_count(eng.N, eng.shape_world, hw),
                                                  dtype=torch.int32, device=full.device)
            jobs.append(_lib.BnMovingJob(full.data_ptr() + 4 * off, full.data_ptr() + 4 * (off + c),
                                         self._mv_counts[key].data_ptr(),
                                         eng.params[op["name"] + "/moving_mean"].data_ptr(),
                                         eng.params[op["name"] + "/moving_variance"].data_ptr(), c, len(blocks), ld, 0))
            blocks.extend([len(jobs) - 1] * ((c + 255) // 256))
            off += 2 * c
        jd = torch.frombuffer(bytearray(b"".join(bytes(j) for j in jobs)), dtype=torch.uint8).to(full.device)
        bj = torch.tensor(blocks, dtype=torch.int32, device=full.device)
        _lib.check(eng.lib.gv_bn_update_moving_batched(jd.data_ptr(), len(jobs), bj.data_ptr(), bj.numel(), eng.Vh,
                                                       float(decay), _st()), "bn_update_moving_batched")
        self._mv_keep = (full, jd, bj)                    # alive until the launch above has run
