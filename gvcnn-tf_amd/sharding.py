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
import torch
import torch.distributed as dist


def shard_range(num_shapes_global, world_size, rank):
    """Shapes [lo, hi) owned by `rank`; the batch must divide evenly (weak scaling: fixed N_l)."""
    if num_shapes_global % world_size != 0:
        raise ValueError("global batch %d does not divide over %d ranks" % (num_shapes_global, world_size))
    n_l = num_shapes_global // world_size
    return rank * n_l, (rank + 1) * n_l


def gather_scores(r_img_local, group=None):
    """All-gather the local scorer responses [N_l*V] -> [P*N_l*V] (global shape-major order)."""
    world = dist.get_world_size(group)
    if world == 1:
        return r_img_local
    out = torch.empty(world * r_img_local.numel(), dtype=r_img_local.dtype, device=r_img_local.device)
    dist.all_gather_into_tensor(out, r_img_local.contiguous(), group=group)
    return out


def gather_descriptors(F_local, group=None):
    """All-gather final view descriptors [N_l, V, h, w, C] -> [P*N_l, V, h, w, C]."""
    world = dist.get_world_size(group)
    if world == 1:
        return F_local
    out = torch.empty((world * F_local.shape[0],) + tuple(F_local.shape[1:]), dtype=F_local.dtype,
                      device=F_local.device)
    dist.all_gather_into_tensor(out, F_local.contiguous(), group=group)
    return out


class ShardedGVCNN:
    """Wraps a per-rank GVCNN engine built for the LOCAL batch (N_l shapes)."""

    def __init__(self, engine, group=None, exchange="allgather"):
        if exchange not in ("allgather", "scores"):
            raise ValueError(exchange)
        self.eng = engine
        self.group = group
        self.exchange = exchange
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def forward(self, views_local, check=True):
        """views_local [N_l, V, H, W, 3].  Returns (scores [V], shape_descriptor, logits): for
        exchange='allgather' over all N_g = P*N_l shapes (identical on every rank), for
        exchange='scores' over the local N_l shapes."""
        eng = self.eng
        eng.run_backbone(views_local)
        eng.compute_scores()                                   # fills eng.r_img (local) + local scores
        if self.world > 1:
            r_all = gather_scores(eng.r_img, self.group)
            eng.finalize_scores(r_all, eng.N * self.world)     # same array, same order on every rank
        eng.assign_groups(check=False)
        if self.exchange == "allgather" and self.world > 1:
            F_all = gather_descriptors(eng.final_view_descriptors(), self.group)
            S, logits = eng.pool_fuse_classify(eng.scheme, eng.weight, F=F_all)
        else:
            S, logits = eng.pool_fuse_classify(eng.scheme, eng.weight)
        if check:
            eng.check_status()
        return eng.scores, S, logits
