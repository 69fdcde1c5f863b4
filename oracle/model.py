"""Oracle: nets/model.py `gvcnn` / `basic` end-to-end on CPU (fp32).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Follows the reference's graph
construction literally: a Python loop over the V views, one backbone call per
view at batch N with shared weights (nets/model.py:126-141), a separate
Dense(1) scorer per view (model.py:145), batch-mean score (model.py:146).
"""
import numpy as np
import torch

from . import backbone as B
from . import grouping as G

# which end points feed the scorer / the view pooling
TAPS = {
    "resnet_v2_50": ("resnet_v2_50/block3", "resnet_v2_50/block4"),   # model.py:144,149
    "inception_v3": ("Mixed_6e", "Mixed_7c"),                         # model.py:193 (+ default raw tap)
}


def run_backbone(name, x, P, mode=None):
    if name == "resnet_v2_50":
        return B.resnet_v2_50(x, P, mode)[1]
    if name == "inception_v3":
        return B.inception_v3_base(x, P, "Mixed_7c", mode)[1]
    raise ValueError(name)


def init_head_params(num_views, raw_channels, final_channels, num_classes, seed=3,
                     spread_scores=True):
    """Keras Dense defaults: glorot-uniform kernel, zero bias (model.py:145,164).
    With spread_scores the V scorer kernels/biases are rescaled so the scores
    land in different bins, away from bin edges (SURVEY §8d synthetic inputs)."""
    g = torch.Generator().manual_seed(seed)
    H = {}
    for v in range(num_views):
        nm = "dense" if v == 0 else "dense_%d" % v
        lim = (6.0 / (raw_channels + 1)) ** 0.5
        H[nm + "/kernel"] = (torch.rand(raw_channels, 1, generator=g) * 2 - 1) * lim
        H[nm + "/bias"] = torch.zeros(1)
        if spread_scores:
            # bias dominates: r ~= bias => score = |r|/(1+|r|) spread over (0,1)
            target = (v + 0.5) / num_views * 0.9 + 0.03
            H[nm + "/bias"] = torch.tensor([target / (1.0 - target)])
            H[nm + "/kernel"] *= 1e-3
    nm = "dense_%d" % num_views
    lim = (6.0 / (final_channels + num_classes)) ** 0.5
    H[nm + "/kernel"] = (torch.rand(final_channels, num_classes, generator=g) * 2 - 1) * lim
    H[nm + "/bias"] = torch.zeros(num_classes)
    return H


def _scorer_names(v):
    nm = "dense" if v == 0 else "dense_%d" % v
    return nm + "/kernel", nm + "/bias"


def gvcnn_scores(inputs, P, H, backbone="resnet_v2_50", raw_tap=None, final_tap=None,
                 is_training=False):
    """Phase 1 of the caller protocol (train.py:270-276): per-view scores and
    the final view descriptors.  inputs [N,V,H,W,3] float32 torch tensor."""
    n_views = inputs.shape[1]
    raw_tap = raw_tap or TAPS[backbone][0]
    final_tap = final_tap or TAPS[backbone][1]
    views = inputs.permute(1, 0, 2, 3, 4)                     # model.py:128
    scores, finals = [], []
    for v in range(n_views):
        batch_view = views[v]                                  # model.py:130
        mode = B._BNMode(is_training, groups=[0] * batch_view.shape[0]) if is_training else None
        ep = run_backbone(backbone, batch_view, P, mode)
        kn, bn = _scorer_names(v)
        s = G.view_score(ep[raw_tap].numpy(), H[kn].numpy(), float(H[bn][0]))   # model.py:144-147
        scores.append(np.float32(s))
        finals.append(ep[final_tap].numpy())                   # model.py:149
    return scores, finals


def gvcnn(inputs, num_classes, P, H, num_group, backbone="resnet_v2_50",
          raw_tap=None, final_tap=None, num_bins=10, pool="max", empty_fill=1.0,
          is_training=False):
    """Both phases (train.py:270-288): scores -> host group_scheme/group_weight
    (model.py:16-41) -> view_pooling -> group_fusion -> GAP -> Dense.
    Returns (scores list[V], shape_descriptor [N,h,w,C], logits [N,C], scheme, weight)."""
    n_views = inputs.shape[1]
    scores, finals = gvcnn_scores(inputs, P, H, backbone, raw_tap, final_tap, is_training)
    scheme = G.group_scheme([np.array(scores, dtype=np.float32)], num_group, n_views, num_bins)
    weight = G.group_weight(scheme)
    kn = "dense_%d/kernel" % n_views
    bn = "dense_%d/bias" % n_views
    assert H[kn].shape[1] == num_classes
    shape_desc, logits = G.grouping_head(finals, scheme, weight, H[kn].numpy(), H[bn].numpy(),
                                         pool=pool, empty_fill=empty_fill)
    return scores, shape_desc, logits, scheme, weight


def basic(inputs, num_classes, P, H, backbone="resnet_v2_50", final_tap=None):
    """nets/model.py:169-206."""
    n_views = inputs.shape[1]
    final_tap = final_tap or TAPS[backbone][1]
    views = inputs.permute(1, 0, 2, 3, 4)
    finals = [run_backbone(backbone, views[v], P)[final_tap].numpy() for v in range(n_views)]
    kn = "dense_%d/kernel" % n_views
    bn = "dense_%d/bias" % n_views
    return G.basic_head(finals, H[kn].numpy(), H[bn].numpy())


def folded_backbone(inputs, P, backbone):
    """One backbone call at batch N*V (the 'folded' timing of BASELINE.md §3);
    identical to the per-view loop in inference mode."""
    n, v = inputs.shape[:2]
    x = inputs.reshape(n * v, *inputs.shape[2:])
    return run_backbone(backbone, x, P)
