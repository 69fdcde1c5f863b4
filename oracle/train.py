"""Oracle for the training step — TEST INFRASTRUCTURE, see oracle/__init__.py.

Loss and gradients of the reference graph by torch autograd on CPU through the oracle's own forward:
per-view backbone call in train mode (batch-statistics BN over the N images of ONE view,
nets/model.py:129-141), scores -> host group_scheme/group_weight (constants of the backward pass,
train.py:277-288), reduce_max view pooling (gradient split equally among ties, like tf.reduce_max),
group fusion, GAP, Dense, mean sparse-softmax cross-entropy (train.py:145).  PARITY UNPINNED (float).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import backbone as B
from . import grouping as G
from . import model as M


def head(stacked, scheme, weight, Wc, bc, labels):
    """nets/model.py:44-102,163-164 + train.py:145 on a stacked [V,N,h,w,C] tensor (autograd-traceable)."""
    acc = None
    for g in range(scheme.shape[0]):
        idx = np.nonzero(scheme[g])[0]
        if idx.size:
            d = torch.amax(stacked[torch.as_tensor(idx)], dim=0)           # even split among ties
        else:
            d = torch.ones_like(stacked[0])
        term = float(weight[g]) * d
        acc = term if acc is None else acc + term
    S = acc / float(weight.sum())
    logits = S.mean(dim=(1, 2)) @ Wc + bc
    loss = F.cross_entropy(logits, torch.as_tensor(labels, dtype=torch.long))
    return loss, logits, S


def loss_and_grads(inputs, labels, P, H, num_group, backbone="resnet_v2_50", num_bins=10,
                   raw_tap=None, final_tap=None, frozen_bn=False, scheme=None, weight=None):
    """inputs [N,V,H,W,3] float32, labels [N] int64.  Returns dict(loss, grads{name: tensor}, scores, scheme,
    weight, logits, shape_descriptor).  frozen_bn: BatchNorm with the moving statistics (slim's is_training=False
    arithmetic, inception_utils.py / resnet_utils.py arg scopes) while autograd still differentiates every variable — not a
    mode the reference trains in; the well-conditioned form of the same graph used to hold the assembled backward pass.
    scheme / weight: fed instead of derived from this call's own scores — they are placeholders of the backward pass in the
    reference too (train.py:127-128, 277-288), so a sub-batch can be differentiated under the scheme of the batch it came from."""
    n_views = inputs.shape[1]
    raw_tap = raw_tap or M.TAPS[backbone][0]
    final_tap = final_tap or M.TAPS[backbone][1]
    Pg = {k: v.clone().requires_grad_(not k.endswith(("moving_mean", "moving_variance"))) for k, v in P.items()}
    kn, bn = "dense_%d/kernel" % n_views, "dense_%d/bias" % n_views
    Wc = H[kn].clone().requires_grad_(True)
    bc = H[bn].clone().requires_grad_(True)
    views = inputs.permute(1, 0, 2, 3, 4)
    finals, scores = [], []
    for v in range(n_views):
        mode = B._BNMode(not frozen_bn, groups=[0] * views[v].shape[0])
        ep = M.run_backbone(backbone, views[v], Pg, mode)
        nm = "dense" if v == 0 else "dense_%d" % v
        s = G.view_score(ep[raw_tap].detach().numpy(), H[nm + "/kernel"].numpy(), float(H[nm + "/bias"][0]))
        scores.append(np.float32(s))
        ep[final_tap].retain_grad()
        finals.append(ep[final_tap])
    if scheme is None:
        scheme = G.group_scheme([np.array(scores, dtype=np.float32)], num_group, n_views, num_bins)
        weight = G.group_weight(scheme)
    else:
        scheme, weight = np.asarray(scheme), np.asarray(weight, dtype=np.float32)
    stacked = torch.stack(finals, dim=0)                                   # [V,N,h,w,C]
    loss, logits, S = head(stacked, scheme, weight, Wc, bc, labels)
    loss.backward()
    grads = {k: v.grad for k, v in Pg.items() if v.requires_grad and v.grad is not None}
    grads[kn], grads[bn] = Wc.grad, bc.grad
    return dict(finals=[f.detach() for f in finals], final_grads=[f.grad for f in finals],
                loss=float(loss.detach()), grads=grads, scores=scores, scheme=scheme, weight=weight,
                logits=logits.detach().numpy(), shape_descriptor=S.detach().numpy())


def moving_average_update(moving_mean, moving_var, means, variances, counts, decay):
    """slim.batch_norm UPDATE_OPS (train.py:178-186), one per view graph copy in view order; fused batch norm
    feeds the UNBIASED batch variance into the moving variance (SURVEY a-note 4).  numpy fp32."""
    mm = np.asarray(moving_mean, np.float32).copy()
    mv = np.asarray(moving_var, np.float32).copy()
    d = np.float32(decay)
    one = np.float32(1.0)
    for g in range(len(means)):
        n = np.float32(counts[g])
        unb = n / (n - one) if n > 1 else one
        mm = (mm * d + np.asarray(means[g], np.float32) * (one - d)).astype(np.float32)
        mv = (mv * d + (np.asarray(variances[g], np.float32) * unb).astype(np.float32) * (one - d)).astype(np.float32)
    return mm, mv
