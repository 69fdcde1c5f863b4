"""Variables of the GVCNN graph under the names TensorFlow-slim / Keras give them
(SURVEY §5 checkpoint row), and their default initialisers — there are no checkpoints or
datasets in this environment, so benchmarks and tests run on freshly initialised variables.

 * conv `weights`: slim.variance_scaling_initializer() (nets/inception_utils.py:74,
   nets/resnet_utils.py:236) = truncated normal, stddev sqrt(1.3 * 2 / fan_in), HWIO;
 * `biases`: zeros; BatchNorm beta 0, gamma 1, moving_mean 0, moving_variance 1;
 * Keras Dense (nets/model.py:145,164): glorot-uniform kernel, zero bias; one Dense(1) scorer per
   view named dense, dense_1, ... dense_{V-1}, then the classifier dense_V.
"""
import math

import torch


def init_backbone_params(shapes, seed=2, perturb_bn=False):
    """shapes: name -> shape (from BackbonePlan.param_shapes()).  perturb_bn=True draws non-trivial
    BN statistics / biases so that the folded scale/shift path is exercised."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        leaf = name.rsplit("/", 1)[1]
        if leaf == "weights":
            std = math.sqrt(1.3 * 2.0 / (shp[0] * shp[1] * shp[2]))
            w = torch.empty(shp)
            torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=g)
            P[name] = w
        elif leaf == "biases":
            P[name] = 0.05 * torch.randn(shp, generator=g) if perturb_bn else torch.zeros(shp)
        elif leaf == "moving_mean":
            P[name] = 0.1 * torch.randn(shp, generator=g) if perturb_bn else torch.zeros(shp)
        elif leaf == "moving_variance":
            P[name] = 0.5 + torch.rand(shp, generator=g) if perturb_bn else torch.ones(shp)
        elif leaf == "beta":
            P[name] = 0.1 * torch.randn(shp, generator=g) if perturb_bn else torch.zeros(shp)
        elif leaf == "gamma":
            P[name] = 0.75 + 0.5 * torch.rand(shp, generator=g) if perturb_bn else torch.ones(shp)
        else:
            raise KeyError("unknown variable kind: %s" % name)
    return P


def scorer_names(v):
    nm = "dense" if v == 0 else "dense_%d" % v
    return nm + "/kernel", nm + "/bias"


def classifier_names(num_views):
    nm = "dense_%d" % num_views
    return nm + "/kernel", nm + "/bias"


def init_head_params(num_views, raw_channels, final_channels, num_classes, seed=3,
                     spread_scores=False):
    """Keras Dense defaults.  spread_scores=True moves the scorer biases so the V scores land in
    different sub-ranges of (0,1), >= 1e-3 away from bin edges (SURVEY §8d synthetic inputs) —
    with the default initialiser every view falls into the same bin."""
    g = torch.Generator().manual_seed(seed)
    H = {}
    for v in range(num_views):
        kn, bn = scorer_names(v)
        lim = (6.0 / (raw_channels + 1)) ** 0.5
        H[kn] = (torch.rand(raw_channels, 1, generator=g) * 2 - 1) * lim
        H[bn] = torch.zeros(1)
        if spread_scores:
            target = (v + 0.5) / num_views * 0.9 + 0.03
            H[bn] = torch.tensor([target / (1.0 - target)])
            H[kn] = H[kn] * 1e-3
    kn, bn = classifier_names(num_views)
    lim = (6.0 / (final_channels + num_classes)) ** 0.5
    H[kn] = (torch.rand(final_channels, num_classes, generator=g) * 2 - 1) * lim
    H[bn] = torch.zeros(num_classes)
    return H
