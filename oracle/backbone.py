"""Oracle: the two per-view backbones of the reference, restated with
torch-functional ops on CPU in fp32 with TensorFlow/slim semantics.

TEST INFRASTRUCTURE — see oracle/__init__.py.  PARITY UNPINNED for the float
path (TensorFlow cannot run here); cross-checked against oracle/naive.c.

Tensors are NHWC at every function boundary (the reference's layout);
weights are HWIO `[kh, kw, Cin, Cout]` under their slim variable names.

TF/slim semantics encoded here (SURVEY §8 a-notes):
 * conv = cross-correlation; `normalizer_fn` set => no bias, BN follows;
   `normalizer_fn=None` => bias.
 * SAME: out=ceil(in/s), pad_total=max((out-1)s+k-in,0), before=pad_total//2.
 * avg_pool SAME divides by the number of valid taps; max_pool pads with -inf.
 * BN inference: (x-mean)*rsqrt(var+eps)*gamma+beta.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# parameter access: `P` is a dict name -> torch tensor.  When `P` is a
# ParamRecorder the network is only traced for names/shapes.
# --------------------------------------------------------------------------
class ParamRecorder(dict):
    """Records (name -> shape) for every variable the network asks for."""

    def __init__(self):
        super().__init__()
        self.shapes = {}

    def want(self, name, shape):
        self.shapes[name] = tuple(int(s) for s in shape)
        if name not in self:
            self[name] = torch.zeros(self.shapes[name], dtype=torch.float32)
        return self[name]


def _get(P, name, shape):
    if isinstance(P, ParamRecorder):
        return P.want(name, shape)
    t = P[name]
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), tuple(shape))
    return t


# --------------------------------------------------------------------------
# TF-semantics primitives (NHWC in / NHWC out)
# --------------------------------------------------------------------------
def same_pads(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv2d(x, w_hwio, stride=1, padding="SAME", bias=None):
    """tf.nn.conv2d with NHWC input and HWIO filter."""
    kh, kw = int(w_hwio.shape[0]), int(w_hwio.shape[1])
    xn = _nchw(x)
    if padding == "SAME":
        pt, pb = same_pads(x.shape[1], kh, stride)
        pl, pr = same_pads(x.shape[2], kw, stride)
        if pt or pb or pl or pr:
            xn = F.pad(xn, (pl, pr, pt, pb))
    elif padding != "VALID":
        pt, pb, pl, pr = padding
        xn = F.pad(xn, (pl, pr, pt, pb))
    w = w_hwio.permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(xn, w, bias=bias, stride=stride)
    return _nhwc(y)


def batch_norm_inference(x, mean, var, beta, gamma, eps):
    inv = torch.rsqrt(var + eps)
    if gamma is not None:
        inv = inv * gamma
    return (x - mean) * inv + beta


def batch_norm_train_grouped(x, beta, gamma, eps, groups):
    """Train-mode BN with statistics per view copy (nets/model.py:129-141 builds
    V graph copies, each normalising over its own N*h*w).  `groups[b]` is the
    view index of image b; returns (y, mean[Vg,C], biased var[Vg,C])."""
    g = torch.as_tensor(groups, dtype=torch.long)
    ng = int(g.max().item()) + 1
    y = torch.empty_like(x)
    means, vars_ = [], []
    for v in range(ng):
        sel = (g == v).nonzero().flatten()
        xv = x[sel]
        m = xv.mean(dim=(0, 1, 2))
        var = xv.var(dim=(0, 1, 2), unbiased=False)
        inv = torch.rsqrt(var + eps)
        if gamma is not None:
            inv = inv * gamma
        y[sel] = (xv - m) * inv + beta
        means.append(m)
        vars_.append(var)
    return y, torch.stack(means), torch.stack(vars_)


def max_pool2d(x, k, stride, padding="VALID"):
    xn = _nchw(x)
    if padding == "SAME":
        pt, pb = same_pads(x.shape[1], k, stride)
        pl, pr = same_pads(x.shape[2], k, stride)
        if pt or pb or pl or pr:
            xn = F.pad(xn, (pl, pr, pt, pb), value=float("-inf"))
    return _nhwc(F.max_pool2d(xn, k, stride))


def avg_pool2d_same3(x):
    """slim.avg_pool2d(net, [3,3]) under stride=1, padding='SAME'
    (inception_v3.py:134-135,152): divisor counts valid taps only."""
    return _nhwc(F.avg_pool2d(_nchw(x), 3, 1, padding=1, count_include_pad=False))


# --------------------------------------------------------------------------
# Inception-v3 base — nets/inception_v3.py:29-410 + inception_utils.py:30-78
# --------------------------------------------------------------------------
INCEPTION_BN_EPS = 0.001      # inception_utils.py:33
RESNET_BN_EPS = 1e-5          # resnet_utils.py:200

INCEPTION_ENDPOINTS = [
    "Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3", "MaxPool_3a_3x3",
    "Conv2d_3b_1x1", "Conv2d_4a_3x3", "MaxPool_5a_3x3", "Mixed_5b", "Mixed_5c",
    "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e",
    "Mixed_7a", "Mixed_7b", "Mixed_7c"]


class _BNMode:
    """inference: moving stats.  train: batch stats per view group."""

    def __init__(self, is_training=False, groups=None, stats_out=None):
        self.is_training = is_training
        self.groups = groups
        self.stats_out = stats_out


def _slim_conv_bn_relu(P, x, scope, cout, k, stride, padding, mode, eps, scale):
    """slim.conv2d under the inception/resnet arg-scope: conv(no bias) -> BN -> ReLU
    (inception_utils.py:70-78, resnet_utils.py:233-240)."""
    kh, kw = k
    cin = int(x.shape[3])
    w = _get(P, scope + "/weights", (kh, kw, cin, cout))
    y = conv2d(x, w, stride, padding)
    beta = _get(P, scope + "/BatchNorm/beta", (cout,))
    gamma = _get(P, scope + "/BatchNorm/gamma", (cout,)) if scale else None
    mean = _get(P, scope + "/BatchNorm/moving_mean", (cout,))
    var = _get(P, scope + "/BatchNorm/moving_variance", (cout,))
    if mode.is_training:
        y, bm, bv = batch_norm_train_grouped(y, beta, gamma, eps, mode.groups)
        if mode.stats_out is not None:
            mode.stats_out[scope + "/BatchNorm"] = (bm, bv)
    else:
        y = batch_norm_inference(y, mean, var, beta, gamma, eps)
    return torch.relu(y)


def inception_v3_base(x, P, final_endpoint="Mixed_7c", mode=None, scope="InceptionV3"):
    """nets/inception_v3.py:93-410.  Returns (net, end_points)."""
    if final_endpoint not in INCEPTION_ENDPOINTS:
        raise ValueError("Unknown final endpoint %s" % final_endpoint)   # inception_v3.py:410
    mode = mode or _BNMode()
    ep = {}

    def conv(net, name, cout, k, stride=1, padding="SAME"):
        if isinstance(k, int):
            k = (k, k)
        return _slim_conv_bn_relu(P, net, scope + "/" + name, cout, k, stride, padding,
                                  mode, INCEPTION_BN_EPS, scale=False)

    def done(name, net):
        ep[name] = net
        return name == final_endpoint

    # stem: stride=1, padding='VALID' defaults (inception_v3.py:94-95)
    net = conv(x, "Conv2d_1a_3x3", 32, 3, 2, "VALID")            # :97
    if done("Conv2d_1a_3x3", net): return net, ep
    net = conv(net, "Conv2d_2a_3x3", 32, 3, 1, "VALID")          # :102
    if done("Conv2d_2a_3x3", net): return net, ep
    net = conv(net, "Conv2d_2b_3x3", 64, 3, 1, "SAME")           # :107
    if done("Conv2d_2b_3x3", net): return net, ep
    net = max_pool2d(net, 3, 2, "VALID")                         # :112
    if done("MaxPool_3a_3x3", net): return net, ep
    net = conv(net, "Conv2d_3b_1x1", 80, 1, 1, "VALID")          # :117
    if done("Conv2d_3b_1x1", net): return net, ep
    net = conv(net, "Conv2d_4a_3x3", 192, 3, 1, "VALID")         # :122
    if done("Conv2d_4a_3x3", net): return net, ep
    net = max_pool2d(net, 3, 2, "VALID")                         # :127
    if done("MaxPool_5a_3x3", net): return net, ep

    # Mixed_5b/5c/5d (inception_v3.py:137-204); scope names differ per block.
    def mixed5(net, name, b1a, b1b, pool_depth):
        s = name + "/"
        b0 = conv(net, s + "Branch_0/Conv2d_0a_1x1", 64, 1)
        b1 = conv(net, s + "Branch_1/" + b1a, 48, 1)
        b1 = conv(b1, s + "Branch_1/" + b1b, 64, 5)
        b2 = conv(net, s + "Branch_2/Conv2d_0a_1x1", 64, 1)
        b2 = conv(b2, s + "Branch_2/Conv2d_0b_3x3", 96, 3)
        b2 = conv(b2, s + "Branch_2/Conv2d_0c_3x3", 96, 3)
        b3 = avg_pool2d_same3(net)
        b3 = conv(b3, s + "Branch_3/Conv2d_0b_1x1", pool_depth, 1)
        return torch.cat([b0, b1, b2, b3], dim=3)

    net = mixed5(net, "Mixed_5b", "Conv2d_0a_1x1", "Conv2d_0b_5x5", 32)   # :137-157
    if done("Mixed_5b", net): return net, ep
    net = mixed5(net, "Mixed_5c", "Conv2d_0b_1x1", "Conv_1_0c_5x5", 64)   # :160-181
    if done("Mixed_5c", net): return net, ep
    net = mixed5(net, "Mixed_5d", "Conv2d_0a_1x1", "Conv2d_0b_5x5", 64)   # :184-204
    if done("Mixed_5d", net): return net, ep

    # Mixed_6a (inception_v3.py:207-223)
    s = "Mixed_6a/"
    b0 = conv(net, s + "Branch_0/Conv2d_1a_1x1", 384, 3, 2, "VALID")
    b1 = conv(net, s + "Branch_1/Conv2d_0a_1x1", 64, 1)
    b1 = conv(b1, s + "Branch_1/Conv2d_0b_3x3", 96, 3)
    b1 = conv(b1, s + "Branch_1/Conv2d_1a_1x1", 96, 3, 2, "VALID")
    b2 = max_pool2d(net, 3, 2, "VALID")
    net = torch.cat([b0, b1, b2], dim=3)
    if done("Mixed_6a", net): return net, ep

    # Mixed_6b..6e (inception_v3.py:226-338)
    def mixed6(net, name, d):
        s = name + "/"
        b0 = conv(net, s + "Branch_0/Conv2d_0a_1x1", 192, 1)
        b1 = conv(net, s + "Branch_1/Conv2d_0a_1x1", d, 1)
        b1 = conv(b1, s + "Branch_1/Conv2d_0b_1x7", d, (1, 7))
        b1 = conv(b1, s + "Branch_1/Conv2d_0c_7x1", 192, (7, 1))
        b2 = conv(net, s + "Branch_2/Conv2d_0a_1x1", d, 1)
        b2 = conv(b2, s + "Branch_2/Conv2d_0b_7x1", d, (7, 1))
        b2 = conv(b2, s + "Branch_2/Conv2d_0c_1x7", d, (1, 7))
        b2 = conv(b2, s + "Branch_2/Conv2d_0d_7x1", d, (7, 1))
        b2 = conv(b2, s + "Branch_2/Conv2d_0e_1x7", 192, (1, 7))
        b3 = avg_pool2d_same3(net)
        b3 = conv(b3, s + "Branch_3/Conv2d_0b_1x1", 192, 1)
        return torch.cat([b0, b1, b2, b3], dim=3)

    for name, d in (("Mixed_6b", 128), ("Mixed_6c", 160), ("Mixed_6d", 160), ("Mixed_6e", 192)):
        net = mixed6(net, name, d)
        if done(name, net): return net, ep

    # Mixed_7a (inception_v3.py:341-360)
    s = "Mixed_7a/"
    b0 = conv(net, s + "Branch_0/Conv2d_0a_1x1", 192, 1)
    b0 = conv(b0, s + "Branch_0/Conv2d_1a_3x3", 320, 3, 2, "VALID")
    b1 = conv(net, s + "Branch_1/Conv2d_0a_1x1", 192, 1)
    b1 = conv(b1, s + "Branch_1/Conv2d_0b_1x7", 192, (1, 7))
    b1 = conv(b1, s + "Branch_1/Conv2d_0c_7x1", 192, (7, 1))
    b1 = conv(b1, s + "Branch_1/Conv2d_1a_3x3", 192, 3, 2, "VALID")
    b2 = max_pool2d(net, 3, 2, "VALID")
    net = torch.cat([b0, b1, b2], dim=3)
    if done("Mixed_7a", net): return net, ep

    # Mixed_7b / 7c (inception_v3.py:362-409); 7c names its 3x1 convs differently.
    def mixed7(net, name, b1_3x1, b2_names):
        s = name + "/"
        b0 = conv(net, s + "Branch_0/Conv2d_0a_1x1", 320, 1)
        b1 = conv(net, s + "Branch_1/Conv2d_0a_1x1", 384, 1)
        b1 = torch.cat([conv(b1, s + "Branch_1/Conv2d_0b_1x3", 384, (1, 3)),
                        conv(b1, s + "Branch_1/" + b1_3x1, 384, (3, 1))], dim=3)
        b2 = conv(net, s + "Branch_2/Conv2d_0a_1x1", 448, 1)
        b2 = conv(b2, s + "Branch_2/Conv2d_0b_3x3", 384, 3)
        b2 = torch.cat([conv(b2, s + "Branch_2/" + b2_names[0], 384, (1, 3)),
                        conv(b2, s + "Branch_2/" + b2_names[1], 384, (3, 1))], dim=3)
        b3 = avg_pool2d_same3(net)
        b3 = conv(b3, s + "Branch_3/Conv2d_0b_1x1", 192, 1)
        return torch.cat([b0, b1, b2, b3], dim=3)

    net = mixed7(net, "Mixed_7b", "Conv2d_0b_3x1", ("Conv2d_0c_1x3", "Conv2d_0d_3x1"))
    if done("Mixed_7b", net): return net, ep
    net = mixed7(net, "Mixed_7c", "Conv2d_0c_3x1", ("Conv2d_0c_1x3", "Conv2d_0d_3x1"))
    done("Mixed_7c", net)
    return net, ep


# --------------------------------------------------------------------------
# ResNet-v2-50 — nets/resnet_v2.py:52-95,163-189,230-248 + resnet_utils.py
# --------------------------------------------------------------------------
RESNET50_BLOCKS = (("block1", 64, 3, 2), ("block2", 128, 4, 2),
                   ("block3", 256, 6, 2), ("block4", 512, 3, 1))     # resnet_v2.py:239-244


def _bn(P, x, scope, mode, eps, scale=True):
    c = int(x.shape[3])
    beta = _get(P, scope + "/beta", (c,))
    gamma = _get(P, scope + "/gamma", (c,)) if scale else None
    mean = _get(P, scope + "/moving_mean", (c,))
    var = _get(P, scope + "/moving_variance", (c,))
    if mode.is_training:
        y, bm, bv = batch_norm_train_grouped(x, beta, gamma, eps, mode.groups)
        if mode.stats_out is not None:
            mode.stats_out[scope] = (bm, bv)
        return y
    return batch_norm_inference(x, mean, var, beta, gamma, eps)


def conv2d_same(P, x, scope, cout, k, stride, with_bn, mode):
    """resnet_utils.py:70-105: stride 1 -> SAME; stride>1 -> explicit pad
    (pad_beg=(k-1)//2, pad_end=k-1-pad_beg) then VALID."""
    cin = int(x.shape[3])
    if stride == 1:
        padding = "SAME"
    else:
        pad_total = k - 1
        pb = pad_total // 2
        pe = pad_total - pb
        padding = (pb, pe, pb, pe)
    if with_bn:
        return _slim_conv_bn_relu(P, x, scope, cout, (k, k), stride, padding, mode,
                                  RESNET_BN_EPS, scale=True)
    w = _get(P, scope + "/weights", (k, k, cin, cout))
    b = _get(P, scope + "/biases", (cout,))
    return conv2d(x, w, stride, padding, bias=b)


def bottleneck(P, x, scope, depth, depth_bottleneck, stride, mode):
    """nets/resnet_v2.py:52-95."""
    depth_in = int(x.shape[3])
    preact = torch.relu(_bn(P, x, scope + "/preact", mode, RESNET_BN_EPS))      # :75
    if depth == depth_in:
        shortcut = x if stride == 1 else x[:, ::stride, ::stride, :]            # :77, utils :64-67
    else:
        w = _get(P, scope + "/shortcut/weights", (1, 1, depth_in, depth))
        b = _get(P, scope + "/shortcut/biases", (depth,))
        shortcut = conv2d(preact, w, stride, "SAME" if stride == 1 else "VALID", bias=b)  # :79-81
    residual = _slim_conv_bn_relu(P, preact, scope + "/conv1", depth_bottleneck, (1, 1), 1,
                                  "SAME", mode, RESNET_BN_EPS, scale=True)       # :83-84
    residual = conv2d_same(P, residual, scope + "/conv2", depth_bottleneck, 3, stride,
                           True, mode)                                           # :85-86
    w3 = _get(P, scope + "/conv3/weights", (1, 1, depth_bottleneck, depth))
    b3 = _get(P, scope + "/conv3/biases", (depth,))
    residual = conv2d(residual, w3, 1, "SAME", bias=b3)                          # :87-89
    return shortcut + residual                                                   # :91


def resnet_v2_50(x, P, mode=None, scope="resnet_v2_50"):
    """nets/resnet_v2.py:163-189,230-248 up to block4 (postnorm/pool5/logits are
    never fetched by nets/model.py:144-149).  Returns end_points with
    '<scope>/block1..4' and every unit, as slim registers them."""
    mode = mode or _BNMode()
    ep = {}
    net = conv2d_same(P, x, scope + "/conv1", 64, 7, 2, False, mode)             # :178-180
    ep[scope + "/conv1"] = net
    net = max_pool2d(net, 3, 2, "SAME")                                          # :181
    for bname, base, units, bstride in RESNET50_BLOCKS:
        for u in range(units):
            stride = bstride if u == units - 1 else 1                            # :219-226
            sc = "%s/%s/unit_%d/bottleneck_v2" % (scope, bname, u + 1)
            net = bottleneck(P, net, sc, base * 4, base, stride, mode)
            ep[sc] = net
        ep["%s/%s" % (scope, bname)] = net                                       # resnet_utils.py:181
    return net, ep


# --------------------------------------------------------------------------
# synthetic parameters (SURVEY §8 a-note 6) — used to create test inputs
# --------------------------------------------------------------------------
def trace_param_shapes(backbone, height=75, width=75):
    rec = ParamRecorder()
    x = torch.zeros(1, height, width, 3)
    if backbone == "inception_v3":
        inception_v3_base(x, rec)
    elif backbone == "resnet_v2_50":
        resnet_v2_50(x, rec)
    else:
        raise ValueError(backbone)
    return rec.shapes


def init_params(shapes, seed=2, fresh_bn=False):
    """conv weights: truncated normal, std=sqrt(1.3*2/fan_in)
    (slim.variance_scaling_initializer defaults); biases 0; BN either fresh
    (mean 0, var 1, beta 0, gamma 1) or perturbed so folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith("/weights"):
            fan_in = shp[0] * shp[1] * shp[2]
            std = math.sqrt(1.3 * 2.0 / fan_in)
            w = torch.empty(shp)
            torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=g)
            P[name] = w
        elif name.endswith("/biases"):
            P[name] = torch.zeros(shp) if fresh_bn else 0.05 * torch.randn(shp, generator=g)
        elif name.endswith("moving_mean"):
            P[name] = torch.zeros(shp) if fresh_bn else 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("moving_variance"):
            P[name] = torch.ones(shp) if fresh_bn else 0.5 + torch.rand(shp, generator=g)
        elif name.endswith("beta"):
            P[name] = torch.zeros(shp) if fresh_bn else 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("gamma"):
            P[name] = torch.ones(shp) if fresh_bn else 0.75 + 0.5 * torch.rand(shp, generator=g)
        else:
            raise KeyError(name)
    return P
