"""The float oracle checks itself (TensorFlow cannot arbitrate here — SURVEY §8c):
two independent CPU implementations must agree, shapes must match the comments
written in the reference, parameter counts must match the survey."""
import numpy as np
import pytest
import torch

from oracle import backbone as B
from oracle import naive


def _tf_out(in_size, k, s, padding):
    if padding == "SAME":
        return -(-in_size // s)
    return (in_size - k) // s + 1


# every (kernel, stride, padding) combination on the path
COMBOS = [
    ((3, 3), 2, "VALID"), ((3, 3), 1, "VALID"), ((3, 3), 1, "SAME"), ((1, 1), 1, "SAME"),
    ((5, 5), 1, "SAME"), ((1, 7), 1, "SAME"), ((7, 1), 1, "SAME"), ((1, 3), 1, "SAME"),
    ((3, 1), 1, "SAME"), ((1, 1), 2, "VALID"), ((7, 7), 2, (3, 3, 3, 3)), ((3, 3), 2, (1, 1, 1, 1)),
]


@pytest.mark.parametrize("k,stride,padding", COMBOS)
@pytest.mark.parametrize("size", [(9, 10), (12, 11)])
def test_conv_torch_vs_naive(k, stride, padding, size):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, size[0], size[1], 5, generator=g)
    w = torch.randn(k[0], k[1], 5, 7, generator=g)
    b = torch.randn(7, generator=g)
    y = B.conv2d(x, w, stride, padding, bias=b).numpy()
    if padding == "SAME":
        pt = B.same_pads(size[0], k[0], stride)[0]
        pl = B.same_pads(size[1], k[1], stride)[0]
    elif padding == "VALID":
        pt = pl = 0
    else:
        pt, pl = padding[0], padding[2]
    yn = naive.conv2d_nhwc(x.numpy(), w.numpy(), stride, (pt, pl), y.shape[1:3], b.numpy())
    assert y.shape == yn.shape
    np.testing.assert_allclose(y, yn, rtol=1e-5, atol=1e-5)
    if isinstance(padding, str):
        assert y.shape[1] == _tf_out(size[0], k[0], stride, padding)
        assert y.shape[2] == _tf_out(size[1], k[1], stride, padding)


@pytest.mark.parametrize("size", [(8, 8), (9, 7), (112 // 8, 15)])
def test_pools_torch_vs_naive(size):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, size[0], size[1], 6, generator=g)
    # max 3x3/2 VALID (inception_v3.py:112), SAME (resnet_v2.py:181, asymmetric (0,1) on even sizes)
    for padding in ("VALID", "SAME"):
        y = B.max_pool2d(x, 3, 2, padding).numpy()
        pt = B.same_pads(size[0], 3, 2)[0] if padding == "SAME" else 0
        pl = B.same_pads(size[1], 3, 2)[0] if padding == "SAME" else 0
        yn = naive.pool2d_nhwc(x.numpy(), 3, 2, (pt, pl), y.shape[1:3], "max")
        np.testing.assert_array_equal(y, yn)
    y = B.avg_pool2d_same3(x).numpy()
    yn = naive.pool2d_nhwc(x.numpy(), 3, 1, (1, 1), y.shape[1:3], "avg")
    np.testing.assert_allclose(y, yn, rtol=1e-6, atol=1e-6)
    # corner divisor is 4, not 9
    ones = torch.ones(1, 5, 5, 1)
    np.testing.assert_allclose(B.avg_pool2d_same3(ones).numpy(), 1.0, rtol=1e-6)
    # subsample = x[:, ::2, ::2] = max_pool 1x1/2 (resnet_utils.py:64-67)
    np.testing.assert_array_equal(B.max_pool2d(x, 1, 2).numpy(), x.numpy()[:, ::2, ::2])


def test_resnet_pool1_pad_is_0_1():
    assert B.same_pads(112, 3, 2) == (0, 1)          # SURVEY a-note 2


def test_bn_torch_vs_naive():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 4, 4, 5, generator=g)
    mean, beta = torch.randn(5, generator=g), torch.randn(5, generator=g)
    var, gamma = torch.rand(5, generator=g) + 0.5, torch.rand(5, generator=g) + 0.5
    for gm in (None, gamma):
        y = torch.relu(B.batch_norm_inference(x, mean, var, beta, gm, 1e-3)).numpy()
        yn = naive.bn_inference(x.numpy(), mean.numpy(), var.numpy(), beta.numpy(),
                                None if gm is None else gm.numpy(), 1e-3, True)
        np.testing.assert_allclose(y, yn, rtol=1e-5, atol=1e-6)


def test_inception_shapes_match_reference_comments():
    """nets/inception_v3.py:96-386 comments: 299 -> 149,147,147,73,73,71,35 | 35x35x256,288,288 |
    17x17x768 | 8x8x1280 | 8x8x2048."""
    shapes = B.trace_param_shapes("inception_v3")
    P = B.init_params(shapes, fresh_bn=True)
    x = torch.zeros(1, 299, 299, 3)
    _, ep = B.inception_v3_base(x, P)
    want = {"Conv2d_1a_3x3": (149, 32), "Conv2d_2a_3x3": (147, 32), "Conv2d_2b_3x3": (147, 64),
            "MaxPool_3a_3x3": (73, 64), "Conv2d_3b_1x1": (73, 80), "Conv2d_4a_3x3": (71, 192),
            "MaxPool_5a_3x3": (35, 192), "Mixed_5b": (35, 256), "Mixed_5c": (35, 288),
            "Mixed_5d": (35, 288), "Mixed_6a": (17, 768), "Mixed_6b": (17, 768), "Mixed_6e": (17, 768),
            "Mixed_7a": (8, 1280), "Mixed_7b": (8, 2048), "Mixed_7c": (8, 2048)}
    for k, (hw, c) in want.items():
        assert tuple(ep[k].shape) == (1, hw, hw, c), k
    assert list(ep) == B.INCEPTION_ENDPOINTS
    n_conv = sum(int(np.prod(s)) for n, s in shapes.items() if n.endswith("/weights"))
    assert len([n for n in shapes if n.endswith("/weights")]) == 94
    assert abs(n_conv / 1e6 - 21.75) < 0.05
    with pytest.raises(ValueError):
        B.inception_v3_base(x, P, "Mixed_8a")


def test_inception_224_shapes():
    shapes = B.trace_param_shapes("inception_v3")
    P = B.init_params(shapes, fresh_bn=True)
    _, ep = B.inception_v3_base(torch.zeros(1, 224, 224, 3), P)
    assert tuple(ep["Mixed_6e"].shape) == (1, 12, 12, 768)
    assert tuple(ep["Mixed_7c"].shape) == (1, 5, 5, 2048)
    assert tuple(ep["Conv2d_1a_3x3"].shape) == (1, 111, 111, 32)


def test_resnet_shapes_and_params():
    """nets/resnet_v2.py:112-121: 224 -> 7x7 ; 225 -> 8x8; taps of model.py:144,149."""
    shapes = B.trace_param_shapes("resnet_v2_50")
    P = B.init_params(shapes, fresh_bn=True)
    _, ep = B.resnet_v2_50(torch.zeros(1, 224, 224, 3), P)
    assert tuple(ep["resnet_v2_50/block3"].shape) == (1, 7, 7, 1024)
    assert tuple(ep["resnet_v2_50/block4"].shape) == (1, 7, 7, 2048)
    assert tuple(ep["resnet_v2_50/block1"].shape) == (1, 28, 28, 256)
    _, ep = B.resnet_v2_50(torch.zeros(1, 225, 225, 3), P)
    assert tuple(ep["resnet_v2_50/block4"].shape) == (1, 8, 8, 2048)
    convs = [n for n in shapes if n.endswith("/weights")]
    assert len(convs) == 53
    n_conv = sum(int(np.prod(shapes[n])) for n in convs)
    assert abs(n_conv / 1e6 - 23.45) < 0.05
    assert "resnet_v2_50/block1/unit_1/bottleneck_v2/shortcut/biases" in shapes
    assert "resnet_v2_50/block1/unit_2/bottleneck_v2/shortcut/weights" not in shapes
    assert "resnet_v2_50/postnorm/beta" not in shapes           # never fetched (model.py:144-149)
