"""ctypes front-end of oracle/naive.c (test infrastructure, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libgvref.so"])
    return os.path.join(_HERE, "libgvref.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgvref.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def conv2d_nhwc(x, w_hwio, stride, pads, out_hw, bias=None):
    """pads = (pad_t, pad_l); out_hw = (oh, ow)."""
    x, xp = _f(x)
    w, wp = _f(w_hwio)
    nb, ih, iw, cin = x.shape
    kh, kw, _, cout = w.shape
    oh, ow = out_hw
    y = np.empty((nb, oh, ow, cout), dtype=np.float32)
    bp = None
    if bias is not None:
        bias, bp = _f(bias)
    lib().gvref_conv2d_nhwc(xp, nb, ih, iw, cin, wp, kh, kw, cout, stride, stride,
                            pads[0], pads[1], oh, ow, bp,
                            y.ctypes.data_as(C.POINTER(C.c_float)))
    return y


def bn_inference(x, mean, var, beta, gamma, eps, relu):
    x = np.array(x, dtype=np.float32, order="C", copy=True)
    c = x.shape[-1]
    _, mp = _f(mean)
    m_keep = _
    v_keep, vp = _f(var)
    b_keep, bp = _f(beta)
    gp = None
    if gamma is not None:
        g_keep, gp = _f(gamma)
    lib().gvref_bn_inference(x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.size // c), c,
                             mp, vp, bp, gp, C.c_float(eps), int(relu))
    return x


def pool2d_nhwc(x, k, stride, pads, out_hw, mode):
    x, xp = _f(x)
    nb, ih, iw, c = x.shape
    oh, ow = out_hw
    y = np.empty((nb, oh, ow, c), dtype=np.float32)
    lib().gvref_pool2d_nhwc(xp, nb, ih, iw, c, k, k, stride, stride, pads[0], pads[1], oh, ow,
                            0 if mode == "max" else 1, y.ctypes.data_as(C.POINTER(C.c_float)))
    return y


def group_assign(scores, G, num_bins=10):
    s, sp = _f(scores)
    V = s.shape[0]
    gidx = np.empty(V, dtype=np.int32)
    scheme = np.empty((G, V), dtype=np.int32)
    weight = np.empty(G, dtype=np.float32)
    bad = lib().gvref_group_assign(sp, V, G, num_bins,
                                   gidx.ctypes.data_as(C.POINTER(C.c_int32)),
                                   scheme.ctypes.data_as(C.POINTER(C.c_int32)),
                                   weight.ctypes.data_as(C.POINTER(C.c_float)))
    return bad, gidx, scheme, weight


def view_pool_fuse(F, scheme, weight, mode="max", fill=1.0):
    """F [V,N,...] -> (D [G,N,...], S [N,...])."""
    F, Fp = _f(F)
    scheme, sp = _i(scheme)
    weight, wp = _f(weight)
    V, N = F.shape[:2]
    E = int(np.prod(F.shape[2:]))
    G = scheme.shape[0]
    D = np.empty((G, N) + F.shape[2:], dtype=np.float32)
    S = np.empty((N,) + F.shape[2:], dtype=np.float32)
    lib().gvref_view_pool_fuse(Fp, V, N, C.c_int64(E), sp, G, wp, 0 if mode == "max" else 1,
                               C.c_float(fill), D.ctypes.data_as(C.POINTER(C.c_float)),
                               S.ctypes.data_as(C.POINTER(C.c_float)))
    return D, S
