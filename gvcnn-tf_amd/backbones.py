"""The two per-view backbones as launch plans for the HIP library.

Mirrors the structure (and slim variable names) of the reference's
`nets/inception_v3.py:29-410` and `nets/resnet_v2.py:52-248` /
`nets/resnet_utils.py:54-195`, but instead of building V TensorFlow graph copies
(nets/model.py:129-141) it emits ONE ordered list of kernel launches over the
whole folded view batch [N*V, H, W, 3]:

  * every slim.conv2d (+BatchNorm +ReLU | +bias) is one implicit-GEMM launch with
    the BN folded into the epilogue's per-channel scale/shift;
  * the branches of an Inception block write straight into channel slices of the
    block's output buffer (pixel stride = concat width), so no tf.concat copy
    exists (inception_v3.py:155,181,204,222,...);
  * a ResNet unit's `shortcut + residual` (resnet_v2.py:91) is the residual input
    of the conv3 launch, and the NEXT unit's pre-activation BN+ReLU
    (resnet_v2.py:75) is that launch's second output;
  * activations live in a small set of reusable device buffers assigned by
    liveness (288 GB of HBM would hold everything, but reuse keeps the working
    set inside the 256 MiB Infinity Cache for small batches).

Nothing here computes: it describes.  `BackbonePlan.bind()` uploads parameters and
`BackbonePlan.run()` enqueues the launches through the C ABI.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

INCEPTION_BN_EPS = 0.001   # nets/inception_utils.py:33
RESNET_BN_EPS = 1e-5       # nets/resnet_utils.py:200

INCEPTION_ENDPOINTS = [
    "Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3", "MaxPool_3a_3x3", "Conv2d_3b_1x1",
    "Conv2d_4a_3x3", "MaxPool_5a_3x3", "Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b",
    "Mixed_6c", "Mixed_6d", "Mixed_6e", "Mixed_7a", "Mixed_7b", "Mixed_7c"]

SLOT_INPUT, SLOT_WEIGHTS, SLOT_SS, SLOT_ACT0 = 0, 1, 2, 3


def _same_pads(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2


def _out_size(in_size, k, s, padding):
    """padding: 'SAME' | 'VALID' | (before, after) explicit (resnet_utils.py:94-105)."""
    if padding == "SAME":
        return _same_pads(in_size, k, s)
    if padding == "VALID":
        return (in_size - k) // s + 1, 0
    before, after = padding
    return (in_size + before + after - k) // s + 1, before


class TRef:
    """A [nb,h,w,c] NHWC tensor inside a (virtual) buffer: element offset + pixel stride.  p3: the tensor holds every
    fp32 value as three bf16 planes, [pixel][channel/16][plane][16] (6 bytes per value: GV_CONV_Y_P3 / GV_CONV_X_P3);
    offsets and strides stay in channels (multiples of 16)."""
    __slots__ = ("vbuf", "off", "nb", "h", "w", "c", "ld", "p3")

    def __init__(self, vbuf, off, nb, h, w, c, ld, p3=False):
        self.vbuf, self.off, self.nb, self.h, self.w, self.c, self.ld, self.p3 = vbuf, off, nb, h, w, c, ld, p3

    def channels(self, lo, hi):
        assert 0 <= lo < hi <= self.c
        assert not self.p3 or (lo % 16 == 0 and hi % 16 == 0), "a three-plane tensor is sliced on 16-channel groups"
        return TRef(self.vbuf, self.off + lo, self.nb, self.h, self.w, hi - lo, self.ld, self.p3)

    @property
    def npix(self):
        return self.nb * self.h * self.w

    def __repr__(self):
        return "TRef(v%d+%d [%d,%d,%d,%d] ld=%d%s)" % (self.vbuf, self.off, self.nb, self.h, self.w,
                                                        self.c, self.ld, " p3" if self.p3 else "")


class DeferredPreact:
    """relu(bn(t)) that no kernel has written: the scale / shift offsets of the folded BatchNorm travel with the tensor
    and the consuming convolution's loader applies them (BackbonePlan.conv, gv_conv2d_fwd_xpre)."""
    __slots__ = ("t", "scale_off", "shift_off")

    def __init__(self, t, scale_off, shift_off):
        self.t, self.scale_off, self.shift_off = t, scale_off, shift_off


class BackbonePlan:
    """Collects ops symbolically, then lowers them to a native gv_plan."""

    def __init__(self, nb, height, width, dtype=_lib.GV_F32, math_mode=_lib.GV_MATH_F32):
        self.lib = _lib.load()
        self.nb, self.height, self.width, self.dtype = nb, height, width, dtype
        self.esz = 4 if dtype == _lib.GV_F32 else 2          # bytes per stored activation / filter element
        self.math_mode = math_mode
        self.ops = []            # dict records
        self.vbufs = []          # [size_elems, persistent]
        self.filters = []        # (weights_name, kh, kw, cin, cout, w_off)
        self.ss_specs = []       # (kind, name, c, eps, has_gamma, scale_off, shift_off)
        self._ss_cache = {}
        self.w_elems = 0
        self.ss_elems = 0
        self.end_points = {}
        self.input = TRef(-1, 0, nb, height, width, 3, 3)    # vbuf -1 = SLOT_INPUT
        self._plan = None
        self._phys = None
        self._bufs = None
        self._ptrs = None
        self.keepalive = []
        self.cur_lane = 0
        self.use_lanes = True
        # fp32 storage + GV_MATH_BF16X3: conv -> conv intermediates are kept as three bf16 planes so that the consumer's
        # LDS-DMA loader only moves bytes (csrc/conv_dma.hip); everything a pool, the grouping module or a caller reads
        # stays fp32
        self.use_p3 = dtype == _lib.GV_F32 and math_mode == _lib.GV_MATH_BF16X3
        self.p3_blocks = None             # None: every block; else the set of block / stem-layer names that use it
        # 16-bit storage: the pre-activation of a ResNet-v2 identity unit is applied by the loader of the unit's conv1
        # (gv_conv2d_fwd_xpre) instead of being stored by the unit before it: that conv3 is HBM-bound and writes 9 instead
        # of 13 channel-quanta per pixel.  Off: the second-output form everywhere (A/B switch; fp32 storage always)
        self.defer_preact = dtype != _lib.GV_F32
        # 16-bit storage: Conv2d_2b_3x3 -> MaxPool_3a_3x3 as one launch that writes only the pooled tensor
        # (GV_CONV_MAXPOOL3S2); off: the two launches (A/B switch)
        self.fuse_maxpool = dtype != _lib.GV_F32 or math_mode == _lib.GV_MATH_BF16X3
        # 16-bit storage, inference plans (make_plan turns it on): conv3 of a ResNet-v2 unit + the next unit's
        # pre-activation + conv1 as ONE launch (gv_bottleneck_chain_fwd) where the bottleneck depth is 64 or 128
        self.fuse_chain = False
        # ... and the unit's conv2 (3x3 / 1) in front of it: one launch per bottleneck unit (gv_bottleneck_unit_fwd)
        self.fuse_unit = False
        # ... and a depth-changing unit's conv1 + projection shortcut as one GEMM (resnet_unit1_pair)
        self.fuse_pair = False
        # ... and ResNet's first pre-activation on the fused conv1 -> pool1 launch's way out (GV_CONV_POOL_ACT2)
        self.fuse_pool_act = False
        # ... and a depth-changing unit's projection shortcut inside its conv3 GEMM (GV_CHAIN_PROJ: ResNet-v2-50's first unit)
        self.fuse_proj = False

    # ---- symbolic construction ----------------------------------------------------------------
    def lane(self, k):
        """Context manager: ops recorded inside go to launch lane k (gv_plan_set_schedule)."""
        plan = self

        class _Lane:
            def __enter__(self_inner):
                self_inner.prev = plan.cur_lane
                plan.cur_lane = k

            def __exit__(self_inner, *exc):
                plan.cur_lane = self_inner.prev
        return _Lane()

    def _record(self, op):
        op["lane"] = self.cur_lane
        self.ops.append(op)

    def new_tensor(self, nb, h, w, c, persistent=False, p3=False):
        p3 = bool(p3) and self.use_p3 and c % 16 == 0
        self.vbufs.append([nb * h * w * c, persistent, p3])
        return TRef(len(self.vbufs) - 1, 0, nb, h, w, c, c, p3)

    def keep(self, t):
        if t.vbuf >= 0:
            self.vbufs[t.vbuf][1] = True
        return t

    def _packed_elems(self, kh, kw, cin, cout):
        """Size of a packed filter in 4-byte units of the weights arena."""
        nbytes = int(self.lib.gv_packed_filter_bytes(kh, kw, cin, cout, self.dtype, self.math_mode))
        assert nbytes > 0 and nbytes % 4 == 0
        return nbytes // 4

    def _filter(self, name, kh, kw, cin, cout):
        n = self._packed_elems(kh, kw, cin, cout)
        off = self.w_elems
        self.filters.append((name, kh, kw, cin, cout, off))
        self.w_elems += (n + 63) // 64 * 64          # keep every filter 256-byte aligned
        return off

    def _scale_shift(self, kind, name, c, eps=0.0, has_gamma=False):
        key = (kind, name)
        if key in self._ss_cache:
            return self._ss_cache[key]
        cpad = (c + 3) // 4 * 4
        so, ho = self.ss_elems, self.ss_elems + cpad
        self.ss_elems += 2 * cpad
        self.ss_specs.append((kind, name, c, eps, has_gamma, so, ho))
        self._ss_cache[key] = (so, ho)
        return so, ho

    def conv_siblings(self, x, branches, first_out, norm, relu=True, pooled=None):
        """Several 1x1/stride-1 slim.conv2d that read the SAME input, as ONE implicit GEMM over their
        concatenated filters (GV_CONV_SPLIT): the first branch's columns land in `first_out` (a slice
        of the block's concat buffer), the others side by side in one scratch tensor whose channel
        slices are returned.  branches = [(scope, cout), ...] in reference order.

        pooled = (scope, cout, pool_name, out): the block's `avg_pool2d 3x3/1 SAME -> conv2d 1x1 -> BN -> ReLU`
        branch (nets/inception_v3.py:152-154,...) joins the GEMM as trailing columns WITHOUT ReLU, followed by
        one GV_POOL_AVG_RELU pass into `out`: relu(avgpool(BN(conv1x1(x)))) == relu(BN(conv1x1(avgpool(x))))
        because the 1x1 conv and the BN affine commute with an average whose weights sum to 1.  The pool then
        moves cout instead of cin channels (4-10x less HBM traffic) and one launch disappears."""
        branches = list(branches)
        n_relu = sum(c for _, c in branches)
        if pooled is not None:
            branches.append((pooled[0], pooled[1]))
        couts = [c for _, c in branches]
        total = sum(couts)
        rest = total - couts[0]
        assert (first_out.nb, first_out.h, first_out.w, first_out.c) == (x.nb, x.h, x.w, couts[0])
        n_each = [self._packed_elems(1, 1, x.c, c) for c in couts]
        w_off = self.w_elems
        off = w_off
        for (scope, c), n in zip(branches, n_each):
            self.filters.append((scope + "/weights", 1, 1, x.c, c, off))
            off += n
        self.w_elems = (off + 63) // 64 * 64
        cpad = (total + 3) // 4 * 4
        so, ho = self.ss_elems, self.ss_elems + cpad
        self.ss_elems += 2 * cpad
        cum = 0
        for scope, c in branches:
            self.ss_specs.append(("bn", scope + "/BatchNorm", c, norm[1], norm[2], so + cum, ho + cum))
            cum += c
        # the members other than the first feed 3x3 / 5x5 / 1x7 convs (and the commuted average pool): three-plane storage
        scratch = self.new_tensor(x.nb, x.h, x.w, rest, p3=all(c % 16 == 0 for c in couts[1:]) and couts[0] % 8 == 0)
        self._record(dict(kind="conv", name="+".join(sc for sc, _ in branches), x=x, y=first_out,
                             y2=scratch, res=None, w_off=w_off, scale_off=so, shift_off=ho,
                             scale2_off=0, shift2_off=0, kh=1, kw=1, stride=1, pad_t=0, pad_l=0,
                             relu=relu, split=couts[0], cout=total,
                             relu_cols=n_relu if pooled is not None else 0,
                             flops=2.0 * x.npix * total * x.c,
                             bytes=float(self.esz) * (x.npix * x.c + x.c * total + x.npix * total)))
        outs, lo = [], 0
        for c in couts[1:]:
            outs.append(scratch.channels(lo, lo + c))
            lo += c
        if pooled is not None:
            z = outs.pop()
            out = pooled[3]
            assert (out.nb, out.h, out.w, out.c) == (x.nb, x.h, x.w, pooled[1])
            self._record(dict(kind="pool", name=pooled[2], x=z, y=out, k=3, stride=1, pad_t=1, pad_l=1,
                              mode=_lib.GV_POOL_AVG_RELU | (_lib.GV_POOL_X_P3 if z.p3 else 0) |
                              (_lib.GV_POOL_Y_P3 if out.p3 else 0), flops=0.0,
                              bytes=float(self.esz) * 2 * z.npix * z.c))
        return outs

    def resnet_unit1_pair(self, x, scope1, d, norm1, scope_sc, depth):
        """conv1 (1x1, BatchNorm + ReLU, d columns: nets/resnet_v2.py:83-84) and the projection shortcut (1x1, biases,
        `depth` columns: :79-81) of a unit whose depth changes read the SAME pre-activation at stride 1 (the strided unit
        is a block's last): ONE GEMM over the concatenated filters — columns [0, d) -> conv1's tensor with ReLU, the rest ->
        the shortcut's without (GV_CONV_SPLIT, relu_cols) — so the pre-activation is read once.  Returns (conv1, shortcut)."""
        total = d + depth
        y1 = self.new_tensor(x.nb, x.h, x.w, d)
        y2 = self.new_tensor(x.nb, x.h, x.w, depth)
        n1, n2 = self._packed_elems(1, 1, x.c, d), self._packed_elems(1, 1, x.c, depth)
        w_off = self.w_elems
        self.filters.append((scope1 + "/weights", 1, 1, x.c, d, w_off))
        self.filters.append((scope_sc + "/weights", 1, 1, x.c, depth, w_off + n1))
        self.w_elems = (w_off + n1 + n2 + 63) // 64 * 64
        cpad = (total + 3) // 4 * 4
        so, ho = self.ss_elems, self.ss_elems + cpad
        self.ss_elems += 2 * cpad
        self.ss_specs.append(("bn", scope1 + "/BatchNorm", d, norm1[1], norm1[2], so, ho))
        self.ss_specs.append(("bias", scope_sc + "/biases", depth, 0.0, False, so + d, ho + d))
        self._record(dict(kind="conv", name=scope1 + "+" + scope_sc, x=x, y=y1, y2=y2, res=None, w_off=w_off,
                          scale_off=so, shift_off=ho, scale2_off=0, shift2_off=0, kh=1, kw=1, stride=1, pad_t=0, pad_l=0,
                          relu=True, split=d, cout=total, relu_cols=d, xpre=None, maxpool=None,
                          flops=2.0 * x.npix * total * x.c,
                          bytes=float(self.esz) * (x.npix * x.c + x.c * total + x.npix * total)))
        return y1, y2

    def conv(self, x, scope, cout, k, stride=1, padding="SAME", out=None, norm=None, relu=True,
             residual=None, next_preact=None, p3=False, defer=False, maxpool=False, pool_act=None):
        """slim.conv2d.  norm = ('bn', eps, has_gamma) -> BatchNorm under scope/BatchNorm, no bias;
        norm = None -> biases, no BN (normalizer_fn=None).  next_preact = (bn_scope, eps) adds the
        second output relu(bn(out)) and returns (out, preact); with defer=True no second output is written and
        `preact` is a DeferredPreact(out, scale/shift offsets) that a 1x1 unpadded conv accepts as its input.
        maxpool: the layer is followed by max_pool2d 3x3 / 2 VALID and nobody else reads it — ONE launch writes the
        pooled tensor (GV_CONV_MAXPOOL3S2; the caller checks `fused_maxpool_ok` first), which is what is returned."""
        xpre = None
        if isinstance(x, DeferredPreact):
            x, xpre = x.t, (x.scale_off, x.shift_off)
        kh, kw = (k, k) if isinstance(k, int) else k
        oh, pad_t = _out_size(x.h, kh, stride, padding if not isinstance(padding, tuple) else padding[0])
        ow, pad_l = _out_size(x.w, kw, stride, padding if not isinstance(padding, tuple) else padding[1])
        if maxpool:
            assert residual is None and next_preact is None and xpre is None and not p3
            maxpool = "VALID" if maxpool is True else maxpool
            if maxpool == "SAME":
                assert oh % 2 == 0 and ow % 2 == 0
                pshape = (x.nb, oh // 2, ow // 2, cout)
            else:
                pshape = (x.nb, (oh - 3) // 2 + 1, (ow - 3) // 2 + 1, cout)
            if out is None:
                out = self.new_tensor(*pshape)
            assert (out.nb, out.h, out.w, out.c) == pshape, (scope, out, pshape)     # (`out`: a channel slice of a wider buffer)
        elif out is None:
            # p3: the output only feeds other convolutions (a conv -> conv intermediate)
            out = self.new_tensor(x.nb, oh, ow, cout, p3=p3 and next_preact is None and residual is None and x.c >= 16)
        assert maxpool or (out.nb, out.h, out.w, out.c) == (x.nb, oh, ow, cout), (scope, out, oh, ow, cout)
        w_off = self._filter(scope + "/weights", kh, kw, x.c, cout)
        if norm is not None:
            so, ho = self._scale_shift("bn", scope + "/BatchNorm", cout, norm[1], norm[2])
        else:
            so, ho = self._scale_shift("bias", scope + "/biases", cout)
        y2 = None
        s2 = h2 = 0
        if pool_act is not None:                              # (bn_scope, eps): BatchNorm + ReLU of the POOLED tensor in the same launch
            assert maxpool and next_preact is None
            s2, h2 = self._scale_shift("bn", pool_act[0], cout, pool_act[1], True)
        if next_preact is not None:
            s2, h2 = self._scale_shift("bn", next_preact[0], cout, next_preact[1], True)
            if not defer:
                y2 = self.new_tensor(x.nb, oh, ow, cout)
        if xpre is not None:
            assert kh == kw == 1 and pad_t == 0 and pad_l == 0 and self.dtype != _lib.GV_F32, scope
        if residual is not None:
            assert (residual.nb, residual.h, residual.w, residual.c) == (x.nb, oh, ow, cout)
        self._record(dict(kind="conv", name=scope, x=x, y=out, y2=y2, res=residual, w_off=w_off,
                             scale_off=so, shift_off=ho, scale2_off=s2, shift2_off=h2,
                             kh=kh, kw=kw, stride=stride, pad_t=pad_t, pad_l=pad_l, relu=relu,
                             split=0, cout=cout, xpre=xpre, maxpool=maxpool or None, pool_act=pool_act is not None, oh=oh, ow=ow,
                             flops=2.0 * x.nb * oh * ow * cout * kh * kw * x.c,
                             bytes=float(self.esz) * (x.npix * x.c + kh * kw * x.c * cout +
                                                      (out.npix * cout if maxpool else x.nb * oh * ow * cout *
                                                       (1 + (residual is not None) + (y2 is not None))))))
        if next_preact is not None:
            return (out, DeferredPreact(out, s2, h2) if defer else y2)
        return out

    def chain_ok(self, x):
        """May conv3 over `x` (d channels -> 4d) and the next unit's preact + conv1 be one launch (csrc/conv_chain.hip)?"""
        return bool(self.fuse_chain) and self.dtype != _lib.GV_F32 and x.c in (64, 128) and x.ld % 8 == 0 and not x.p3

    def unit(self, x, scope2, norm2, scope3, shortcut, pre_scope, pre_eps, scope1, norm1):
        """chain() with the unit's conv2 (3x3 / 1 SAME + BatchNorm + ReLU, nets/resnet_v2.py:85-86) in front: x is the unit's
        conv1 output; conv2's output is never stored.  Recorded like a chain op, `chain["front"]` holds conv2's operands."""
        d, cout = x.c, 4 * x.c
        out, z = self.chain(x, scope3, shortcut, pre_scope, pre_eps, scope1, norm1)
        op = self.ops[-1]
        w2_off = self._filter(scope2 + "/weights", 3, 3, d, d)
        s2, h2 = self._scale_shift("bn", scope2 + "/BatchNorm", d, norm2[1], norm2[2])
        op["chain"]["front"] = dict(w_off=w2_off, scale_off=s2, shift_off=h2)
        op["name"] = scope2 + "+" + op["name"]
        op["flops"] += 2.0 * x.npix * 9 * d * d
        op["bytes"] += float(self.esz) * 9 * d * d
        return out, z

    def chain(self, x, scope3, shortcut, pre_scope, pre_eps, scope1, norm1, proj=None):
        """nets/resnet_v2.py:87-91 of one unit (conv3 1x1 + biases, `shortcut + residual`) and :75, :83-84 of the next
        (preact BatchNorm + ReLU, conv1 1x1 + BatchNorm + ReLU) as ONE launch.  Returns (unit output [.., 4d], next unit's
        conv1 output [.., d]).  Recorded as a conv op with a `chain` record: x, res, y as for conv3, y2 = the conv1 output.
        proj = the scope of the unit's PROJECTION shortcut (:79-81; the unit's depth changes, shortcut is None): x is
        [conv2 output | the unit's pre-activation] (2d channels), conv3's and the shortcut's filters are concatenated along
        cin and their biases added — conv3(x2) + shortcut(x0) is one accumulation and the shortcut tensor never exists
        (GV_CHAIN_PROJ)."""
        d = x.c // 2 if proj else x.c
        cout = 4 * d
        assert (shortcut is None) == bool(proj)
        assert proj or (shortcut.nb, shortcut.h, shortcut.w, shortcut.c) == (x.nb, x.h, x.w, cout)
        out = self.new_tensor(x.nb, x.h, x.w, cout)
        z = self.new_tensor(x.nb, x.h, x.w, d)
        if proj:
            w3_off = self._filter(scope3 + "/weights|" + proj + "/weights", 1, 1, 2 * d, cout)
            so, ho = self._scale_shift("bias_sum", scope3 + "/biases|" + proj + "/biases", cout)
        else:
            w3_off = self._filter(scope3 + "/weights", 1, 1, d, cout)
            so, ho = self._scale_shift("bias", scope3 + "/biases", cout)
        ps, ph = self._scale_shift("bn", pre_scope, cout, pre_eps, True)
        w1_off = self._filter(scope1 + "/weights", 1, 1, cout, d)
        s1, h1 = self._scale_shift("bn", scope1 + "/BatchNorm", d, norm1[1], norm1[2])
        self._record(dict(kind="conv", name=(proj + "+" if proj else "") + scope3 + "+" + scope1, x=x, y=out, y2=z, res=shortcut,
                          w_off=w3_off, scale_off=so, shift_off=ho, scale2_off=ps, shift2_off=ph,
                          chain=dict(w1_off=w1_off, scale1_off=s1, shift1_off=h1, proj=bool(proj)),
                          kh=1, kw=1, stride=1, pad_t=0, pad_l=0, relu=False, split=0, cout=cout, xpre=None, maxpool=None,
                          oh=x.h, ow=x.w, flops=2.0 * x.npix * (x.c * cout + cout * d),
                          bytes=float(self.esz) * (x.npix * (x.c + (0 if proj else cout) + cout + d) + (x.c + d) * cout)))
        return out, z

    def fused_maxpool_ok(self, x, cout, k, padding, stride=1, pool_padding="VALID"):
        """May `conv(x, ..., cout, k, stride, padding, maxpool=pool_padding)` be one launch?  The classes
        GV_CONV_MAXPOOL3S2 / _SAME serve (include/gvcnn_hip.h), 16-bit storage: 3x3 / stride 1 from 32 to 64 channels
        (VALID pool), or a 3x3 / 7x7 stride-2 stem on the fp32 images at 64 channels (VALID, or SAME on an even map)."""
        pads = padding if isinstance(padding, tuple) else (padding, padding)
        oh = _out_size(x.h, k, stride, pads[0])[0]
        ow = _out_size(x.w, k, stride, pads[1])[0]
        if not (self.fuse_maxpool and cout == 64 and not x.p3 and min(oh, ow) >= 3):
            return False
        if self.dtype == _lib.GV_F32:                         # three-plane math: the halo kernel's 30-pixel strip form, VALID pool
            return (self.math_mode == _lib.GV_MATH_BF16X3 and pool_padding == "VALID" and k == 3 and stride == 1 and x.c == 32
                    and x.vbuf >= 0 and x.ld % 4 == 0 and -(-ow // 16) * 16 * 5 > -(-ow // 30) * 32 * 4)
        if pool_padding == "SAME" and (oh % 2 or ow % 2):
            return False
        if x.vbuf < 0:                                        # the network input: the stem strip kernel
            return k in (3, 7) and stride == 2 and x.c == 3
        return pool_padding == "VALID" and k == 3 and stride == 1 and x.c == 32 and x.ld % 8 == 0

    def pool(self, x, k, stride, padding, mode, out=None, name="pool", p3=False):
        oh, pad_t = _out_size(x.h, k, stride, padding)
        ow, pad_l = _out_size(x.w, k, stride, padding)
        if out is None:
            out = self.new_tensor(x.nb, oh, ow, x.c, p3=p3)
        mode = mode | (_lib.GV_POOL_X_P3 if x.p3 else 0) | (_lib.GV_POOL_Y_P3 if out.p3 else 0)
        assert (out.nb, out.h, out.w, out.c) == (x.nb, oh, ow, x.c)
        self._record(dict(kind="pool", name=name, x=x, y=out, k=k, stride=stride, pad_t=pad_t,
                             pad_l=pad_l, mode=mode, flops=0.0,
                             bytes=float(self.esz) * (x.npix * x.c + out.npix * out.c)))
        return out

    def pooled_branch(self, x, conv_scope, pool_name, depth, dst, norm):
        """avg_pool2d 3x3/1 SAME -> conv2d 1x1 -> BN -> ReLU in the reference order (un-fused plans)."""
        t = self.pool(x, 3, 1, "SAME", _lib.GV_POOL_AVG, name=pool_name)
        return self.conv(t, conv_scope, depth, 1, out=dst, norm=norm, relu=True)

    def bn_relu(self, x, bn_scope, eps, name, out=None):
        """Stand-alone slim.batch_norm(activation_fn=relu) (resnet_v2.py:75, first unit only)."""
        if out is None:
            out = self.new_tensor(x.nb, x.h, x.w, x.c)
        assert (out.nb, out.h, out.w, out.c) == (x.nb, x.h, x.w, x.c)
        so, ho = self._scale_shift("bn", bn_scope, x.c, eps, True)
        self._record(dict(kind="ssa", name=name, x=x, y=out, scale_off=so, shift_off=ho, relu=True,
                             flops=0.0, bytes=2.0 * self.esz * x.npix * x.c))
        return out

    # ---- lowering ----------------------------------------------------------------------------
    def _assign_buffers(self):
        last_use = {}
        first_def = {}
        lanes_of = {}
        for i, op in enumerate(self.ops):
            for key in ("x", "y", "y2", "res"):
                t = op.get(key)
                if t is not None and t.vbuf >= 0:
                    last_use[t.vbuf] = i
                    first_def.setdefault(t.vbuf, i)
                    lanes_of.setdefault(t.vbuf, set()).add(op.get("lane", 0))
        phys_sizes = []
        phys_p3 = []               # a physical buffer holds either fp32 / 16-bit tensors or three-plane tensors
        self._phys_p3 = phys_p3
        free = []                  # (physical id, op index at which it was released, lanes that touched it)
        vmap = {}
        # A buffer released by one lane is not handed to ANOTHER lane for REUSE_DELAY ops: immediate
        # reuse would order the independent branches behind each other (a write-after-read hazard the
        # schedule then has to honour).  Memory is not the constraint here (288 GB of HBM).
        REUSE_DELAY = 16 if self.use_lanes else 0
        for i, op in enumerate(self.ops):
            lane = op.get("lane", 0)
            for key in ("y", "y2"):
                t = op.get(key)
                if t is None or t.vbuf in vmap:
                    continue
                need = self.vbufs[t.vbuf][0]
                fmt = self.vbufs[t.vbuf][2]
                ok = [f for f in free if (f[2] == {lane} or f[1] <= i - REUSE_DELAY) and phys_p3[f[0]] == fmt]
                cand = [f for f in ok if phys_sizes[f[0]] >= need]
                if cand:
                    f = min(cand, key=lambda q: phys_sizes[q[0]])
                    free.remove(f)
                    p = f[0]
                elif ok:                           # enlarge the biggest reusable buffer instead of adding one
                    f = max(ok, key=lambda q: phys_sizes[q[0]])
                    free.remove(f)
                    p = f[0]
                    phys_sizes[p] = need
                else:
                    p = len(phys_sizes)
                    phys_sizes.append(need)
                    phys_p3.append(fmt)
                vmap[t.vbuf] = p
            for v, lu in list(last_use.items()):
                if lu == i and v in vmap and not self.vbufs[v][1]:
                    free.append((vmap[v], i, lanes_of[v]))
                    del last_use[v]
        return vmap, phys_sizes

    def lower(self, device):
        """Create the native plan and the device buffers."""
        assert self._plan is None
        lib = self.lib
        vmap, phys_sizes = self._assign_buffers()
        self._phys = vmap
        plan = C.c_void_p()
        _lib.check(lib.gv_plan_create(C.byref(plan)), "gv_plan_create")
        self._plan = plan

        def ref(t):
            if t is None:
                return -1, 0
            if t.vbuf < 0:
                return SLOT_INPUT, t.off
            return SLOT_ACT0 + vmap[t.vbuf], t.off

        lowp = self.dtype != _lib.GV_F32
        wmul = 2 if lowp else 1                               # arena offsets are 4-byte units; plan offsets are elements
        for op in self.ops:
            x, y = op["x"], op["y"]
            xs, xo = ref(x)
            ys, yo = ref(y)
            if op["kind"] == "conv" and op.get("chain") and op["chain"].get("front"):
                ch, res, y2 = op["chain"], op["res"], op["y2"]
                fr = ch["front"]
                d = _lib.UnitDesc(x.nb, x.h, x.w, x.c, x.ld, res.ld, y.ld, y2.ld, self.dtype, _lib.GV_CONV_RELU2, 0)
                rs, ro = ref(res)
                y2s, y2o = ref(y2)
                _lib.check(lib.gv_plan_add_unit(plan, C.byref(d), xs, xo, SLOT_WEIGHTS, fr["w_off"] * wmul, op["w_off"] * wmul,
                                                ch["w1_off"] * wmul, SLOT_SS, fr["scale_off"], fr["shift_off"], op["scale_off"],
                                                op["shift_off"], op["scale2_off"], op["shift2_off"], ch["scale1_off"],
                                                ch["shift1_off"], rs, ro, ys, yo, y2s, y2o), "gv_plan_add_unit(%s)" % op["name"])
            elif op["kind"] == "conv" and op.get("chain"):
                ch, res, y2 = op["chain"], op["res"], op["y2"]
                if ch.get("proj"):                            # GV_CHAIN_PROJ: x = [conv2 output | pre-activation], no shortcut operand
                    d = _lib.ChainDesc(x.npix, x.c // 2, x.ld, 0, y.ld, y2.ld, self.dtype, _lib.GV_CONV_RELU2 | _lib.GV_CHAIN_PROJ, 0)
                    rs, ro = -1, 0
                else:
                    d = _lib.ChainDesc(x.npix, x.c, x.ld, res.ld, y.ld, y2.ld, self.dtype, _lib.GV_CONV_RELU2, 0)
                    rs, ro = ref(res)
                y2s, y2o = ref(y2)
                _lib.check(lib.gv_plan_add_chain(plan, C.byref(d), xs, xo, SLOT_WEIGHTS, op["w_off"] * wmul,
                                                 ch["w1_off"] * wmul, SLOT_SS, op["scale_off"], op["shift_off"],
                                                 op["scale2_off"], op["shift2_off"], ch["scale1_off"], ch["shift1_off"],
                                                 rs, ro, ys, yo, y2s, y2o), "gv_plan_add_chain(%s)" % op["name"])
            elif op["kind"] == "conv":
                y2, res = op["y2"], op["res"]
                split = op["split"]
                flags = _lib.GV_CONV_RELU if op["relu"] else 0
                if split:
                    flags |= _lib.GV_CONV_SPLIT
                elif y2 is not None:
                    flags |= _lib.GV_CONV_RELU2
                if lowp and x.vbuf < 0:
                    flags |= _lib.GV_CONV_X_F32               # the images stay fp32; the stem's loader rounds them
                if x.p3:
                    flags |= _lib.GV_CONV_X_P3
                if y.p3:
                    flags |= _lib.GV_CONV_Y_P3
                if y2 is not None and y2.p3:
                    assert split, "a three-plane second destination exists only for fused sibling convs"
                    flags |= _lib.GV_CONV_Y2_P3
                if op.get("maxpool"):                         # y is the pooled tensor; oh / ow stay the convolution's
                    flags |= _lib.GV_CONV_MAXPOOL3S2_SAME if op["maxpool"] == "SAME" else _lib.GV_CONV_MAXPOOL3S2
                    if op.get("pool_act"):                    # ... stored through a second BatchNorm + ReLU
                        flags |= _lib.GV_CONV_POOL_ACT2 | _lib.GV_CONV_RELU2
                d = _lib.ConvDesc(x.nb, x.h, x.w, x.c, x.ld, op["kh"], op["kw"], op["stride"],
                                  op["pad_t"], op["pad_l"], op.get("oh", y.h), op.get("ow", y.w), op["cout"], y.ld,
                                  res.ld if res is not None else 0, y2.ld if y2 is not None else 0,
                                  flags, self.dtype, split, op.get("tile", 0), self.math_mode, 0,
                                  op.get("relu_cols", 0))
                rs, ro = ref(res)
                y2s, y2o = ref(y2)
                _lib.check(lib.gv_plan_add_conv(plan, C.byref(d), xs, xo, SLOT_WEIGHTS, op["w_off"] * wmul,
                                                SLOT_SS, op["scale_off"], op["shift_off"], rs, ro,
                                                ys, yo, y2s, y2o, op["scale2_off"], op["shift2_off"]),
                           "gv_plan_add_conv(%s)" % op["name"])
                if op.get("xpre") is not None:
                    _lib.check(lib.gv_plan_set_conv_xpre(plan, lib.gv_plan_num_ops(plan) - 1, *op["xpre"]),
                               "gv_plan_set_conv_xpre(%s)" % op["name"])
            elif op["kind"] == "pool":
                d = _lib.PoolDesc(x.nb, x.h, x.w, x.c, x.ld, op["k"], op["k"], op["stride"],
                                  op["pad_t"], op["pad_l"], y.h, y.w, y.ld, op["mode"], self.dtype)
                _lib.check(lib.gv_plan_add_pool(plan, C.byref(d), xs, xo, ys, yo),
                           "gv_plan_add_pool(%s)" % op["name"])
            else:
                _lib.check(lib.gv_plan_add_scale_shift_act(plan, x.npix, x.c, x.ld, y.ld, 1, self.dtype,
                                                           xs, xo, SLOT_SS, op["scale_off"],
                                                           op["shift_off"], ys, yo),
                           "gv_plan_add_scale_shift_act(%s)" % op["name"])
        self._schedule(vmap)
        self.tdtype = TORCH_DTYPES[self.dtype]
        self.weights = torch.zeros(max(self.w_elems, 4), dtype=torch.float32, device=device)   # 4-byte units
        self.ss = torch.zeros(max(self.ss_elems, 4), dtype=torch.float32, device=device)
        self.act = [torch.empty(((n + 7) // 8 * 8) * 3, dtype=torch.int16, device=device) if p3_ else
                    torch.empty((n + 7) // 8 * 8, dtype=self.tdtype, device=device)
                    for n, p3_ in zip(phys_sizes, self._phys_p3)]
        self._bufs = [None, self.weights, self.ss] + self.act
        self.act_bytes = sum(n * (6 if p3_ else self.esz) for n, p3_ in zip(phys_sizes, self._phys_p3))
        return self

    def _schedule(self, vmap):
        """Cross-lane dependencies: an op waits for every earlier op whose access to the same physical
        buffer conflicts with its own (read-after-write, write-after-read, write-after-write);
        accesses to disjoint channel slices of one tensor do not conflict."""
        lanes = sorted({op["lane"] for op in self.ops})
        self.lanes_used = len(lanes)
        if not self.use_lanes or len(lanes) == 1:
            return
        history = {}                      # phys -> [(op, is_write, vbuf, lo, hi)]
        hb = []                           # hb[i]: ops known complete before op i starts
        last_on_lane = {}
        for i, op in enumerate(self.ops):
            acc = []
            for key, is_w in (("x", False), ("res", False), ("y", True), ("y2", True)):
                t = op.get(key)
                if t is not None and t.vbuf >= 0:
                    lo = t.off % t.ld
                    acc.append((vmap[t.vbuf], is_w, t.vbuf, lo, lo + t.c))
            need = set()
            for ph, is_w, vb, lo, hi in acc:
                for (j, jw, jvb, jlo, jhi) in history.get(ph, ()):
                    if not (is_w or jw):
                        continue
                    if jvb == vb and (hi <= jlo or jhi <= lo):
                        continue
                    need.add(j)
            prev = last_on_lane.get(op["lane"])
            known = set() if prev is None else (hb[prev] | {prev})
            deps = []
            for j in sorted(need, reverse=True):          # latest first: it usually implies the older ones
                if j in known:
                    continue
                assert self.ops[j]["lane"] != op["lane"]
                deps.append(j)
                known |= hb[j] | {j}
            hb.append(known)
            last_on_lane[op["lane"]] = i
            for ph, is_w, vb, lo, hi in acc:
                history.setdefault(ph, []).append((i, is_w, vb, lo, hi))
            op["deps"] = sorted(deps)
            arr = (C.c_int32 * max(len(deps), 1))(*sorted(deps))
            _lib.check(self.lib.gv_plan_set_schedule(self._plan, i, op["lane"], arr, len(deps)),
                       "gv_plan_set_schedule(%s)" % op["name"])

    def view(self, t):
        """torch view [nb,h,w,c] of a plan tensor (strided when it is a channel slice)."""
        base = self._bufs[SLOT_ACT0 + self._phys[t.vbuf]]
        if t.p3:                                          # three planes: an fp32 COPY (exact sum), not a view
            from . import p3 as _p3
            g = torch.as_strided(base, (t.nb, t.h, t.w, t.c // 16, 3, 16),
                                 (t.h * t.w * t.ld * 3, t.w * t.ld * 3, t.ld * 3, 48, 16, 1), t.off * 3)
            return _p3.from_p3(g)
        return torch.as_strided(base, (t.nb, t.h, t.w, t.c), (t.h * t.w * t.ld, t.w * t.ld, t.ld, 1),
                                t.off)

    # ---- parameters ----------------------------------------------------------------------------
    def bind(self, params, stream=None):
        """Upload slim-named parameters: pack HWIO filters on device, fold BN into scale/shift.
        params: dict name -> array-like (numpy / torch, any device), fp32."""
        dev = self.weights.device
        st = _stream_ptr(stream)
        for name, kh, kw, cin, cout, off in self.filters:
            # ("a|b": the two variables concatenated along cin — GV_CHAIN_PROJ's [conv3 ; shortcut] filter)
            w = torch.cat([torch.as_tensor(params[n]).to(device=dev, dtype=torch.float32) for n in name.split("|")], dim=2).contiguous()
            assert tuple(w.shape) == (kh, kw, cin, cout), (name, tuple(w.shape), (kh, kw, cin, cout))
            _lib.check(self.lib.gv_pack_filter_hwio(w.data_ptr(), kh, kw, cin, cout,
                                                    self.weights.data_ptr() + 4 * off, self.dtype,
                                                    self.math_mode, st),
                       "gv_pack_filter_hwio(%s)" % name)
            self.keepalive.append(w)
        host = np.zeros(max(self.ss_elems, 4), dtype=np.float32)

        def arr(n):
            return np.asarray(torch.as_tensor(params[n]).detach().cpu(), dtype=np.float64)

        for kind, name, c, eps, has_gamma, so, ho in self.ss_specs:
            if kind == "bias":
                host[so:so + c] = 1.0
                host[ho:ho + c] = arr(name)
            elif kind == "bias_sum":                          # ("a|b": the biases of two convolutions whose outputs are added)
                host[so:so + c] = 1.0
                host[ho:ho + c] = sum(arr(n) for n in name.split("|"))
            else:
                inv = 1.0 / np.sqrt(arr(name + "/moving_variance") + eps)
                if has_gamma:
                    inv = inv * arr(name + "/gamma")
                host[so:so + c] = inv
                host[ho:ho + c] = arr(name + "/beta") - arr(name + "/moving_mean") * inv
        self.ss.copy_(torch.from_numpy(host))
        torch.cuda.synchronize(dev)
        self.keepalive.clear()
        return self

    def param_shapes(self):
        """slim variable name -> shape, for every variable the plan reads."""
        shapes = {}
        for name, kh, kw, cin, cout, _ in self.filters:
            for n in name.split("|"):
                shapes[n] = (kh, kw, cin // len(name.split("|")), cout)
        for kind, name, c, eps, has_gamma, _, _ in self.ss_specs:
            if kind in ("bias", "bias_sum"):
                for n in name.split("|"):
                    shapes[n] = (c,)
            else:
                for leaf in ("beta", "moving_mean", "moving_variance") + (("gamma",) if has_gamma else ()):
                    shapes[name + "/" + leaf] = (c,)
        return shapes

    # ---- execution ----------------------------------------------------------------------------
    def _ptr_table(self, x):
        assert x.is_contiguous() and x.dtype == torch.float32
        assert x.numel() == self.nb * self.height * self.width * 3, (tuple(x.shape), self.nb)
        if self._ptrs is None:
            self._ptrs = (C.c_void_p * len(self._bufs))()
            for i, b in enumerate(self._bufs):
                self._ptrs[i] = b.data_ptr() if b is not None else 0
        self._ptrs[SLOT_INPUT] = x.data_ptr()
        return self._ptrs

    def run(self, x, stream=None):
        """Enqueue the whole backbone on `stream` (default: torch's current stream)."""
        ptrs = self._ptr_table(x)
        _lib.check(self.lib.gv_plan_run(self._plan, ptrs, len(self._bufs), _stream_ptr(stream)),
                   "gv_plan_run")

    def run_range(self, x, first, count, stream=None):
        ptrs = self._ptr_table(x)
        _lib.check(self.lib.gv_plan_run_range(self._plan, first, count, ptrs, len(self._bufs),
                                              _stream_ptr(stream)), "gv_plan_run_range")

    def declined_fused_pools(self):
        """Names of the fused ops (conv -> max-pool, bottleneck chain) the library declines on this device
        (GV_E_UNSUPPORTED); any other error raises.  One launch per such op on the plan's own (unbound) buffers: the values are irrelevant."""
        fused = [i for i, op in enumerate(self.ops) if op["kind"] == "conv" and (op.get("maxpool") or op.get("chain"))]
        if not fused:
            return []
        x = torch.zeros(self.nb * self.height * self.width * 3, dtype=torch.float32, device=self.weights.device)
        declined = []
        for i in fused:
            rc = self.lib.gv_plan_run_range(self._plan, i, 1, self._ptr_table(x), len(self._bufs), _stream_ptr(None))
            if rc == _lib.GV_E_UNSUPPORTED:
                declined.append(self.ops[i]["name"])
            else:
                _lib.check(rc, "gv_plan_run_range(%s)" % self.ops[i]["name"])
        torch.cuda.synchronize(self.weights.device)
        self._ptrs[SLOT_INPUT] = 0
        return declined

    def apply_tiles(self, table):
        """Install a previously measured {op name: tile configuration} table (no launches)."""
        for i, op in enumerate(self.ops):
            if op["kind"] == "conv" and op["name"] in table and not op.get("maxpool") and not op.get("chain"):   # (one kernel serves those forms)
                op["tile"] = int(table[op["name"]]) + 1
                _lib.check(self.lib.gv_plan_set_conv_tile(self._plan, i, op["tile"]), "gv_plan_set_conv_tile")

    def autotune(self, x, iters=3, verbose=False, in_sequence=4):
        """Pick, per conv launch, the fastest tile configuration by timing each on this device with
        the plan's own buffers (hipEvents on the launch stream).  A speed choice: tiles of ONE kernel family sum k in
        the same order (bitwise the same result); tiles of different families (register-staged, LDS-DMA chunk-major,
        wave-specialised) sum k in another order — fp32-rounding-level differences, <= 6e-6 of a tensor's largest value
        (include/gvcnn_hip.h, tile_cfg).  Timing noise decides between near-equal tiles, so two processes that each
        tune are NOT bit-reproducible against each other: bit reproducibility across runs needs a persisted table
        (apply_tiles / bench.py --tile-cache); inside one process the installed table is fixed and runs repeat bitwise.

        Two passes.  (1) Every configuration of every launch as a WARM REPEAT of itself (gv_plan_time): cheap, but a
        repeated launch finds its input and its filter in the XCD's L2, which the launch inside the network does not —
        there the input was just written by another kernel and comes from the Infinity Cache or HBM, and a tile whose
        short ring is the fastest on L2 hits (two stages, three workgroups per CU) can lose to a deeper one.  Isolated
        repeats of the Mixed_6 layers run at 600 - 830 TFLOP/s, the same launches in sequence at 400 - 550
        (profiles/r3_conv_probe_bf16.txt against r3_step_times_*).  So (2) the `in_sequence` best configurations of pass 1
        are timed IN SEQUENCE (gv_plan_time_each: an event pair behind every op over whole passes of the plan; round r
        gives every launch its r-th candidate) and each launch keeps the candidate that was fastest where it actually
        runs.  in_sequence <= 1: pass 1 only (round 3's behaviour)."""
        lib = self.lib
        ncfg_plan = lib.gv_conv2d_num_tile_cfgs(self.math_mode if self.dtype == _lib.GV_F32 else -1)   # -1: 16-bit storage
        ncfg_p3 = lib.gv_conv2d_num_tile_cfgs(-3)             # three-plane input: the LDS-DMA kernel's own table
        self.run(x)
        chosen, cands = {}, {}
        try:
            for i, op in enumerate(self.ops):
                if op["kind"] != "conv" or op.get("chain"):
                    continue
                ncfg = ncfg_p3 if op["x"].p3 else ncfg_plan
                timed = []
                for t in range(ncfg):
                    lib.gv_conv2d_set_tile_override(t)
                    try:
                        timed.append((self.time_range(x, i, 1, iters), t))
                    except _lib.GvError:
                        continue
                # a second, longer look at the three fastest: one noisy sample must not pick the tile of a 0.1 ms launch
                finals = []
                for _, t in sorted(timed)[:3]:
                    lib.gv_conv2d_set_tile_override(t)
                    finals.append((min(self.time_range(x, i, 1, 2 * iters), self.time_range(x, i, 1, 2 * iters)), t))
                best_ms, best = min(finals) if finals else (float("inf"), 0)
                op["tile"] = best + 1
                chosen[op["name"]] = (best, best_ms)
                order = [best] + [t for _, t in sorted(timed) if t != best]
                cands[i] = order[:max(int(in_sequence), 1)]
                _lib.check(lib.gv_plan_set_conv_tile(self._plan, i, best + 1), "gv_plan_set_conv_tile")
                if verbose:
                    print("autotune %-60s cfg %2d  %.4f ms" % (op["name"][-60:], best, best_ms))
        finally:
            lib.gv_conv2d_set_tile_override(-1)
        if int(in_sequence) > 1 and cands:
            seq = {i: {} for i in cands}                          # op -> {tile: in-sequence ms}
            for r in range(int(in_sequence)):
                for i, c in cands.items():
                    _lib.check(lib.gv_plan_set_conv_tile(self._plan, i, c[min(r, len(c) - 1)] + 1), "gv_plan_set_conv_tile")
                each = [min(a, b) for a, b in zip(self.time_each(x, 2 * iters), self.time_each(x, 2 * iters))]
                for i, c in cands.items():
                    t = c[min(r, len(c) - 1)]
                    seq[i][t] = min(each[i], seq[i].get(t, float("inf")))
            moved = 0
            for i, c in cands.items():
                op = self.ops[i]
                best = min(c, key=lambda t: (seq[i][t], c.index(t)))
                moved += best != c[0]
                op["tile"] = best + 1
                chosen[op["name"]] = (best, seq[i][best])
                _lib.check(lib.gv_plan_set_conv_tile(self._plan, i, best + 1), "gv_plan_set_conv_tile")
                if verbose:
                    print("in sequence %-57s cfg %2d  %.4f ms  (warm-repeat choice cfg %2d: %.4f ms in sequence)"
                          % (op["name"][-57:], best, seq[i][best], c[0], seq[i][c[0]]))
            self.autotune_moved = moved                           # launches whose in-sequence winner is not the warm-repeat one
        return chosen

    def time_range(self, x, first, count, iters, stream=None):
        """Average ms of ops [first, first+count) via hipEvents on the launch stream."""
        ptrs = self._ptr_table(x)
        ms = C.c_float(0)
        _lib.check(self.lib.gv_plan_time(self._plan, first, count, ptrs, len(self._bufs), iters,
                                         C.byref(ms), _stream_ptr(stream)), "gv_plan_time")
        return ms.value

    def time_each(self, x, iters, stream=None):
        """[ms per op] with every op timed IN SEQUENCE over `iters` whole passes (single launch lane)."""
        ptrs = self._ptr_table(x)
        out = (C.c_float * len(self.ops))()
        _lib.check(self.lib.gv_plan_time_each(self._plan, ptrs, len(self._bufs), iters, out, _stream_ptr(stream)),
                   "gv_plan_time_each")
        return list(out)

    def __del__(self):
        try:
            if self._plan is not None:
                self.lib.gv_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    @property
    def total_flops(self):
        return sum(op["flops"] for op in self.ops)


def _stream_ptr(stream):
    if stream is None:
        return torch.cuda.current_stream().cuda_stream
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


# ------------------------------------------------------------------------------------------------
# Inception-v3 base — nets/inception_v3.py:93-410 under inception_arg_scope (inception_utils.py:52-78):
# every conv = conv(no bias) -> BN(no gamma, eps 1e-3) -> ReLU.
# ------------------------------------------------------------------------------------------------
def build_inception_v3(b, final_endpoint="Mixed_7c", keep=("Mixed_6e", "Mixed_7c"), scope="InceptionV3",
                       fuse_siblings=True):
    if final_endpoint not in INCEPTION_ENDPOINTS:
        raise ValueError("Unknown final endpoint %s" % final_endpoint)       # inception_v3.py:410
    BN = ("bn", INCEPTION_BN_EPS, False)
    MAX, AVG = _lib.GV_POOL_MAX, _lib.GV_POOL_AVG

    def conv(x, name, cout, k, stride=1, padding="SAME", out=None, mid=False):
        """mid: the output only feeds further convolutions of the same branch -> three-plane storage where the plan
        uses it (an end point somebody taps stays fp32; `b.p3_blocks` names the blocks that use it: a speed choice)."""
        blk = name.split("/")[0]
        if mid and b.use_p3 and name not in keep and (b.p3_blocks is None or blk in b.p3_blocks):
            return b.conv(x, scope + "/" + name, cout, k, stride, padding, out=out, norm=BN, relu=True, p3=True)
        return b.conv(x, scope + "/" + name, cout, k, stride, padding, out=out, norm=BN, relu=True)

    def siblings(x, branches, first_out, pooled=None):
        """The 1x1 convs of one block that read the block input (e.g. inception_v3.py:140,142,146), and the
        block's pooled branch: pooled = (block scope, depth, destination slice)."""
        if not fuse_siblings:
            outs = []
            for i, (name, c) in enumerate(branches):
                t = conv(x, name, c, 1, out=first_out if i == 0 else None, mid=i > 0)
                if i:
                    outs.append(t)
            if pooled is not None:
                s, depth, dst = pooled
                with b.lane(2):
                    b.pooled_branch(x, scope + "/" + s + "Branch_3/Conv2d_0b_1x1", s + "Branch_3/AvgPool_0a_3x3",
                                    depth, dst, BN)
            return outs
        pb = None
        if pooled is not None:
            s, depth, dst = pooled
            pb = (scope + "/" + s + "Branch_3/Conv2d_0b_1x1", depth, s + "Branch_3/AvgPool_0a_3x3", dst)
        blk = branches[0][0].split("/")[0]
        saved = b.use_p3
        b.use_p3 = saved and (b.p3_blocks is None or blk in b.p3_blocks)
        try:
            return b.conv_siblings(x, [(scope + "/" + n, c) for n, c in branches], first_out, BN, pooled=pb)
        finally:
            b.use_p3 = saved

    def cat_p3(name):
        """A block output nobody outside the backbone reads: three-plane storage, so that the next block's sibling GEMM
        (and Mixed_6a / 7a's strided 3x3) also run on the LDS-DMA kernel.  Tapped end points stay fp32."""
        return b.use_p3 and name not in keep and (b.p3_blocks is None or "concat" in b.p3_blocks)

    def done(name, t):
        b.end_points[name] = t
        if name in keep:
            b.keep(t)
        return name == final_endpoint

    net = conv(b.input, "Conv2d_1a_3x3", 32, 3, 2, "VALID")
    if done("Conv2d_1a_3x3", net): return net
    net = conv(net, "Conv2d_2a_3x3", 32, 3, 1, "VALID", mid=final_endpoint != "Conv2d_2a_3x3")
    if done("Conv2d_2a_3x3", net): return net
    if (final_endpoint != "Conv2d_2b_3x3" and "Conv2d_2b_3x3" not in keep and b.fused_maxpool_ok(net, 64, 3, "SAME")):
        # inception_v3.py:111-113 as ONE launch: the un-pooled Conv2d_2b_3x3 is never written (nobody taps it)
        net = b.conv(net, scope + "/Conv2d_2b_3x3", 64, 3, 1, "SAME", norm=BN, relu=True, maxpool=True)
    else:
        net = conv(net, "Conv2d_2b_3x3", 64, 3, 1, "SAME")
        if done("Conv2d_2b_3x3", net): return net
        net = b.pool(net, 3, 2, "VALID", MAX, name="MaxPool_3a_3x3")
    if done("MaxPool_3a_3x3", net): return net
    net = conv(net, "Conv2d_3b_1x1", 80, 1, 1, "VALID", mid=final_endpoint != "Conv2d_3b_1x1")
    if done("Conv2d_3b_1x1", net): return net
    net = conv(net, "Conv2d_4a_3x3", 192, 3, 1, "VALID", mid=cat_p3("Conv2d_4a_3x3") and final_endpoint != "Conv2d_4a_3x3")
    if done("Conv2d_4a_3x3", net): return net
    net = b.pool(net, 3, 2, "VALID", MAX, name="MaxPool_5a_3x3", **({"p3": True} if cat_p3("MaxPool_5a_3x3") else {}))
    if done("MaxPool_5a_3x3", net): return net

    def mixed5(x, name, b1a, b1b, pool_depth):                  # inception_v3.py:137-204
        s = name + "/"
        out = b.new_tensor(x.nb, x.h, x.w, 64 + 64 + 96 + pool_depth, p3=cat_p3(name))
        t1, t2 = siblings(x, [(s + "Branch_0/Conv2d_0a_1x1", 64), (s + "Branch_1/" + b1a, 48),
                              (s + "Branch_2/Conv2d_0a_1x1", 64)], out.channels(0, 64),
                          pooled=(s, pool_depth, out.channels(224, 224 + pool_depth)))
        with b.lane(1):
            conv(t1, s + "Branch_1/" + b1b, 64, 5, out=out.channels(64, 128))
        t = conv(t2, s + "Branch_2/Conv2d_0b_3x3", 96, 3, mid=True)
        conv(t, s + "Branch_2/Conv2d_0c_3x3", 96, 3, out=out.channels(128, 224))
        return out

    net = mixed5(net, "Mixed_5b", "Conv2d_0a_1x1", "Conv2d_0b_5x5", 32)
    if done("Mixed_5b", net): return net
    net = mixed5(net, "Mixed_5c", "Conv2d_0b_1x1", "Conv_1_0c_5x5", 64)
    if done("Mixed_5c", net): return net
    net = mixed5(net, "Mixed_5d", "Conv2d_0a_1x1", "Conv2d_0b_5x5", 64)
    if done("Mixed_5d", net): return net

    # Mixed_6a — inception_v3.py:207-223
    s = "Mixed_6a/"
    oh = (net.h - 3) // 2 + 1
    ow = (net.w - 3) // 2 + 1
    out = b.new_tensor(net.nb, oh, ow, 384 + 96 + net.c, p3=cat_p3("Mixed_6a"))
    conv(net, s + "Branch_0/Conv2d_1a_1x1", 384, 3, 2, "VALID", out=out.channels(0, 384))
    with b.lane(1):
        t = conv(net, s + "Branch_1/Conv2d_0a_1x1", 64, 1, mid=True)
        t = conv(t, s + "Branch_1/Conv2d_0b_3x3", 96, 3, mid=True)
        conv(t, s + "Branch_1/Conv2d_1a_1x1", 96, 3, 2, "VALID", out=out.channels(384, 480))
    with b.lane(2):
        b.pool(net, 3, 2, "VALID", MAX, out=out.channels(480, 480 + net.c), name=s + "Branch_2/MaxPool_1a_3x3")
    net = out
    if done("Mixed_6a", net): return net

    def mixed6(x, name, d):                                     # inception_v3.py:226-338
        s = name + "/"
        out = b.new_tensor(x.nb, x.h, x.w, 768, p3=cat_p3(name))
        t1, t2 = siblings(x, [(s + "Branch_0/Conv2d_0a_1x1", 192), (s + "Branch_1/Conv2d_0a_1x1", d),
                              (s + "Branch_2/Conv2d_0a_1x1", d)], out.channels(0, 192),
                          pooled=(s, 192, out.channels(576, 768)))
        with b.lane(1):
            t = conv(t1, s + "Branch_1/Conv2d_0b_1x7", d, (1, 7), mid=True)
            conv(t, s + "Branch_1/Conv2d_0c_7x1", 192, (7, 1), out=out.channels(192, 384))
        t = conv(t2, s + "Branch_2/Conv2d_0b_7x1", d, (7, 1), mid=True)
        t = conv(t, s + "Branch_2/Conv2d_0c_1x7", d, (1, 7), mid=True)
        t = conv(t, s + "Branch_2/Conv2d_0d_7x1", d, (7, 1), mid=True)
        conv(t, s + "Branch_2/Conv2d_0e_1x7", 192, (1, 7), out=out.channels(384, 576))
        return out

    for name, d in (("Mixed_6b", 128), ("Mixed_6c", 160), ("Mixed_6d", 160), ("Mixed_6e", 192)):
        net = mixed6(net, name, d)
        if done(name, net): return net

    # Mixed_7a — inception_v3.py:341-360
    s = "Mixed_7a/"
    oh = (net.h - 3) // 2 + 1
    ow = (net.w - 3) // 2 + 1
    out = b.new_tensor(net.nb, oh, ow, 320 + 192 + net.c, p3=cat_p3("Mixed_7a"))
    t = conv(net, s + "Branch_0/Conv2d_0a_1x1", 192, 1, mid=True)
    conv(t, s + "Branch_0/Conv2d_1a_3x3", 320, 3, 2, "VALID", out=out.channels(0, 320))
    with b.lane(1):
        t = conv(net, s + "Branch_1/Conv2d_0a_1x1", 192, 1, mid=True)
        t = conv(t, s + "Branch_1/Conv2d_0b_1x7", 192, (1, 7), mid=True)
        t = conv(t, s + "Branch_1/Conv2d_0c_7x1", 192, (7, 1), mid=True)
        conv(t, s + "Branch_1/Conv2d_1a_3x3", 192, 3, 2, "VALID", out=out.channels(320, 512))
    with b.lane(2):
        b.pool(net, 3, 2, "VALID", MAX, out=out.channels(512, 512 + net.c), name=s + "Branch_2/MaxPool_1a_3x3")
    net = out
    if done("Mixed_7a", net): return net

    def mixed7(x, name, b1_3x1, b2_names):                      # inception_v3.py:362-409
        s = name + "/"
        out = b.new_tensor(x.nb, x.h, x.w, 2048, p3=cat_p3(name))
        t1, t2 = siblings(x, [(s + "Branch_0/Conv2d_0a_1x1", 320), (s + "Branch_1/Conv2d_0a_1x1", 384),
                              (s + "Branch_2/Conv2d_0a_1x1", 448)], out.channels(0, 320),
                          pooled=(s, 192, out.channels(1856, 2048)))
        with b.lane(1):
            conv(t1, s + "Branch_1/Conv2d_0b_1x3", 384, (1, 3), out=out.channels(320, 704))
            conv(t1, s + "Branch_1/" + b1_3x1, 384, (3, 1), out=out.channels(704, 1088))
        t = conv(t2, s + "Branch_2/Conv2d_0b_3x3", 384, 3, mid=True)
        conv(t, s + "Branch_2/" + b2_names[0], 384, (1, 3), out=out.channels(1088, 1472))
        conv(t, s + "Branch_2/" + b2_names[1], 384, (3, 1), out=out.channels(1472, 1856))
        return out

    net = mixed7(net, "Mixed_7b", "Conv2d_0b_3x1", ("Conv2d_0c_1x3", "Conv2d_0d_3x1"))
    if done("Mixed_7b", net): return net
    net = mixed7(net, "Mixed_7c", "Conv2d_0c_3x1", ("Conv2d_0c_1x3", "Conv2d_0d_3x1"))
    done("Mixed_7c", net)
    return net


# ------------------------------------------------------------------------------------------------
# ResNet-v2-50 up to block4 — nets/resnet_v2.py:163-189,230-248; postnorm/pool5/logits are never
# fetched by nets/model.py:144-149 and are not built.
# ------------------------------------------------------------------------------------------------
RESNET50_BLOCKS = (("block1", 64, 3, 2), ("block2", 128, 4, 2), ("block3", 256, 6, 2),
                   ("block4", 512, 3, 1))


def build_resnet_v2_50(b, keep=("resnet_v2_50/block3", "resnet_v2_50/block4"), scope="resnet_v2_50"):
    BN = ("bn", RESNET_BN_EPS, True)
    MAX = _lib.GV_POOL_MAX
    # The first unit changes the depth at stride 1 and its pre-activation has as many channels as its bottleneck (64): its
    # projection shortcut rides in conv3's GEMM (GV_CHAIN_PROJ, csrc/conv_chain.hip) — the pre-activation is written next to
    # where conv2 will write, [conv2 | preact] is conv3's 128-channel input, and the 256-channel shortcut tensor never exists
    b0 = RESNET50_BLOCKS[0]
    proj0 = (getattr(b, "fuse_proj", False) and getattr(b, "fuse_chain", False) and b.dtype != _lib.GV_F32 and b0[1] == 64 and
             b0[2] >= 2 and scope + "/pool1" not in keep and "%s/%s/unit_1/bottleneck_v2" % (scope, b0[0]) not in keep)
    xb = None

    def first_preact_out(h, w):
        nonlocal xb
        if not proj0:
            return None
        xb = b.new_tensor(b.nb, h, w, 128)
        return xb.channels(64, 128)
    # conv1: explicit pad 3 + VALID, bias, no BN/ReLU (resnet_v2.py:178-180, resnet_utils.py:94-105)
    if scope + "/conv1" not in keep and getattr(b, "fuse_maxpool", False) and \
            b.fused_maxpool_ok(b.input, 64, 7, ((3, 3), (3, 3)), 2, "SAME"):
        # conv1 -> pool1 (resnet_v2.py:178-181) as ONE launch: the un-pooled conv1 is never written (nobody taps it).  The first
        # unit changes the depth, so pool1's only reader is that unit's `preact` BatchNorm + ReLU (:75): it rides on the
        # pooled tensor's way out of the same launch (16-bit storage, GV_CONV_POOL_ACT2) and pool1 itself is never stored
        first_pre = "%s/%s/unit_1/bottleneck_v2/preact" % (scope, RESNET50_BLOCKS[0][0])
        fold_pre = (b.dtype != _lib.GV_F32 and getattr(b, "fuse_pool_act", False) and RESNET50_BLOCKS[0][1] * 4 != 64 and
                    scope + "/pool1" not in keep)
        ph, pw = _out_size(b.height, 7, 2, (3, 3))[0] // 2, _out_size(b.width, 7, 2, (3, 3))[0] // 2
        net = b.conv(b.input, scope + "/conv1", 64, 7, 2, ((3, 3), (3, 3)), norm=None, relu=False, maxpool="SAME",
                     pool_act=(first_pre, RESNET_BN_EPS) if fold_pre else None, out=first_preact_out(ph, pw) if fold_pre else None)
        pre_folded = fold_pre
    else:
        net = b.conv(b.input, scope + "/conv1", 64, 7, 2, ((3, 3), (3, 3)), norm=None, relu=False)
        b.end_points[scope + "/conv1"] = net
        net = b.pool(net, 3, 2, "SAME", MAX, name=scope + "/pool1")                    # resnet_v2.py:181
        pre_folded = False
    units = []
    for bname, base, n_units, bstride in RESNET50_BLOCKS:
        for u in range(n_units):
            units.append((bname, base, u, n_units, bstride if u == n_units - 1 else 1))
    first_sc = "%s/%s/unit_1/bottleneck_v2" % (scope, units[0][0])
    # (pre_folded: `net` already IS the first pre-activation; nothing reads the raw pool1 — the first unit's shortcut is a
    # projection of the pre-activation, resnet_v2.py:79-81)
    if not pre_folded:
        o = first_preact_out(net.h, net.w)                 # (a training plan's bn_relu has no `out`: it never asks for one)
        preact = b.bn_relu(net, first_sc + "/preact", RESNET_BN_EPS, first_sc + "/preact", **({"out": o} if o is not None else {}))
    else:
        preact = net
    c1_ready = None                                     # this unit's conv1 output, when the previous unit's chain launch made it
    for i, (bname, base, u, n_units, stride) in enumerate(units):
        sc = "%s/%s/unit_%d/bottleneck_v2" % (scope, bname, u + 1)
        depth, depth_in = base * 4, net.c
        pair = None
        proj = i == 0 and xb is not None and stride == 1 and depth != depth_in and depth_in == base and \
            units[1][1] * 4 == depth and preact.vbuf == xb.vbuf
        if proj:
            shortcut = None                                                        # (inside conv3's GEMM)
        elif depth == depth_in:                                                    # resnet_v2.py:76-77
            shortcut = net if stride == 1 else b.pool(net, 1, stride, "VALID", MAX, name=sc + "/shortcut")
        elif (stride == 1 and getattr(b, "fuse_pair", False) and not isinstance(preact, DeferredPreact) and base % 8 == 0
              and c1_ready is None and base >= 128):
            # (block1's pair — 64 + 256 columns over K = 64 — measured slower than its two launches: 0.283 against 0.269 ms)
            # the projection shortcut and conv1 read the same pre-activation: one GEMM (16-bit inference plans)
            pair = b.resnet_unit1_pair(preact, sc + "/conv1", base, BN, sc + "/shortcut", depth)
            shortcut = pair[1]
        else:                                                                      # resnet_v2.py:79-81
            shortcut = b.conv(preact, sc + "/shortcut", depth, 1, stride, "VALID", norm=None, relu=False)
        if c1_ready is not None:
            r, c1_ready = c1_ready, None
        elif pair is not None:
            r = pair[0]
        else:
            r = b.conv(preact, sc + "/conv1", base, 1, 1, "SAME", norm=BN, relu=True)  # :83-84
        pad = "SAME" if stride == 1 else ((1, 1), (1, 1))                          # resnet_utils.py:94-105
        # conv3 + the next unit's preact + conv1 as ONE launch (csrc/conv_chain.hip): the next unit keeps this depth
        # (identity shortcut, conv1 the pre-activation's only reader) and the bottleneck depth is one the kernel serves;
        # with fuse_unit this unit's conv2 (stride 1 here: the strided unit is a block's last) runs in front of it
        chains = i + 1 < len(units) and units[i + 1][1] * 4 == depth and getattr(b, "fuse_chain", False) and b.chain_ok(r)
        assert chains or not proj
        # (whole units where that pays: d = 64.  At d = 128 the unit launch measured level with chain + conv2 — 0.276 against
        # 0.166 + 0.107 ms, profiles/r6_seq_vs_warm_c4_*.txt — its conv2 phase is matrix-heavy and its one 8-wave workgroup per
        # CU runs the phases in lockstep; fuse_unit="all" asks for it anyway)
        fu = getattr(b, "fuse_unit", False)
        whole = chains and bool(fu) and stride == 1 and (r.c <= 64 or fu == "all") and not proj
        if not whole:
            r = b.conv(r, sc + "/conv2", base, 3, stride, pad, norm=BN, relu=True,      # :85-86
                       out=xb.channels(0, 64) if proj else None)
        nxt = None
        kw = {}
        if i + 1 < len(units):
            nb_, nbase, nu, _, _ = units[i + 1]
            nxt = ("%s/%s/unit_%d/bottleneck_v2/preact" % (scope, nb_, nu + 1), RESNET_BN_EPS)
            if chains:
                nsc = "%s/%s/unit_%d/bottleneck_v2" % (scope, nb_, nu + 1)
                if whole:
                    net, c1_ready = b.unit(r, sc + "/conv2", BN, sc + "/conv3", shortcut, nxt[0], nxt[1], nsc + "/conv1", BN)
                elif proj:
                    net, c1_ready = b.chain(xb, sc + "/conv3", None, nxt[0], nxt[1], nsc + "/conv1", BN, proj=sc + "/shortcut")
                else:
                    net, c1_ready = b.chain(r, sc + "/conv3", shortcut, nxt[0], nxt[1], nsc + "/conv1", BN)
                preact = None
                b.end_points[sc] = net
                if u == n_units - 1:
                    b.end_points["%s/%s" % (scope, bname)] = net
                    if "%s/%s" % (scope, bname) in keep:
                        b.keep(net)
                continue
            # the next unit keeps this depth (identity shortcut): its conv1 is the pre-activation's only reader.  Not in
            # block4 (2048 channels over 7x7 maps): its conv3 is no longer HBM-bound and the conv1 loses more on the
            # register-staged loader than the conv3 gains (measured: gpurun_out/r2/lt_res_xpre.txt vs lt_res_stored.txt)
            if getattr(b, "defer_preact", False) and nbase * 4 == depth and depth <= 1024:
                kw["defer"] = True
        res = b.conv(r, sc + "/conv3", depth, 1, 1, "SAME", norm=None, relu=False,
                     residual=shortcut, next_preact=nxt, **kw)                     # :87-91
        net, preact = res if nxt is not None else (res, None)
        b.end_points[sc] = net
        if u == n_units - 1:
            b.end_points["%s/%s" % (scope, bname)] = net                           # resnet_utils.py:181
            if "%s/%s" % (scope, bname) in keep:
                b.keep(net)
    return net


BACKBONES = {"inception_v3": build_inception_v3, "resnet_v2_50": build_resnet_v2_50}
# (raw-descriptor tap, final-descriptor tap): nets/model.py:144,149 for ResNet; for Inception only the
# final tap is written in the reference (model.py:193), the raw tap defaults to the stride-16 stage.
TAPS = {"resnet_v2_50": ("resnet_v2_50/block3", "resnet_v2_50/block4"),
        "inception_v3": ("Mixed_6e", "Mixed_7c")}


# where three-plane intermediates pay (measured per block on MI355X, tools/layer_times.py --p3 none|default|all,
# gpurun_out/r2/lt_x3_p3_*.txt): every Mixed block and Conv2d_3b -> Conv2d_4a; the 32-channel stem pair Conv2d_2a -> 2b
# runs faster on fp32 storage (halo kernel for 2a; at N = 64 the 9x im2col re-read of a 6-byte operand is L2-bound).
# "concat" (block outputs / MaxPool_5a in three planes too, so the sibling GEMMs and the strided 3x3 of Mixed_6a / 7a
# also take the LDS-DMA kernel) is built and tested but measured neutral to slightly negative (the 6-byte stores of
# every concat writer and the pools cost what the sibling GEMMs gain: gpurun_out/r2/lt_x3_p3c_*.txt): not in the default
P3_DEFAULT_BLOCKS = ("Conv2d_3b_1x1", "Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e", "Mixed_7a", "Mixed_7b",
                     "Mixed_7c")

DTYPES = {"f32": _lib.GV_F32, "bf16": _lib.GV_BF16, "f16": _lib.GV_F16}
TORCH_DTYPES = {_lib.GV_F32: torch.float32, _lib.GV_BF16: torch.bfloat16, _lib.GV_F16: torch.float16}
MATH_MODES = {"f32": _lib.GV_MATH_F32, "bf16x3": _lib.GV_MATH_BF16X3, "bf16x2": _lib.GV_MATH_BF16X2,
              "bf16x1": _lib.GV_MATH_BF16X1}


def make_plan(backbone, nb, height, width, device, raw_tap=None, final_tap=None, dtype=_lib.GV_F32,
              math="f32", lanes=True, p3=True, defer_preact=True, fuse_maxpool=True, fuse_chain=True, fuse_unit=True,
              fuse_proj=True):
    """p3: under fp32 storage + math 'bf16x3', keep conv -> conv intermediates as three bf16 planes (value neutral:
    the planes sum exactly to the fp32 value and the products are the same six MFMAs in the same order).  True: in the
    blocks of P3_DEFAULT_BLOCKS; "all": everywhere; a collection of block names: there; False: nowhere.
    fuse_maxpool: 16-bit storage issues `Conv2d_2b_3x3 -> MaxPool_3a_3x3` (Inception) / `conv1 -> pool1` (ResNet) as ONE
    launch that writes only the pooled tensor (GV_CONV_MAXPOOL3S2[_SAME]; bit-identical; False = the two launches)."""
    dtype = DTYPES[dtype] if isinstance(dtype, str) else dtype
    b = BackbonePlan(nb, height, width, dtype, MATH_MODES[math] if isinstance(math, str) else math)
    b.use_lanes = bool(lanes)
    b.use_p3 = b.use_p3 and bool(p3)
    b.defer_preact = b.defer_preact and bool(defer_preact)      # (A/B switch: False = every pre-activation stored)
    b.fuse_maxpool = b.fuse_maxpool and bool(fuse_maxpool)      # (A/B switch: False = Conv2d_2b and MaxPool_3a as two launches)
    # conv3 + next preact + conv1 as one launch (ResNet-v2 blocks 1 and 2 on 16-bit storage); GV_NO_CHAIN=1: whole-plan A/B
    b.fuse_chain = bool(fuse_chain) and dtype != _lib.GV_F32 and os.environ.get("GV_NO_CHAIN") is None
    b.fuse_unit = (fuse_unit if b.fuse_chain and fuse_unit and os.environ.get("GV_NO_UNIT") is None else False)   # (GV_NO_UNIT=1: chain only, A/B)
    if b.fuse_unit and os.environ.get("GV_UNIT_ALL") is not None:
        b.fuse_unit = "all"
    b.fuse_pair = bool(fuse_chain) and dtype != _lib.GV_F32 and os.environ.get("GV_NO_PAIR") is None
    b.fuse_pool_act = b.fuse_maxpool and dtype != _lib.GV_F32 and os.environ.get("GV_NO_POOL_ACT") is None
    # the first ResNet unit's projection shortcut inside its conv3 GEMM (GV_CHAIN_PROJ; GV_NO_PROJ=1: whole-plan A/B)
    b.fuse_proj = b.fuse_chain and bool(fuse_proj) and os.environ.get("GV_NO_PROJ") is None
    if isinstance(p3, (set, frozenset, list, tuple)):
        b.p3_blocks = set(p3)
    elif p3 is True:
        b.p3_blocks = set(P3_DEFAULT_BLOCKS)
    else:
        b.p3_blocks = None                 # "all": every conv -> conv intermediate
    raw_tap = raw_tap or TAPS[backbone][0]
    final_tap = final_tap or TAPS[backbone][1]
    if backbone == "inception_v3":
        build_inception_v3(b, final_endpoint=final_tap if final_tap in INCEPTION_ENDPOINTS else "Mixed_7c",
                           keep=(raw_tap, final_tap))
    elif backbone == "resnet_v2_50":
        build_resnet_v2_50(b, keep=(raw_tap, final_tap))
    else:
        raise ValueError("unknown backbone %r" % backbone)
    for t in (raw_tap, final_tap):
        if t not in b.end_points:
            raise ValueError("end point %r not in backbone %s" % (t, backbone))
        b.keep(b.end_points[t])
    b.raw_tap, b.final_tap = raw_tap, final_tap
    plan = b.lower(device)
    # The builder decides a conv -> max-pool fusion from a Python copy of the kernels' predicates (fused_maxpool_ok) and
    # never allocates the un-pooled tensor; the KERNEL is the authority.  On a device, ask it once (one launch of each
    # fused op on the plan's own buffers): if it declines (GV_E_UNSUPPORTED), rebuild the plan with the two launches.
    if (b.fuse_maxpool or b.fuse_chain) and torch.device(device).type == "cuda" and plan.declined_fused_pools():
        del plan
        return make_plan(backbone, nb, height, width, device, raw_tap=raw_tap, final_tap=final_tap, dtype=dtype, math=math,
                         lanes=lanes, p3=p3, defer_preact=defer_preact, fuse_maxpool=False, fuse_chain=False, fuse_unit=False)
    return plan
