"""Host-side mirror of the reference's `nets/model.py` on top of the HIP library.

Reference surface reproduced (same names, argument meaning and error behaviour):

    group_scheme(view_discrimination_score, num_group, num_views)      nets/model.py:16
    group_weight(g_schemes)                                            nets/model.py:28
    view_pooling(final_view_descriptors, group_scheme)                 nets/model.py:44
    group_fusion(group_descriptors, group_weight)                      nets/model.py:77
    gvcnn(inputs, num_classes, group_scheme, group_weight, ...)        nets/model.py:105
    basic(inputs, num_classes, ...)                                    nets/model.py:169

plus the fused, device-resident forms BASELINE.json:north_star asks for:

    GVCNN(...).forward(views)            scores -> scheme/weight -> pooling -> fusion -> logits,
                                         nothing visits the host between the backbone and the logits
    GVCNN.forward_phase1 / forward_phase2  the two-partial_run protocol of train.py:264-288
    grouping_module(scores, num_groups)  scheme + weight on device

Every array that crosses this API is a torch tensor on the HIP device (PyTorch is plumbing:
device memory + streams).  All arithmetic happens in libgvcnn_hip.so; there is no CPU path.
"""
import numpy as np
import torch

from . import _lib
from . import backbones
from . import params as _params

AUTO_REUSE = "AUTO_REUSE"          # tf.compat.v1.AUTO_REUSE stand-in (model.py:106)

_POOL_MODES = {"max": _lib.GV_VIEWPOOL_MAX, "mean": _lib.GV_VIEWPOOL_MEAN}


def _st(stream=None):
    return backbones._stream_ptr(stream)


def pin_device(cls):
    """Class decorator: every public method runs with the engine's OWN device current, so an engine built on cuda:1
    launches on cuda:1 and its stream whatever the caller's current device is (the C ABI launches on the current
    device; `_st()` resolves the current stream of the current device)."""
    import functools

    def wrap(fn):
        @functools.wraps(fn)
        def inner(self, *a, **k):
            with torch.cuda.device(self.device):
                return fn(self, *a, **k)
        return inner
    for name, fn in list(vars(cls).items()):
        if callable(fn) and not name.startswith("_") and not isinstance(fn, (staticmethod, classmethod, type)):
            setattr(cls, name, wrap(fn))
    return cls


def _dev(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("gvcnn-tf_amd needs a HIP device (MI355X); there is no CPU fallback")
    return torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())


# ------------------------------------------------------------------------------------------------
# grouping module — device functions
# ------------------------------------------------------------------------------------------------
def grouping_module(scores, num_groups, num_bins=10, check=True):
    """scores: device fp32 [V].  Returns (gidx int32 [V], scheme int32 [G,V], weight fp32 [G]) on
    device — nets/model.py:16-41 without the host round trip of train.py:270-288.
    num_bins=10 is the literal of model.py:23.  With check=True an out-of-range bin raises
    IndexError like the reference's numpy indexing (this reads one int back: a sync)."""
    lib = _lib.load()
    scores = scores.contiguous()
    assert scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 1
    V = scores.numel()
    dev = scores.device
    gidx = torch.empty(V, dtype=torch.int32, device=dev)
    scheme = torch.empty((num_groups, V), dtype=torch.int32, device=dev)
    weight = torch.empty(num_groups, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.gv_group_assign(scores.data_ptr(), V, num_groups, num_bins, gidx.data_ptr(),
                                   scheme.data_ptr(), weight.data_ptr(), status.data_ptr(), _st()),
               "gv_group_assign")
    if check:
        _raise_for_status(int(status.item()), gidx, num_groups)
    return gidx, scheme, weight


def _raise_for_status(st, gidx, num_groups):
    if st & 2:
        raise ValueError("cannot convert float NaN to integer")
    if st & 1:
        bad = [int(b) for b in gidx.tolist() if b >= num_groups or b < 0]
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (bad[0], num_groups))


def group_scheme(view_discrimination_score, num_group, num_views):
    """nets/model.py:16-25.  `view_discrimination_score` is a 1-element sequence wrapping the V
    scores (the reference indexes [0]).  Returns an int64 numpy array [num_group, num_views] like the
    reference; the binning itself runs on the device (gv_group_assign)."""
    s = view_discrimination_score[0]
    dev = _dev(s.device if torch.is_tensor(s) and s.is_cuda else None)
    if torch.is_tensor(s):
        sc = s.detach().to(device=dev, dtype=torch.float32).reshape(-1)
    else:
        sc = torch.from_numpy(np.asarray([float(np.float32(x)) for x in s], dtype=np.float32)).to(dev)
    assert sc.numel() == num_views or num_views >= sc.numel()
    pad = num_views - sc.numel()
    _, scheme, _ = grouping_module(sc, num_group, 10, check=True)
    out = scheme.cpu().numpy().astype(np.int64)
    if pad:
        out = np.concatenate([out, np.zeros((num_group, pad), dtype=np.int64)], axis=1)
    return out


def group_weight(g_schemes):
    """nets/model.py:28-41: weight[g] = 1 + number of entries equal to 1 in row g (fp32 numpy)."""
    lib = _lib.load()
    dev = _dev(g_schemes.device if torch.is_tensor(g_schemes) and g_schemes.is_cuda else None)
    sch = torch.as_tensor(np.asarray(g_schemes.cpu() if torch.is_tensor(g_schemes) else g_schemes))
    sch = sch.to(device=dev, dtype=torch.int32).contiguous()
    G, V = sch.shape
    w = torch.empty(G, dtype=torch.float32, device=dev)
    _lib.check(lib.gv_group_weight(sch.data_ptr(), G, V, w.data_ptr(), _st()), "gv_group_weight")
    return w.cpu().numpy()


def _pool_fuse(F, V, N, E, view_stride, shape_stride, scheme, weight, mode, fill, want_D, want_S):
    lib = _lib.load()
    dev = F.device
    G = scheme.shape[0]
    D = torch.empty((G, N, E), dtype=torch.float32, device=dev) if want_D else None
    S = torch.empty((N, E), dtype=torch.float32, device=dev) if want_S else None
    _lib.check(lib.gv_view_pool_fuse_fwd(F.data_ptr(), V, N, E, view_stride, shape_stride,
                                         scheme.data_ptr(), G, weight.data_ptr(), _POOL_MODES[mode],
                                         float(fill), D.data_ptr() if want_D else None,
                                         S.data_ptr() if want_S else None, _lib.GV_F32, _st()),
               "gv_view_pool_fuse_fwd")
    return D, S


def _as_dev_scheme(scheme, dev):
    if torch.is_tensor(scheme):
        return scheme.to(device=dev, dtype=torch.int32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(scheme), dtype=np.int32)).to(dev)


def _as_dev_weight(weight, dev):
    if torch.is_tensor(weight):
        return weight.to(device=dev, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(weight), dtype=np.float32)).to(dev)


def view_pooling(final_view_descriptors, group_scheme, pool="max", empty_fill=1.0):
    """nets/model.py:44-74.  final_view_descriptors: list of V device tensors [N,h,w,C] (or one
    stacked [V,N,h,w,C]); group_scheme [G,V].  Returns {g: [N,h,w,C]} — reduce_max over the group's
    views, ones for an empty group.  (pool='mean', empty_fill=0 is unit_test.py:21,30.)"""
    if torch.is_tensor(final_view_descriptors):
        stacked = final_view_descriptors
    else:
        stacked = torch.stack(list(final_view_descriptors), dim=0)       # the tf.stack of model.py:63
    stacked = stacked.to(torch.float32).contiguous()
    V, N = stacked.shape[:2]
    E = stacked[0, 0].numel()
    dev = stacked.device
    sch = _as_dev_scheme(group_scheme, dev)
    ones = torch.ones(sch.shape[0], dtype=torch.float32, device=dev)
    D, _ = _pool_fuse(stacked, V, N, E, N * E, E, sch, ones, pool, empty_fill, True, False)
    D = D.reshape((sch.shape[0],) + tuple(stacked.shape[1:]))
    return {g: D[g] for g in range(sch.shape[0])}


def group_fusion(group_descriptors, group_weight):
    """nets/model.py:77-102: sum_g w_g*D_g / sum_g w_g.  Runs the same fused kernel with every
    group descriptor presented as a one-member group."""
    keys = list(group_descriptors.keys())
    stacked = torch.stack([group_descriptors[k] for k in keys], dim=0).to(torch.float32).contiguous()
    G, N = stacked.shape[:2]
    E = stacked[0, 0].numel()
    dev = stacked.device
    w = _as_dev_weight(group_weight, dev)
    w = w[torch.as_tensor(keys, device=dev, dtype=torch.long)].contiguous()
    eye = torch.eye(G, dtype=torch.int32, device=dev)
    _, S = _pool_fuse(stacked, G, N, E, N * E, E, eye, w, "max", 1.0, False, True)
    return S.reshape(stacked.shape[1:])


# ------------------------------------------------------------------------------------------------
# the engine: one built "graph" per (backbone, N, V, H, W) like train.py:121-136 builds it once
# ------------------------------------------------------------------------------------------------
@pin_device
class GVCNN:
    """Per-view backbone + grouping module for a fixed batch geometry.

    views layout: [N, V, H, W, 3] fp32 on the device, values in [-0.5, 0.5] (train_data.py:101).
    The view batch is folded to [N*V] images in memory order (image b = n*V + v): no transpose,
    no per-view gather (model.py:126-130 costs two copies of the input in the reference)."""

    def __init__(self, backbone="resnet_v2_50", num_shapes=1, num_views=12, height=224, width=224,
                 num_classes=40, num_group=10, backbone_params=None, head_params=None, device=None,
                 raw_tap=None, final_tap=None, num_bins=10, pool="max", empty_fill=1.0, seed=2,
                 math="f32", lanes=True, storage="f32", per_shape=False, weight_mode="count", p3=True):
        """storage: 'f32' (configs c1/c2), 'bf16' (c3/c4) or 'f16' (c5): the type activations, filters and
        descriptors are kept in; accumulation and every epilogue are fp32 (`math` applies to 'f32' only)."""
        self.lib = _lib.load()
        # per_shape: the paper's grouping — every shape scores / bins / fuses its own views; the
        # reference's batch-mean score (model.py:146) is the default.  weight_mode: 'count' (model.py:28-41) or
        # 'mean_score' (score-derived group weights)
        self.per_shape = bool(per_shape)
        self.weight_mode = {"count": _lib.GV_WEIGHT_COUNT, "mean_score": _lib.GV_WEIGHT_MEAN_SCORE}[weight_mode]
        if self.weight_mode != _lib.GV_WEIGHT_COUNT and not self.per_shape:
            raise ValueError("weight_mode='mean_score' needs per_shape=True")
        self.dtype = backbones.DTYPES[storage]
        self.tdtype = backbones.TORCH_DTYPES[self.dtype]
        self.device = _dev(device)
        self.backbone = backbone
        self.N, self.V, self.H, self.W = num_shapes, num_views, height, width
        self.num_classes, self.G, self.num_bins = num_classes, num_group, num_bins
        self.pool, self.empty_fill = pool, float(empty_fill)
        if num_views > 64 or num_group > 64:
            raise ValueError("num_views and num_group are limited to 64")
        with torch.cuda.device(self.device):
            self.plan = backbones.make_plan(backbone, num_shapes * num_views, height, width,
                                            self.device, raw_tap, final_tap, dtype=self.dtype, math=math,
                                            lanes=lanes, p3=p3)
            self.raw = self.plan.end_points[self.plan.raw_tap]
            self.final = self.plan.end_points[self.plan.final_tap]
            if backbone_params is None:
                backbone_params = _params.init_backbone_params(self.plan.param_shapes(), seed=seed)
            self.plan.bind(backbone_params)
            if head_params is None:
                head_params = _params.init_head_params(num_views, self.raw.c, self.final.c,
                                                       num_classes, seed=seed + 1)
            self.set_head(head_params)
            dev, f32, i32 = self.device, torch.float32, torch.int32
            nb = num_shapes * num_views
            self.r_img = torch.empty(nb, dtype=f32, device=dev)
            self.scores = torch.empty(num_views, dtype=f32, device=dev)
            self.gidx = torch.empty(num_views, dtype=i32, device=dev)
            self.scheme = torch.empty((num_group, num_views), dtype=i32, device=dev)
            self.weight = torch.empty(num_group, dtype=f32, device=dev)
            self.status = torch.zeros(1, dtype=i32, device=dev)
            f = self.final
            self.shape_descriptor = torch.empty((num_shapes, f.h, f.w, f.c), dtype=self.tdtype, device=dev)
            self.gap = torch.empty((num_shapes, f.c), dtype=f32, device=dev)
            self.logits = torch.empty((num_shapes, num_classes), dtype=f32, device=dev)
            if self.per_shape:
                self.scores_ps = torch.empty((num_shapes, num_views), dtype=f32, device=dev)
                self.gidx_ps = torch.empty((num_shapes, num_views), dtype=i32, device=dev)
                self.scheme_ps = torch.empty((num_shapes, num_group, num_views), dtype=i32, device=dev)
                self.weight_ps = torch.empty((num_shapes, num_group), dtype=f32, device=dev)
            self._all_ones_scheme = torch.ones((1, num_views), dtype=i32, device=dev)
            self._one = torch.ones(1, dtype=f32, device=dev)

    # -- parameters -------------------------------------------------------------------------------
    def set_head(self, H):
        dev = self.device
        ks, bs = [], []
        for v in range(self.V):
            kn, bn = _params.scorer_names(v)
            ks.append(torch.as_tensor(H[kn], dtype=torch.float32).reshape(-1))
            bs.append(torch.as_tensor(H[bn], dtype=torch.float32).reshape(-1)[:1])
        self.score_kernel = torch.stack(ks).to(dev).contiguous()            # [V, Cr]
        self.score_bias = torch.cat(bs).to(dev).contiguous()                # [V]
        assert self.score_kernel.shape == (self.V, self.raw.c)
        kn, bn = _params.classifier_names(self.V)
        self.cls_kernel = torch.as_tensor(H[kn], dtype=torch.float32).to(dev).contiguous()
        self.cls_bias = torch.as_tensor(H[bn], dtype=torch.float32).to(dev).contiguous()
        assert self.cls_kernel.shape == (self.final.c, self.num_classes)

    def _check_input(self, views):
        if not (torch.is_tensor(views) and views.is_cuda):
            raise TypeError("views must be a torch tensor on the HIP device")
        if tuple(views.shape) != (self.N, self.V, self.H, self.W, 3):
            raise ValueError("views has shape %s, engine was built for %s"
                             % (tuple(views.shape), (self.N, self.V, self.H, self.W, 3)))
        if views.dtype != torch.float32 or not views.is_contiguous():
            views = views.to(torch.float32).contiguous()
        return views

    # -- phase 1: backbone + scores (train.py:270-276) --------------------------------------------
    def run_backbone(self, views):
        self.plan.run(self._check_input(views))

    def compute_scores(self):
        """r_img from the raw-descriptor tap, then the V scores (model.py:144-147)."""
        lib, r = self.lib, self.raw
        raw_ptr = self.plan.view(r).data_ptr()
        _lib.check(lib.gv_view_score_partial(raw_ptr, r.nb, r.h * r.w, r.c, r.ld,
                                             self.score_kernel.data_ptr(), self.score_bias.data_ptr(),
                                             self.V, _lib.GV_ORDER_SHAPE_MAJOR, self.r_img.data_ptr(),
                                             self.dtype, _st()), "gv_view_score_partial")
        self.finalize_scores(self.r_img, self.N)
        return self.scores

    def finalize_scores(self, r_img, num_shapes, order=_lib.GV_ORDER_SHAPE_MAJOR):
        _lib.check(self.lib.gv_view_score_finalize(r_img.data_ptr(), num_shapes, self.V, order,
                                                   self.scores.data_ptr(), _st()),
                   "gv_view_score_finalize")
        return self.scores

    def forward_phase1(self, views):
        """Returns the V view discrimination scores (device tensor [V]); descriptors stay on device."""
        self.run_backbone(views)
        return self.compute_scores()

    # -- phase 2: pooling + fusion + classifier (train.py:281-288) -------------------------------
    def assign_groups(self, check=True):
        _lib.check(self.lib.gv_group_assign(self.scores.data_ptr(), self.V, self.G, self.num_bins,
                                            self.gidx.data_ptr(), self.scheme.data_ptr(),
                                            self.weight.data_ptr(), self.status.data_ptr(), _st()),
                   "gv_group_assign")
        if check:
            self.check_status()
        return self.scheme, self.weight

    def check_status(self):
        _raise_for_status(int(self.status.item()), self.gidx, self.G)

    def final_view_descriptors(self):
        """[N, V, h, w, C] view of the final-descriptor tap (shape-major image order)."""
        f = self.final
        return self.plan.view(f).view(self.N, self.V, f.h, f.w, f.c)

    def raw_view_descriptors(self):
        r = self.raw
        return self.plan.view(r).view(self.N, self.V, r.h, r.w, r.c)

    def pool_fuse_classify(self, scheme, weight, F=None, num_shapes=None, out=None):
        """view_pooling + group_fusion + GAP + Dense (model.py:154-164) on device.
        F: optional descriptor tensor [N', V, h, w, C] (defaults to this engine's tap)."""
        lib, f = self.lib, self.final
        E = f.h * f.w * f.c
        if F is None:
            assert f.ld == f.c
            F_ptr, N = self.plan.view(f).data_ptr(), self.N
            S, gap, logits = self.shape_descriptor, self.gap, self.logits
        else:
            F = F.to(self.tdtype).contiguous()
            N = F.shape[0]
            F_ptr = F.data_ptr()
            S = torch.empty((N, f.h, f.w, f.c), dtype=self.tdtype, device=self.device)
            gap = torch.empty((N, f.c), dtype=torch.float32, device=self.device)
            logits = torch.empty((N, self.num_classes), dtype=torch.float32, device=self.device)
        G = scheme.shape[0]
        _lib.check(lib.gv_view_pool_fuse_fwd(F_ptr, self.V, N, E, E, self.V * E, scheme.data_ptr(), G,
                                             weight.data_ptr(), _POOL_MODES[self.pool], self.empty_fill,
                                             None, S.data_ptr(), self.dtype, _st()),
                   "gv_view_pool_fuse_fwd")
        _lib.check(lib.gv_global_avg_pool(S.data_ptr(), N, f.h * f.w, f.c, f.c, gap.data_ptr(),
                                          self.dtype, _st()), "gv_global_avg_pool")
        _lib.check(lib.gv_dense_fwd(gap.data_ptr(), N, f.c, self.cls_kernel.data_ptr(),
                                    self.cls_bias.data_ptr(), self.num_classes, logits.data_ptr(),
                                    _st()), "gv_dense_fwd")
        return S, logits

    def forward_phase2(self, g_scheme=None, g_weight=None):
        """Feed scheme/weight (host arrays like train.py:281-288, or device tensors); None = use the
        device-resident result of assign_groups()."""
        if g_scheme is None:
            scheme, weight = self.scheme, self.weight
        else:
            scheme = _as_dev_scheme(g_scheme, self.device)
            weight = _as_dev_weight(g_weight, self.device)
        return self.pool_fuse_classify(scheme, weight)

    def forward_per_shape(self, views, check=True):
        """Per-shape grouping: (scores [N,V], shape_descriptor, logits).  Nothing couples the shapes
        of the batch, so a shape-sharded job needs no exchange."""
        lib, f = self.lib, self.final
        self.forward_phase1(views)                     # fills r_img (the batch-mean scores are not used)
        _lib.check(lib.gv_view_score_per_shape(self.r_img.data_ptr(), self.N * self.V, self.scores_ps.data_ptr(),
                                               _st()), "gv_view_score_per_shape")
        _lib.check(lib.gv_group_assign_per_shape(self.scores_ps.data_ptr(), self.N, self.V, self.G, self.num_bins,
                                                 self.weight_mode, self.gidx_ps.data_ptr(), self.scheme_ps.data_ptr(),
                                                 self.weight_ps.data_ptr(), self.status.data_ptr(), _st()),
                   "gv_group_assign_per_shape")
        E = f.h * f.w * f.c
        S = self.shape_descriptor
        _lib.check(lib.gv_view_pool_fuse_fwd_per_shape(self.plan.view(f).data_ptr(), self.V, self.N, E, E, self.V * E,
                                                       self.scheme_ps.data_ptr(), self.G, self.weight_ps.data_ptr(),
                                                       _POOL_MODES[self.pool], self.empty_fill, None, S.data_ptr(),
                                                       self.dtype, _st()), "gv_view_pool_fuse_fwd_per_shape")
        _lib.check(lib.gv_global_avg_pool(S.data_ptr(), self.N, f.h * f.w, f.c, f.c, self.gap.data_ptr(),
                                          self.dtype, _st()), "gv_global_avg_pool")
        _lib.check(lib.gv_dense_fwd(self.gap.data_ptr(), self.N, f.c, self.cls_kernel.data_ptr(),
                                    self.cls_bias.data_ptr(), self.num_classes, self.logits.data_ptr(), _st()),
                   "gv_dense_fwd")
        if check:
            st = int(self.status.item())
            if st:
                _raise_for_status(st, self.gidx_ps.reshape(-1), self.G)
        return self.scores_ps, S, self.logits

    def forward(self, views, check=True):
        """Fused forward: (scores [V], shape_descriptor [N,h,w,C], logits [N,num_classes]), all on
        device; the only host interaction is the optional status read (check=True)."""
        if self.per_shape:
            return self.forward_per_shape(views, check)
        self.forward_phase1(views)
        self.assign_groups(check=False)
        S, logits = self.pool_fuse_classify(self.scheme, self.weight)
        if check:
            self.check_status()
        return self.scores, S, logits

    # -- hipGraph replay (small batches are launch-bound) ---------------------------------------------------
    def capture(self, views):
        """Record one fused forward on `views` (a device tensor that stays alive: the graph keeps its address) into a
        hipGraph and return replay(): one launch per step instead of ~90.  Replays re-read `views` in place — copy the
        next batch INTO it.  The status word is not checked inside the graph; call check_status() when needed."""
        import ctypes as C
        views = self._check_input(views)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        handle = C.c_void_p()
        with torch.cuda.stream(side):
            self.forward(views, check=False)                 # warm: lazily created streams / attributes exist
            side.synchronize()
            _lib.check(self.lib.gv_capture_begin(side.cuda_stream), "gv_capture_begin")
            try:
                self.forward(views, check=False)
            finally:
                rc = self.lib.gv_capture_end(side.cuda_stream, C.byref(handle))
            _lib.check(rc, "gv_capture_end")
        self._graphs = getattr(self, "_graphs", [])
        self._graphs.append((handle, views, side))

        def replay():
            with torch.cuda.device(self.device):
                _lib.check(self.lib.gv_graph_launch(handle, _st()), "gv_graph_launch")
            return (self.scores_ps if self.per_shape else self.scores), self.shape_descriptor, self.logits
        replay.close = self.close_graphs
        return replay

    def close_graphs(self):
        """Destroy every captured hipGraph (exec + graph) of this engine; replay functions are dead afterwards."""
        for handle, _views, side in getattr(self, "_graphs", []):
            side.synchronize()
            self.lib.gv_graph_destroy(handle)
        self._graphs = []

    def __del__(self):
        try:
            self.close_graphs()
        except Exception:
            pass

    def forward_basic(self, views):
        """nets/model.py:169-206 (MVCNN baseline): max over all views -> GAP -> Dense."""
        self.run_backbone(views)
        lib, f = self.lib, self.final
        E = f.h * f.w * f.c
        S = self.shape_descriptor
        _lib.check(lib.gv_view_pool_fuse_fwd(self.plan.view(f).data_ptr(), self.V, self.N, E, E,
                                             self.V * E, self._all_ones_scheme.data_ptr(), 1,
                                             self._one.data_ptr(), _lib.GV_VIEWPOOL_MAX, 1.0, None,
                                             S.data_ptr(), self.dtype, _st()), "gv_view_pool_fuse_fwd")
        _lib.check(lib.gv_global_avg_pool(S.data_ptr(), self.N, f.h * f.w, f.c, f.c,
                                          self.gap.data_ptr(), self.dtype, _st()), "gv_global_avg_pool")
        _lib.check(lib.gv_dense_fwd(self.gap.data_ptr(), self.N, f.c, self.cls_kernel.data_ptr(),
                                    self.cls_bias.data_ptr(), self.num_classes,
                                    self.logits.data_ptr(), _st()), "gv_dense_fwd")
        return S, self.logits


# ------------------------------------------------------------------------------------------------
# reference-shaped functional entry points with a variable store (AUTO_REUSE semantics)
# ------------------------------------------------------------------------------------------------
_ENGINES = {}
_CONFIG = {"backbone": "resnet_v2_50",          # the live call at nets/model.py:137-141
           "backbone_params": None, "head_params": None, "raw_tap": None, "final_tap": None}


def configure(backbone=None, backbone_params=None, head_params=None, raw_tap=None, final_tap=None):
    """Choose the backbone ('resnet_v2_50' as at HEAD of the reference, or 'inception_v3', the
    commented call at model.py:131-136) and optionally supply variables by their slim/Keras names.
    Clears the engine cache (the analogue of resetting the TF default graph)."""
    for k, v in (("backbone", backbone), ("backbone_params", backbone_params),
                 ("head_params", head_params), ("raw_tap", raw_tap), ("final_tap", final_tap)):
        if v is not None:
            _CONFIG[k] = v
    _ENGINES.clear()
    _TRAIN_ENGINES.clear()


def _engine(inputs, num_classes, num_group):
    if not (torch.is_tensor(inputs) and inputs.is_cuda and inputs.dim() == 5 and inputs.shape[-1] == 3):
        raise TypeError("inputs must be a device tensor [N, V, H, W, 3]")
    N, V, H, W, _ = inputs.shape
    key = (_CONFIG["backbone"], N, V, H, W, num_classes, num_group, str(inputs.device))
    if key not in _ENGINES:
        _ENGINES[key] = GVCNN(_CONFIG["backbone"], N, V, H, W, num_classes, num_group,
                              _CONFIG["backbone_params"], _CONFIG["head_params"], inputs.device,
                              _CONFIG["raw_tap"], _CONFIG["final_tap"])
    return _ENGINES[key]


_TRAIN_ENGINES = {}


def _train_engine(inputs, num_classes, num_group):
    """Engine for is_training=True: batch-statistics BatchNorm per view (and the backward pass,
    gvcnn-tf_amd/training.py).  Shares the variable store of configure()."""
    from .training import TrainGVCNN
    N, V, H, W, _ = inputs.shape
    key = (_CONFIG["backbone"], N, V, H, W, num_classes, num_group, str(inputs.device))
    if key not in _TRAIN_ENGINES:
        _TRAIN_ENGINES[key] = TrainGVCNN(_CONFIG["backbone"], N, V, H, W, num_classes, num_group,
                                         _CONFIG["backbone_params"], _CONFIG["head_params"], inputs.device,
                                         _CONFIG["raw_tap"], _CONFIG["final_tap"])
    return _TRAIN_ENGINES[key]


def gvcnn(inputs, num_classes, group_scheme, group_weight, is_training=True,
          dropout_keep_prob=0.8, reuse=AUTO_REUSE):
    """nets/model.py:105-166.  Returns (view_discrimination_scores: list of V 0-d tensors,
    shape_descriptor [N,h,w,C], logits [N,num_classes]).  group_scheme [G,V] / group_weight [G] are the
    fed placeholders of train.py:127-128 (host arrays or device tensors).  dropout_keep_prob is
    accepted and has no effect, as in the reference (it only reaches heads that are never fetched).
    Variables are created on first use and reused afterwards (AUTO_REUSE).  is_training=True (the
    reference default) normalises with batch statistics per view, like the V graph copies of the reference."""
    G = (group_scheme.shape[0] if hasattr(group_scheme, "shape") else len(group_scheme))
    if is_training:
        teng = _train_engine(inputs, num_classes, G)
        scores, S, logits, _ = teng.forward(inputs, None, g_scheme=np.asarray(
            group_scheme.cpu() if torch.is_tensor(group_scheme) else group_scheme), g_weight=np.asarray(
            group_weight.cpu() if torch.is_tensor(group_weight) else group_weight))
        return [scores[v] for v in range(teng.V)], S, logits
    eng = _engine(inputs, num_classes, G)
    scores = eng.forward_phase1(inputs)
    S, logits = eng.forward_phase2(group_scheme, group_weight)
    return [scores[v] for v in range(eng.V)], S, logits


def basic(inputs, num_classes, is_training=True, dropout_keep_prob=0.8, reuse=AUTO_REUSE):
    """nets/model.py:169-206."""
    if is_training:                                   # max over all views == one group holding every view
        teng = _train_engine(inputs, num_classes, 1)
        _, S, logits, _ = teng.forward(inputs, None, g_scheme=np.ones((1, inputs.shape[1]), np.int32),
                                       g_weight=np.ones(1, np.float32))
        return S, logits
    eng = _engine(inputs, num_classes, 1)
    return eng.forward_basic(inputs)


def gvcnn_fused(views, num_classes, num_groups, check=True):
    """north_star form: model.gvcnn(views, num_classes, num_groups) with scheme/weight on device."""
    eng = _engine(views, num_classes, num_groups)
    return eng.forward(views, check=check)
