"""Training step of the hot path: forward in train mode + loss + backward + Momentum.

What the reference does per step (train.py:259-288, utils/train_utils.py:217-259):
  partial_run #1  backbone with is_training=True (BatchNorm on BATCH statistics, one graph copy per view
                  => statistics over the N*h*w values of one view) -> the V view scores
  host            group_scheme / group_weight (constants of the backward pass: fed placeholders)
  partial_run #2  view pooling + group fusion + classifier -> mean sparse-softmax CE (+ L2), gradients
                  wrt every backbone variable (shared by the V views => summed), BN beta/gamma, the
                  classifier; the V scorer Dense(1) layers receive no gradient; MomentumOptimizer(lr, 0.9).

Here: the same op list as the inference plan, built by the same builder functions, but kept un-fused
(conv -> z, train-mode BN -> y) with every tensor retained for the backward pass; each op's backward is
one or two launches of libgvcnn_hip.so, executed in reverse order with accumulate semantics.
The data gradient of a convolution is the forward implicit-GEMM kernel itself on dZ with the filter
flipped/transposed (stride-2 layers: zero-dilated input); the filter gradient is its own kernel.
fp32 storage throughout.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import backbones
from . import params as _params
from .backbones import _out_size
from .model import _st, _dev, _raise_for_status, pin_device


class TrainPlan(backbones.BackbonePlan):
    """Symbolic op list for training: un-fused, nothing recycled."""

    def __init__(self, nb, height, width, math_mode, dtype=_lib.GV_F32):
        # (the training plan keeps every tensor in its storage type: no three-plane intermediates)
        super().__init__(nb, height, width, dtype, math_mode)
        self.use_lanes = False
        self.use_p3 = False
        self.defer_preact = False          # BatchNorm is its own op here (batch statistics)
        self.fuse_maxpool = False          # (the pool -> BatchNorm form of TrainGVCNN is the training path's fusion)

    def conv(self, x, scope, cout, k, stride=1, padding="SAME", out=None, norm=None, relu=True,
             residual=None, next_preact=None):
        kh, kw = (k, k) if isinstance(k, int) else k
        oh, pad_t = _out_size(x.h, kh, stride, padding if not isinstance(padding, tuple) else padding[0])
        ow, pad_l = _out_size(x.w, kw, stride, padding if not isinstance(padding, tuple) else padding[1])
        if norm is None and out is not None:
            z = out
        else:
            z = self.new_tensor(x.nb, oh, ow, cout)
        self.ops.append(dict(lane=self.cur_lane, kind="conv", name=scope, x=x, y=z, res=residual, kh=kh, kw=kw, stride=stride,
                             pad_t=pad_t, pad_l=pad_l, bias=None if norm is not None else scope + "/biases",
                             flops=2.0 * x.nb * oh * ow * cout * kh * kw * x.c))
        result = z
        if norm is not None:
            a = out if out is not None else self.new_tensor(x.nb, oh, ow, cout)
            self.ops.append(dict(lane=self.cur_lane, kind="bn", name=scope + "/BatchNorm", x=z, y=a, eps=norm[1], has_gamma=norm[2],
                                 relu=relu))
            result = a
        if next_preact is not None:
            pre = self.new_tensor(x.nb, oh, ow, cout)
            self.ops.append(dict(lane=self.cur_lane, kind="bn", name=next_preact[0], x=result, y=pre, eps=next_preact[1],
                                 has_gamma=True, relu=True))
            return result, pre
        return result

    def conv_siblings(self, x, branches, first_out, norm, relu=True, pooled=None):
        """The 1x1 convolutions of one block that read the SAME input (and the pooled branch's 1x1, which commutes
        with its average pool) as ONE GEMM over their concatenated filters: z_all [pixels, sum of couts], each
        member's train-mode BatchNorm reads its channel slice.  Forward reads x once instead of 4 times; backward
        is ONE data-gradient launch (no read-modify-write fan-in on dX) and ONE filter gradient that lands directly
        in the members' gradients (column slices of one block, see TrainGVCNN.__init__).  The variables keep the
        reference's names and shapes."""
        members = list(branches) + ([(pooled[0], pooled[1])] if pooled is not None else [])
        total = sum(c for _, c in members)
        zall = self.new_tensor(x.nb, x.h, x.w, total)
        cols, off = [], 0
        for scope, c in members:
            cols.append((scope + "/weights", off, c))
            off += c
        self.ops.append(dict(lane=self.cur_lane, kind="conv", name="+".join(sc for sc, _ in members), x=x, y=zall,
                             res=None, kh=1, kw=1, stride=1, pad_t=0, pad_l=0, bias=None, members=cols,
                             flops=2.0 * x.npix * total * x.c))
        outs = []
        for i, (scope, c) in enumerate(branches):
            z = zall.channels(cols[i][1], cols[i][1] + c)
            a = first_out if i == 0 else self.new_tensor(x.nb, x.h, x.w, c)
            self.ops.append(dict(lane=self.cur_lane, kind="bn", name=scope + "/BatchNorm", x=z, y=a, eps=norm[1],
                                 has_gamma=norm[2], relu=relu))
            if i:
                outs.append(a)
        if pooled is not None:
            scope, depth, pool_name, dst = pooled
            z = zall.channels(cols[-1][1], cols[-1][1] + depth)
            with self.lane(2):
                pz = self.pool(z, 3, 1, "SAME", _lib.GV_POOL_AVG, name=pool_name)
                self.ops.append(dict(lane=self.cur_lane, kind="bn", name=scope + "/BatchNorm", x=pz, y=dst, eps=norm[1],
                                     has_gamma=norm[2], relu=True))
        return outs

    def pooled_branch(self, x, conv_scope, pool_name, depth, dst, norm):
        """The pooled Inception branch as BN(avgpool(conv1x1(x))): a 1x1 convolution commutes with the average,
        so the pool (forward AND backward) moves `depth` instead of x.c channels; the train-mode BatchNorm stays
        after the pool, i.e. its batch statistics are those of the reference's tensor."""
        z = self.new_tensor(x.nb, x.h, x.w, depth)
        self.ops.append(dict(lane=self.cur_lane, kind="conv", name=conv_scope, x=x, y=z, res=None, kh=1, kw=1, stride=1, pad_t=0, pad_l=0,
                             bias=None, flops=2.0 * x.npix * depth * x.c))
        p = self.pool(z, 3, 1, "SAME", _lib.GV_POOL_AVG, name=pool_name)
        self.ops.append(dict(lane=self.cur_lane, kind="bn", name=conv_scope + "/BatchNorm", x=p, y=dst, eps=norm[1], has_gamma=norm[2],
                             relu=True))
        return dst

    def pool(self, x, k, stride, padding, mode, out=None, name="pool"):
        oh, pad_t = _out_size(x.h, k, stride, padding)
        ow, pad_l = _out_size(x.w, k, stride, padding)
        if out is None:
            out = self.new_tensor(x.nb, oh, ow, x.c)
        self.ops.append(dict(lane=self.cur_lane, kind="pool", name=name, x=x, y=out, k=k, stride=stride, pad_t=pad_t, pad_l=pad_l,
                             mode=mode))
        return out

    def bn_relu(self, x, bn_scope, eps, name):
        out = self.new_tensor(x.nb, x.h, x.w, x.c)
        self.ops.append(dict(lane=self.cur_lane, kind="bn", name=bn_scope, x=x, y=out, eps=eps, has_gamma=True, relu=True))
        return out

    def param_shapes(self):
        shapes = {}
        for op in self.ops:
            if op["kind"] == "conv":
                if op.get("members"):
                    for wname, _, c in op["members"]:
                        shapes[wname] = (1, 1, op["x"].c, c)
                    continue
                shapes[op["name"] + "/weights"] = (op["kh"], op["kw"], op["x"].c, op["y"].c)
                if op["bias"]:
                    shapes[op["bias"]] = (op["y"].c,)
            elif op["kind"] == "bn":
                c = op["x"].c
                for leaf in ("beta", "moving_mean", "moving_variance") + (("gamma",) if op["has_gamma"] else ()):
                    shapes[op["name"] + "/" + leaf] = (c,)
        return shapes


@pin_device
class TrainGVCNN:
    """One training step of GVCNN for a fixed batch geometry (views [N, V, H, W, 3] on the device)."""

    def __init__(self, backbone="resnet_v2_50", num_shapes=2, num_views=6, height=224, width=224,
                 num_classes=40, num_group=10, backbone_params=None, head_params=None, device=None,
                 raw_tap=None, final_tap=None, num_bins=10, pool="max", empty_fill=1.0, math="bf16x3", seed=2,
                 head_views=None, view_offset=0, per_shape=False, weight_mode="count", storage="f32",
                 fuse_siblings=True, frozen_bn=False, deterministic=True, dw_workspace_mb=256):
        """head_views / view_offset: view-sharded data parallelism (sharding.ShardedTrainGVCNN) — this engine
        runs the backbone for views [view_offset, view_offset + num_views) of the head_views views of every shape
        (so each view's BatchNorm statistics stay on one rank, exactly the reference's per-view statistics),
        while the grouping head works on all head_views views."""
        self.lib = _lib.load()
        # deterministic: every filter gradient through gv_conv2d_wgrad_ws — pixel slices store their partial images of dW
        # into one workspace shared by all layers and are added in slice order, so two runs of a step give the same bits
        # (the reference's CPU gradients are deterministic, utils/train_utils.py:217-259; False: fp32 atomics, A/B).
        # The workspace is re-used layer after layer, which keeps it in the 256 MiB Infinity Cache.
        self.deterministic = bool(deterministic)
        self._dw_ws_bytes = int(dw_workspace_mb) << 20
        self._dw_ws = None
        # storage: element type of activations and activation gradients in HBM.  "bf16" is BASELINE configs[2]
        # (bf16 forward + backward): 16-bit MFMA convolutions (forward, data and filter gradient), fp32 master
        # weights / parameter gradients / optimizer state, fp64 batch statistics.
        self.dt = backbones.DTYPES[storage]
        self.tdt = backbones.TORCH_DTYPES[self.dt]
        self.es = 4 if self.dt == _lib.GV_F32 else 2
        # No up-front zero fill of the activation gradients: the first contribution to a tensor's gradient in a
        # backward pass STORES (data gradient without residual, BatchNorm backward with accumulate=0, 16-bit pool
        # backward in store mode; ops that can only add get that one tensor zeroed first), later ones add; and the
        # ReLU mask of a BatchNorm backward is recomputed from z instead of read from y.
        self._lazy = True                                 # (False: the plain zero-fill + accumulate form, kept for tests)
        self.pool_argmax = True                           # max pools record their argmax (False: backward re-reads x; A/B)
        self._zacc = self.es == 2                         # pre-zeroed per-layer fp64 accumulators (16-bit entry points)
        self._written = set()
        self._lane_streams = None
        self._cur_lane = 0
        # per_shape: the paper's grouping (model.GVCNN(per_shape=True)) in the training step
        self.per_shape = bool(per_shape)
        self.weight_mode = {"count": _lib.GV_WEIGHT_COUNT, "mean_score": _lib.GV_WEIGHT_MEAN_SCORE}[weight_mode]
        # shape-sharded data parallelism (sharding.ShardedTrainGVCNN(mode='shapes')): bn_sync(accum) all-reduces the
        # per-(view, channel) BatchNorm sums over the ranks, shape_world = number of ranks sharing every view
        self.bn_sync = None
        self.bn_sync_calls = 0                            # statistics all-reduces issued so far (two per BatchNorm and step
        self.coalesce_bn_sync = True                      # uncoalesced; the members of a fused sibling GEMM share one)
        self.shape_world = 1
        # frozen_bn: BatchNorm normalises with the MOVING statistics (slim's is_training=False arithmetic) while every
        # variable still gets its gradient: dz = gamma*inv*g without the two batch-statistics terms.  The reference never
        # trains this way; it exists because train-mode statistics over a handful of samples amplify rounding
        # chaotically, and with them frozen the whole assembled step can be held to 1e-3 against the oracle.  Built from
        # the split entry points: the forward accumulator is FILLED with the sums that reproduce the moving statistics,
        # the backward accumulator is zeroed between its two halves after beta/gamma took their gradients from it.
        self.frozen_bn = bool(frozen_bn)
        self.Vh = head_views if head_views is not None else num_views
        self.view_offset = view_offset
        assert 0 <= view_offset and view_offset + num_views <= self.Vh
        self.device = dev = _dev(device)
        self.backbone = backbone
        self.N, self.V, self.H, self.W = num_shapes, num_views, height, width
        self.num_classes, self.G, self.num_bins = num_classes, num_group, num_bins
        self.pool_mode = {"max": _lib.GV_VIEWPOOL_MAX, "mean": _lib.GV_VIEWPOOL_MEAN}[pool]
        self.empty_fill = float(empty_fill)
        self.math_mode = backbones.MATH_MODES[math]
        if self.math_mode == _lib.GV_MATH_F32:
            raise ValueError("training uses the bf16-plane convolution kernels (math='bf16x3')")
        nb = num_shapes * num_views
        p = TrainPlan(nb, height, width, self.math_mode, self.dt)
        raw_tap = raw_tap or backbones.TAPS[backbone][0]
        final_tap = final_tap or backbones.TAPS[backbone][1]
        if backbone == "inception_v3":
            backbones.build_inception_v3(p, keep=(raw_tap, final_tap), fuse_siblings=fuse_siblings)
        else:
            backbones.build_resnet_v2_50(p, keep=(raw_tap, final_tap))
        self.plan = p
        self.raw, self.final = p.end_points[raw_tap], p.end_points[final_tap]
        f32 = torch.float32
        with torch.cuda.device(dev):
            self.act = [torch.empty((n + 7) // 8 * 8, dtype=self.tdt, device=dev) for n, *_ in p.vbufs]
            self.grad = [None] * len(p.vbufs)
            shapes = p.param_shapes()
            if backbone_params is None:
                backbone_params = _params.init_backbone_params(shapes, seed=seed)
            if head_params is None:
                head_params = _params.init_head_params(self.Vh, self.raw.c, self.final.c, num_classes, seed=seed + 1)
            ks, bs = [], []
            for v in range(view_offset, view_offset + num_views):
                kn, bn = _params.scorer_names(v)
                ks.append(torch.as_tensor(head_params[kn], dtype=f32).reshape(-1))
                bs.append(torch.as_tensor(head_params[bn], dtype=f32).reshape(-1)[:1])
            self.score_kernel = torch.stack(ks).to(dev).contiguous()
            self.score_bias = torch.cat(bs).to(dev).contiguous()
            kn, bn = _params.classifier_names(self.Vh)
            self.cls_names = (kn, bn)
            # Trainable variables, their gradients and the Momentum slots live in three FLAT fp32 buffers (the dicts
            # hold views): zeroing the gradients is one fill and the optimizer two launches (slim's L2 term applies to
            # the conv filters, which come first) instead of one per variable.
            init = {k: torch.as_tensor(backbone_params[k], dtype=f32) for k in shapes}
            init[kn] = torch.as_tensor(head_params[kn], dtype=f32)
            init[bn] = torch.as_tensor(head_params[bn], dtype=f32)
            moving = [k for k in init if k.endswith(("moving_mean", "moving_variance"))]
            train = [k for k in init if k not in moving]
            train = [k for k in train if k.endswith("/weights")] + [k for k in train if not k.endswith("/weights")]
            # The members of a fused sibling GEMM are COLUMN SLICES of one [1,1,cin,sum of couts] block (same place in
            # all three buffers): the fused filter gradient lands directly in the members' gradients, the forward and
            # data-gradient images are packed from the block (row stride = its width).  Their dict entries are strided
            # views with the reference's shapes.
            fused_of = {}
            for op in p.ops:
                if op["kind"] == "conv" and op.get("members"):
                    for wname, col, c in op["members"]:
                        fused_of[wname] = (op, col, c)
            offs, total = {}, 0
            for k in train:
                if total and not k.endswith("/weights") and "n_wd" not in offs:
                    offs["n_wd"] = total
                if k in fused_of:
                    op = fused_of[k][0]
                    if "flat_off" not in op:
                        op["flat_off"] = total
                        total += (op["x"].c * op["y"].c + 15) // 16 * 16
                    continue
                offs[k] = total
                total += (init[k].numel() + 15) // 16 * 16
            self._n_wd = offs.pop("n_wd", total)
            # where each conv's filter gradient starts in the flat buffer: the filters come first, in layer order, so a
            # backward pass finalises that region from its end towards offset 0 (backward_backbone(progress=...))
            lows = []
            for op in p.ops:
                if op["kind"] == "conv":
                    op["g_lo"] = op["flat_off"] if op.get("members") else offs.get(op["name"] + "/weights")
                    if op["g_lo"] is not None:
                        lows.append(op["g_lo"])
            self._g_monotone = all(a < b for a, b in zip(lows, lows[1:]))
            self._flat_p, self._flat_g, self._flat_m = (torch.zeros(total, dtype=f32, device=dev) for _ in range(3))
            self.params, self.grads, self.momentum = {}, {}, {}
            for k in train:
                if k in fused_of:
                    op, col, c = fused_of[k]
                    o, cin, tot_c = op["flat_off"], op["x"].c, op["y"].c
                    blk = lambda buf: buf[o:o + cin * tot_c].view(1, 1, cin, tot_c)
                    self.params[k] = blk(self._flat_p)[..., col:col + c]
                    self.grads[k] = blk(self._flat_g)[..., col:col + c]
                    self.momentum[k] = blk(self._flat_m)[..., col:col + c]
                    op["w_fused"], op["dw_fused"] = blk(self._flat_p), blk(self._flat_g)
                else:
                    o, n, shp = offs[k], init[k].numel(), tuple(init[k].shape)
                    self.params[k] = self._flat_p[o:o + n].view(shp)
                    self.grads[k] = self._flat_g[o:o + n].view(shp)
                    self.momentum[k] = self._flat_m[o:o + n].view(shp)
                self.params[k].copy_(init[k])
            for k in moving:
                self.params[k] = init[k].to(dev).contiguous().clone()
            # per-op device state
            cmax = max(op["x"].c for op in p.ops if op["kind"] == "bn") if any(o["kind"] == "bn" for o in p.ops) else 4
            cmax = max(cmax, max(max(op["y"].c, op["x"].c) for op in p.ops if op["kind"] == "conv"))
            self.accum = torch.zeros(2 * num_views * cmax, dtype=torch.float64, device=dev)
            # 16-bit step: one fp64 accumulator per BatchNorm layer and pass (forward sums, backward sums), all zeroed
            # by ONE fill per step instead of one memset per call
            bns = [op for op in p.ops if op["kind"] == "bn"]
            tot = max(sum(2 * num_views * op["x"].c for op in bns), 4)
            self._accum_f = torch.zeros(tot, dtype=torch.float64, device=dev)
            self._accum_b = torch.zeros(tot, dtype=torch.float64, device=dev)
            o = 0
            for op in bns:
                n = 2 * num_views * op["x"].c
                op["acc_f"], op["acc_b"] = self._accum_f[o:o + n], self._accum_b[o:o + n]
                o += n
            self.ones = torch.ones(cmax, dtype=f32, device=dev)
            self.zeros = torch.zeros(cmax, dtype=f32, device=dev)
            self._counts = {}
            for op in p.ops:
                if op["kind"] == "bn":
                    c = op["x"].c
                    op["stat"] = {k: torch.empty((num_views, c), dtype=f32, device=dev)
                                  for k in ("mean", "var", "inv", "scale", "shift")}
                elif op["kind"] == "conv":
                    kh, kw, cin, cout = op["kh"], op["kw"], op["x"].c, op["y"].c
                    nf = self.lib.gv_packed_filter_bytes(kh, kw, cin, cout, self.dt, self.math_mode)
                    nd = self.lib.gv_packed_filter_bytes(kh, kw, cout, cin, self.dt, self.math_mode)
                    op["w_fwd"] = torch.zeros(nf, dtype=torch.uint8, device=dev)
                    op["w_dgrad"] = torch.zeros(nd, dtype=torch.uint8, device=dev) if op["x"].vbuf >= 0 else None
                    # stride-2 layer on 16-bit storage: its data gradient as FOUR parity classes (gv_conv_desc.y_step):
                    # dX at rows of parity py / columns of parity px only receives the taps r = (py + pad) mod 2, +2, ...
                    # — a small stride-1 convolution over the un-dilated dZ per class, 1/4 of the multiply-adds of the
                    # zero-dilated form.  (th, tw): taps per class; pad: zero rows in front of dZ; (A, B): class outputs
                    if op["stride"] == 2 and self.es == 2 and op["x"].vbuf >= 0 and not op.get("members"):
                        cls = []
                        for py in (0, 1):
                            for px in (0, 1):
                                r0, s0 = (py + op["pad_t"]) % 2, (px + op["pad_l"]) % 2
                                th, tw = len(range(r0, kh, 2)), len(range(s0, kw, 2))
                                pt = (th - 1) - (py + op["pad_t"] - r0) // 2
                                pl = (tw - 1) - (px + op["pad_l"] - s0) // 2
                                A, B = len(range(py, op["x"].h, 2)), len(range(px, op["x"].w, 2))
                                if th < 1 or tw < 1 or pt < 0 or pl < 0 or A < 1 or B < 1:
                                    cls = None
                                    break
                                nb_ = self.lib.gv_packed_filter_bytes(th, tw, cout, cin, self.dt, self.math_mode)
                                cls.append(dict(py=py, px=px, r0=r0, s0=s0, th=th, tw=tw, pad_t=pt, pad_l=pl, A=A, B=B, tile=0,
                                                w=torch.zeros(nb_, dtype=torch.uint8, device=dev)))
                            if cls is None:
                                break
                        if cls:
                            op["s2"] = cls
            nbv = nb
            self.r_img = torch.empty(nbv, dtype=f32, device=dev)
            self.scores = torch.empty(self.Vh, dtype=f32, device=dev)
            self.gidx = torch.empty(self.Vh, dtype=torch.int32, device=dev)
            self.scheme = torch.empty((num_group, self.Vh), dtype=torch.int32, device=dev)
            self.weight = torch.empty(num_group, dtype=f32, device=dev)
            self.status = torch.zeros(1, dtype=torch.int32, device=dev)
            if self.per_shape:
                self.scores_ps = torch.empty((num_shapes, self.Vh), dtype=f32, device=dev)
                self.gidx_ps = torch.empty((num_shapes, self.Vh), dtype=torch.int32, device=dev)
                self.scheme_ps = torch.empty((num_shapes, num_group, self.Vh), dtype=torch.int32, device=dev)
                self.weight_ps = torch.empty((num_shapes, num_group), dtype=f32, device=dev)
            f = self.final
            self.S = torch.empty((num_shapes, f.h, f.w, f.c), dtype=self.tdt, device=dev)
            self.dS = torch.empty((num_shapes, f.h, f.w, f.c), dtype=f32, device=dev)
            self.gap = torch.empty((num_shapes, f.c), dtype=f32, device=dev)
            self.dgap = torch.empty_like(self.gap)
            self.logits = torch.empty((num_shapes, num_classes), dtype=f32, device=dev)
            self.dlogits = torch.empty_like(self.logits)
            self.loss = torch.zeros(1, dtype=f32, device=dev)
        self._packed_dirty = True
        self._pack_jobs = None
        self._moving_jobs = None
        # Train-mode BatchNorm sums folded into the launch that produces the tensor (gv_conv2d_fwd_bnstats): the forward
        # sums of z in the convolution's epilogue, the backward sums of dy in the epilogue of the data-gradient launch
        # that writes the FINAL dy (16-bit storage; False: the separate sums passes, kept for A/B and tests)
        self.fuse_bn_stats = self.es == 2
        self.fuse_bn_stats_res = True                     # ... also where the producing convolution adds a residual (ResNet conv3; A/B)
        self.fuse_bn_pool = self.es == 2                  # BatchNorm -> max pool pairs of the stem as pool -> BatchNorm (A/B)
        self.alias_residual_grad = True                   # residual fan-in: the shortcut's gradient shares dy's buffer (False: copy; A/B)
        # Bias gradients that need no pass over dy (ResNet-v2; `_bias_sources`): a bias in front of a train-mode BatchNorm has
        # a zero gradient, one behind a residual add has the gradient of the bias of that add.  False: every bias summed
        self.bias_grad_identities = True
        self._bias_src = None
        self._bias_src_key = None
        self.s2_classes = True                            # stride-2 data gradients by parity classes (False: zero-dilated dZ; A/B)
        self.s2_concurrent = False                        # ... their four launches side by side on extra streams: measured
                                                          # 15.66 k against 15.89 k views/s in sequence (fork / join cost more
                                                          # than the overlap returns); kept as an A/B switch
        self._plan_bn_fusion()

    # -- BatchNorm sums folded into the producing launch --------------------------------------------------
    def _plan_bn_fusion(self):
        """Which launch produces each BatchNorm's sums.  Forward: the convolution whose output (or a channel slice of
        it: members of a fused sibling GEMM) the BatchNorm reads.  Backward: the data-gradient launch that is the LAST
        contributor to the gradient of the BatchNorm's output — the backward pass runs the ops in reverse, so that is the
        FIRST op (in forward order) that reads the tensor; it must be a convolution whose input covers the tensor (a
        pool or a residual fan-in as last contributor leaves the separate sums pass in place)."""
        ops = self.plan.ops
        inside = lambda t, u: t.vbuf == u.vbuf and u.off <= t.off and t.off + t.c <= u.off + u.c
        overlap = lambda t, u: t.vbuf == u.vbuf and t.off < u.off + u.c and u.off < t.off + t.c
        for op in ops:
            op.pop("st_f", None), op.pop("st_b", None), op.pop("fused_f", None), op.pop("fused_b", None)
        # ops that take part in a backward pass (their output receives a gradient), as backward_backbone decides it
        reached, live = [self.final], set()
        for i in range(len(ops) - 1, -1, -1):
            y = ops[i]["y"]
            if y.vbuf < 0 or not any(overlap(y, r) for r in reached):
                continue
            live.add(i)
            reached.append(ops[i]["x"])
            if ops[i].get("res") is not None:
                reached.append(ops[i]["res"])
        for bi, b in enumerate(ops):
            if b["kind"] != "bn":
                continue
            # forward: the producer of b.x
            for c in ops[:bi]:
                if c["kind"] == "conv" and b["x"].vbuf >= 0 and inside(b["x"], c["y"]) and (c.get("res") is None or self.fuse_bn_stats_res):
                    c.setdefault("st_f", []).append(b)
                    b["fused_f"] = c
            # backward: the first reader of b.y among the ops that run in the backward pass
            if bi not in live:
                continue
            readers = [(i, o) for i, o in enumerate(ops) if i in live and i > bi and
                       (overlap(o["x"], b["y"]) or (o.get("res") is not None and overlap(o["res"], b["y"])))]
            if not readers:
                continue
            if any(not inside(b["y"], o["x"]) or (o.get("res") is not None and overlap(o["res"], b["y"])) for _, o in readers):
                continue                                      # a reader of a part of it, or a residual fan-in
            first = readers[0][1]
            if first["kind"] == "conv" and first["x"].vbuf >= 0:
                first.setdefault("st_b", []).append(b)
                b["fused_b"] = first
        for op in ops:
            for key in ("st_f", "st_b"):
                if len(op.get(key, ())) > _lib.GV_BN_STATS_MAX_SEG:
                    for b in op.pop(key):
                        b.pop("fused_f" if key == "st_f" else "fused_b")
        # BatchNorm (+ReLU, no gamma: a positive scale) whose ONLY reader is a 3x3/2 VALID max pool (Conv2d_2b ->
        # MaxPool_3a, Conv2d_4a -> MaxPool_5a): the pool takes z itself, the BatchNorm is applied to the pooled tensor and
        # its backward pass is finished by the pool's backward kernel (gv_pool2d_bwd_argmax_bn) — neither the activation
        # nor its gradient exists at the un-pooled size
        same = lambda t, u: t.vbuf == u.vbuf and t.off == u.off and t.c == u.c
        for op in ops:
            op.pop("pool_after", None), op.pop("bn_before", None)
        if self.es == 2:
            for bi, b in enumerate(ops):
                if b["kind"] != "bn" or not b["relu"] or b["has_gamma"] or b["y"].vbuf < 0:
                    continue
                readers = [o for o in ops[bi + 1:] if overlap(o["x"], b["y"]) or
                           (o.get("res") is not None and overlap(o["res"], b["y"]))]
                taps = [t for t in (self.raw, self.final) if overlap(t, b["y"])]
                if len(readers) != 1 or taps:
                    continue
                p_ = readers[0]
                if (p_["kind"] == "pool" and p_["mode"] == _lib.GV_POOL_MAX and p_["k"] == 3 and p_["stride"] == 2 and
                        p_["pad_t"] == 0 and p_["pad_l"] == 0 and same(p_["x"], b["y"]) and b["x"].c % 8 == 0):
                    b["pool_after"], p_["bn_before"] = p_, b
                    c = b["x"].c
                    if "pz" not in p_:
                        p_["pz"] = torch.empty(p_["y"].npix * c, dtype=self.tdt, device=self.device)
                        b["coef"] = [torch.empty((self.V, c), dtype=torch.float32, device=self.device) for _ in range(3)]

    def _bn_stats(self, op, key):
        """The gv_bn_stats of convolution `op` (key 'st_f': forward sums of its output; 'st_b': backward sums of its
        data gradient), built once: every pointer in it is fixed for the life of the engine."""
        cache = "_" + key + "_struct"
        if cache not in op:
            fwd = key == "st_f"
            base = op["y"] if fwd else op["x"]
            st = _lib.BnStats()
            st.mode = _lib.GV_BN_STATS_FWD if fwd else _lib.GV_BN_STATS_BWD
            st.groups, st.nseg = self.V, len(op[key])
            for i, b in enumerate(op[key]):
                t = b["x"] if fwd else b["y"]
                sg = st.seg[i]
                sg.c0, sg.c1 = t.off - base.off, t.off - base.off + t.c
                sg.acc = (b["acc_f"] if fwd else b["acc_b"]).data_ptr()
                if not fwd:
                    sg.z, sg.z_ld = self._ptr(b["x"]), b["x"].ld
                    if b["relu"]:
                        sg.scale, sg.shift = b["stat"]["scale"].data_ptr(), b["stat"]["shift"].data_ptr()
            op[cache] = st
        return op[cache]

    def _fusing(self):
        return self.fuse_bn_stats and self._zacc and self._lazy and not self.frozen_bn

    def _pool_fused(self, op):
        """Is this BatchNorm / max pool executed as the fused pair (pool z, normalise the pooled tensor)?"""
        return (self.fuse_bn_pool and self._zacc and self._lazy and not self.frozen_bn and self.pool_argmax and
                (op.get("pool_after") is not None or op.get("bn_before") is not None))

    # -- helpers -----------------------------------------------------------------------------------------
    def _ptr(self, t, grad=False):
        if t.vbuf < 0:
            assert not grad
            return self._x.data_ptr() + self.es * t.off
        if grad:
            if self.grad[t.vbuf] is None:
                self.grad[t.vbuf] = torch.zeros_like(self.act[t.vbuf])
            return self.grad[t.vbuf].data_ptr() + self.es * t.off
        return self.act[t.vbuf].data_ptr() + self.es * t.off

    # -- launch lanes: the independent branches of an Inception block on separate HIP streams -------------------------
    def enable_lanes(self, n=3):
        """Run the ops of lane k > 0 (backbones.build_inception_v3 puts the branches of a block on lanes 0..2) on
        their own streams.  Ordering comes from the tensors: an op waits for the last writer of everything it reads
        or updates (events, at most one wait per source lane), so the fan-in of a block input's gradient stays a
        chain; every phase forks from and joins back into the caller's stream.  A speed choice only, and measured
        NEUTRAL at 32 shapes x 12 views (31.9 vs 32.0 ms: every kernel of the step already fills the chip); it is kept
        for small batches.  Capturing the multi-stream step into one graph crashes hipStreamEndCapture in this runtime
        (tools/lanes_capture_probe.py) — the single-stream step captures fine."""
        self._lane_streams = [None] + [torch.cuda.Stream(self.device) for _ in range(n - 1)]
        self._lane_accum = [self.accum] + [torch.zeros_like(self.accum) for _ in range(n - 1)]
        self._ready = {}
        # events come from a fixed pool (one per op and phase boundary): none is created or destroyed while a
        # stream capture is in progress
        self._ev_pool = [torch.cuda.Event() for _ in range(2 * len(self.plan.ops) + 4 * n + 8)]
        self._ev_next = 0

    def _event(self):
        ev = self._ev_pool[self._ev_next % len(self._ev_pool)]
        self._ev_next += 1
        return ev

    def _phase_begin(self):
        if self._lane_streams is None:
            return
        self._ready = {}
        self._lane_seen = [dict() for _ in self._lane_streams]
        self._ev_next = 0 if self._ev_next >= len(self._ev_pool) // 2 else len(self._ev_pool) // 2   # two halves
        ev = self._event()
        ev.record(torch.cuda.current_stream(self.device))
        for s_ in self._lane_streams[1:]:
            s_.wait_event(ev)

    def _phase_end(self):
        if self._lane_streams is None:
            return
        main = torch.cuda.current_stream(self.device)
        for s_ in self._lane_streams[1:]:
            ev = self._event()
            ev.record(s_)
            main.wait_event(ev)
        self._ready = {}

    def _on_lane(self, op, kind, touched, written, fn):
        """Run fn() on the op's lane after the last writers of `touched` (tensors read or updated); record this op
        as the last writer of `written`."""
        if self._lane_streams is None:
            return fn()
        lane = min(op.get("lane", 0), len(self._lane_streams) - 1)
        main = torch.cuda.current_stream(self.device)
        stream = main if lane == 0 else self._lane_streams[lane]
        latest = {}                                       # per source lane only its most recent event: a stream is ordered
        for t in touched:
            if t is None or t.vbuf < 0:
                continue
            for off, c, ev, ln, seq in self._ready.get((kind, t.vbuf), ()):
                if ln != lane and off < t.off + t.c and t.off < off + c and seq > latest.get(ln, (-1, None))[0]:
                    latest[ln] = (seq, ev)
        for ln, (seq, ev) in latest.items():
            if seq > self._lane_seen[lane].get(ln, -1):   # (not already ordered after it by an earlier wait)
                stream.wait_event(ev)
                self._lane_seen[lane][ln] = seq
        saved = self.accum
        self.accum = self._lane_accum[lane]
        self._cur_lane = lane
        try:
            if lane == 0:
                fn()
            else:
                with torch.cuda.stream(stream):
                    fn()
        finally:
            self.accum = saved
            self._cur_lane = 0
        ev = self._event()
        ev.record(stream)
        for t in written:
            if t is None or t.vbuf < 0:
                continue
            lst = [e for e in self._ready.get((kind, t.vbuf), []) if not (t.off <= e[0] and e[0] + e[1] <= t.off + t.c)]
            lst.append((t.off, t.c, ev, lane, self._ev_next))
            self._ready[(kind, t.vbuf)] = lst

    def _wgrad(self, d, x_ptr, dz_ptr, dz_ld, dw_ptr):
        """One filter-gradient launch (plus, deterministic, the launch that adds its slices in order)."""
        if not self.deterministic:
            return self.lib.gv_conv2d_wgrad(C.byref(d), x_ptr, dz_ptr, dz_ld, dw_ptr, _st())
        # ONE workspace per launch lane: with enable_lanes() the filter gradients of different Inception branches run on
        # different streams, and a shared workspace would let one lane's slice stores overwrite what another lane's
        # slice reduce is still reading
        lane = self._cur_lane if self._lane_streams is not None else 0
        if self._dw_ws is None:
            self._dw_ws = {}
        ws = self._dw_ws.get(lane)
        if ws is None:
            ws = self._dw_ws[lane] = torch.empty(self._dw_ws_bytes, dtype=torch.uint8, device=self.device)
        return self.lib.gv_conv2d_wgrad_ws(C.byref(d), x_ptr, dz_ptr, dz_ld, dw_ptr, ws.data_ptr(), ws.numel(), _st())

    def _members(self, op):
        """[(variable name, first column, columns)] of a convolution: one entry, or the members of a fused sibling GEMM."""
        return op.get("members") or [(op["name"] + "/weights", 0, op["y"].c)]

    def _dw(self, op):
        """Where the filter gradient of `op` is accumulated: the variable's gradient, or the fused scratch."""
        return op["dw_fused"] if op.get("members") else self.grads[op["name"] + "/weights"]

    def _claim(self, t):
        """True for the FIRST gradient contribution to tensor t in this backward pass (it must store, or zero t first);
        always False in the plain form (_lazy = False), whose gradient buffers are zero-filled up front."""
        if not self._lazy:
            return False
        key = (t.vbuf, t.off, t.c)
        first = key not in self._written
        self._written.add(key)
        return first

    def _has_grad(self, t):
        """Has the gradient of t (or of a concat tensor that contains it) been produced in this backward pass?"""
        if not self._lazy:
            return self.grad[t.vbuf] is not None
        pos = t.off                                           # union of the written channel ranges covers [off, off + c)
        for o, c in sorted((o, c) for v, o, c in self._written if v == t.vbuf):
            if o <= pos < o + c:
                pos = o + c
        return pos >= t.off + t.c

    def _can_alias_grad(self, r, y):
        """May the gradient of r share the buffer of the gradient of y (same geometry, each alone in its buffer)?"""
        return (self.alias_residual_grad and self._lazy and self._lane_streams is None and r.vbuf >= 0 and
                r.vbuf != y.vbuf and (r.nb, r.h, r.w, r.c) == (y.nb, y.h, y.w, y.c) and r.off == 0 and y.off == 0 and
                r.ld == r.c and y.ld == y.c and self.act[r.vbuf].numel() == self.act[y.vbuf].numel())

    def _bias_sources(self):
        """{bias name: None (sum dy: gv_bias_grad_t) | [] (zero) | [other bias names] (the sum of THEIR gradients)}.

        db = sum over pixels of dy, dy the gradient of the tensor the bias was added into.  That tensor's gradient is the
        sum of what its readers send back, and two kinds of reader send something whose pixel sum is known without
        looking at it: a train-mode BatchNorm (batch statistics over exactly these pixels, per view) returns
        dz_i = A (g_i - mean g - zhat_i mean(g zhat)), whose sum over i is 0 because sum zhat_i = 0 — the textbook
        redundancy of a bias in front of a BatchNorm; a residual add `y' = conv(..) + bias' + t` (resnet_v2.py:91) returns
        dy' itself, whose pixel sum IS db'; a max / average / sub-sampling pool returns a scatter of its own dy that keeps
        the sum.  In ResNet-v2-50 that leaves two biases to sum (the last units of block3 and block4, whose outputs are
        tapped) out of twenty-one: conv1 and the last conv3 of blocks 1 and 2 are zero, the other sixteen are copies.  In exact
        arithmetic these ARE the gradients (the oracle's fp64 autograd shows the zeros and the equalities to 1e-6,
        tests/test_gpu_train.py); the summed form differs from them by the rounding noise of the stored 16-bit dy.  With
        frozen statistics (`frozen_bn`) a BatchNorm's return does not sum to zero: only the residual rule applies."""
        # (cached per (switch, frozen statistics, the tapped buffers): toggling bias_grad_identities or frozen_bn after
        # the first backward pass must take effect — with statistics frozen later, a cached zero rule would be wrong)
        key = (bool(self.bias_grad_identities), bool(self.frozen_bn), self.raw.vbuf, self.final.vbuf)
        if self._bias_src is not None and self._bias_src_key == key:
            return self._bias_src
        self._bias_src_key = key
        ops = self.plan.ops
        readers = {}
        for op in ops:
            for key in ("x", "res"):
                u = op.get(key)
                if u is not None and u.vbuf >= 0:
                    readers.setdefault(u.vbuf, []).append((op, key))
        tapped = {self.raw.vbuf, self.final.vbuf}

        def terms(t, depth=0):
            """bias names whose gradients add up to the pixel sum of dt ([] = zero), or None = unknown."""
            if t.vbuf in tapped or depth > 64 or t.off != 0 or t.ld != t.c:
                return None
            out = []
            for c, key in readers.get(t.vbuf, ()):
                if c["kind"] == "bn" and key == "x":
                    if self.frozen_bn:
                        return None
                    continue
                if c["kind"] == "conv" and key == "res":
                    sub = [c["bias"]] if c.get("bias") else terms(c["y"], depth + 1)
                elif c["kind"] == "pool" and key == "x":
                    sub = terms(c["y"], depth + 1)
                else:
                    return None
                if sub is None:
                    return None
                out += sub
            return out if readers.get(t.vbuf) else None

        src = {}
        for op in ops:
            if op["kind"] == "conv" and op.get("bias"):
                src[op["bias"]] = terms(op["y"]) if self.bias_grad_identities else None
        self._bias_src = src
        return src

    def _zero_grad_of(self, t):
        self._ptr(t, grad=True)
        self.view(t, grad=True).zero_()

    def view(self, t, grad=False):
        base = self.grad[t.vbuf] if grad else self.act[t.vbuf]
        return torch.as_strided(base, (t.nb, t.h, t.w, t.c), (t.h * t.w * t.ld, t.w * t.ld, t.ld, 1), t.off)

    def _count(self, hw):
        key = (hw, self.shape_world)
        if key not in self._counts:                       # pixels of one view over the GLOBAL batch
            self._counts[key] = torch.full((self.V,), self.N * hw * self.shape_world, dtype=torch.int32,
                                           device=self.device)
        return self._counts[key]

    def _conv_desc(self, op, dgrad=False, wgrad=False):
        x, y = op["x"], op["y"]
        if wgrad:                                         # descriptor of the forward conv, filter-gradient launch choice
            return _lib.ConvDesc(x.nb, x.h, x.w, x.c, x.ld, op["kh"], op["kw"], op["stride"], op["pad_t"],
                                 op["pad_l"], y.h, y.w, y.c, y.ld, 0, 0, 0, self.dt, 0, op.get("tile_w", 0),
                                 self.math_mode, 0)
        if not dgrad:
            # the first layer of a 16-bit engine reads the fp32 images themselves (GV_CONV_X_F32: rounded by the loader to
            # the very values the 16-bit copy holds), which is what the strip kernel of the 3-channel stems takes
            xf32 = _lib.GV_CONV_X_F32 if (x.vbuf < 0 and self.es == 2) else 0
            return _lib.ConvDesc(x.nb, x.h, x.w, x.c, x.ld, op["kh"], op["kw"], op["stride"], op["pad_t"],
                                 op["pad_l"], y.h, y.w, y.c, y.ld, op["res"].ld if op["res"] is not None else 0,
                                 0, xf32, self.dt, 0, op.get("tile_f", 0), self.math_mode, 0)
        # data gradient: dX = conv(dilate(dZ, stride), flip(W)^T), pad' = k-1-pad, accumulate into dX
        return _lib.ConvDesc(y.nb, y.h, y.w, y.c, y.ld, op["kh"], op["kw"], 1, op["kh"] - 1 - op["pad_t"],
                             op["kw"] - 1 - op["pad_l"], x.h, x.w, x.c, x.ld, x.ld, 0, 0, self.dt, 0,
                             op.get("tile_d", 0), self.math_mode, op["stride"] if op["stride"] > 1 else 0)

    def _s2_streams(self):
        if getattr(self, "_s2_side", None) is None:
            self._s2_side = [torch.cuda.Stream(self.device) for _ in range(3)]
            self._s2_events = [torch.cuda.Event() for _ in range(4)]
        return self._s2_side

    def _s2_class(self, op, ci, c_, dz, dx, store, fuse):
        """One parity-class launch of a stride-2 layer's data gradient (with or without the folded BatchNorm sums)."""
        lib = self.lib
        dc = self._conv_desc_s2(op, c_, not store)
        rc = _lib.GV_E_UNSUPPORTED
        if fuse and op.get("_s2_stats_ok", True):
            rc = lib.gv_conv2d_fwd_bnstats(C.byref(dc), dz, c_["w"].data_ptr(), self.ones.data_ptr(),
                                           self.zeros.data_ptr(), None if store else dx, dx,
                                           C.byref(self._bn_stats(op, "st_b")), _st())
            if rc == _lib.GV_E_UNSUPPORTED:       # this class' tile cannot fold the sums: nobody does (from the next step on)
                op["_s2_stats_ok"] = False
        if rc == _lib.GV_E_UNSUPPORTED:
            rc = lib.gv_conv2d_fwd(C.byref(dc), dz, c_["w"].data_ptr(), self.ones.data_ptr(), self.zeros.data_ptr(),
                                   None if store else dx, dx, None, None, None, _st())
        _lib.check(rc, "dgrad (parity class %d) %s" % (ci, op["name"]))

    def _conv_desc_s2(self, op, c_, accumulate):
        """Descriptor of ONE parity class of a stride-2 layer's data gradient: a stride-1 convolution over dZ whose
        output pixel (a, b) is dX pixel (2a + py, 2b + px)."""
        x, y = op["x"], op["y"]
        return _lib.ConvDesc(y.nb, y.h, y.w, y.c, y.ld, c_["th"], c_["tw"], 1, c_["pad_t"], c_["pad_l"], c_["A"], c_["B"],
                             x.c, x.ld, x.ld if accumulate else 0, 0, 0, self.dt, 0, c_["tile"], self.math_mode, 0, 0,
                             2, c_["py"], c_["px"], x.h, x.w)

    def autotune(self, iters=2):
        """Measure, per convolution, the fastest tile configuration of the forward launch and of the data-gradient
        launch (hipEvents on the launch stream, this engine's own buffers; gradients buffers are scratch here).
        A speed choice; values move at fp32-rounding level when the winner belongs to another kernel family (another k
        summation order: include/gvcnn_hip.h, tile_cfg) — bit reproducibility ACROSS processes needs the same table
        installed, inside one process the step repeats bitwise.  forward() must have run once (buffers and packed
        filters exist)."""
        lib = self.lib
        if self._packed_dirty:
            self.repack()
        ncfg = lib.gv_conv2d_num_tile_cfgs(self.math_mode if self.dt == _lib.GV_F32 else -1)
        ms = C.c_float(0)
        if self._fusing():
            # what the SEPARATE sums passes of every fusable BatchNorm cost: a tile that cannot fold the sums competes as
            # plain launch + these
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for op in self.plan.ops:
                op.pop("_sums_ms_f", None), op.pop("_sums_ms_b", None), op.pop("_nofuse_f", None), op.pop("_nofuse_b", None)
            for b in self.plan.ops:
                if b["kind"] != "bn":
                    continue
                x, y, st, hw = b["x"], b["y"], b["stat"], b["x"].h * b["x"].w
                calls = []
                if b.get("fused_f") is not None:
                    calls.append(("_sums_ms_f", b["fused_f"], lambda: lib.gv_bn_sums_grouped_t(
                        self._ptr(x), x.nb, hw, x.c, x.ld, self.V, self.accum.data_ptr(), self.dt, _st())))
                if b.get("fused_b") is not None:
                    sc = st["scale"].data_ptr() if b["relu"] else None
                    sh = st["shift"].data_ptr() if b["relu"] else None
                    calls.append(("_sums_ms_b", b["fused_b"], lambda: lib.gv_bn_relu_bwd_sums_grouped_t(
                        self._ptr(y, True), y.ld, None, y.ld, self._ptr(x), x.ld, st["mean"].data_ptr(), st["inv"].data_ptr(),
                        x.nb, hw, x.c, self.V, self.accum.data_ptr(), sc, sh, self.dt, _st())))
                for key, conv, fn in calls:
                    if fn() != 0:
                        continue
                    ev0.record()
                    for _ in range(iters):
                        fn()
                    ev1.record()
                    ev1.synchronize()
                    conv[key] = conv.get(key, 0.0) + ev0.elapsed_time(ev1) / iters
        for op in self.plan.ops:
            if op["kind"] != "conv":
                continue
            x, y = op["x"], op["y"]
            jobs = [("tile_f", False, self._x32.data_ptr() + 4 * x.off if (x.vbuf < 0 and self.es == 2) else self._ptr(x),
                     op["w_fwd"], self._ptr(y))]
            if x.vbuf >= 0:
                jobs.append(("tile_d", True, self._ptr(y, True), op["w_dgrad"], self._ptr(x, True)))
            for key, dgrad, src, w, dst in jobs:
                op[key] = 0
                best, best_ms = 0, float("inf")
                # a launch that also produces BatchNorm sums is timed AS that launch (its epilogue differs; a tile that
                # cannot fold them is timed plain plus nothing: the separate sums pass then costs what it costs)
                skey = "st_b" if dgrad else "st_f"
                fused = bool(op.get(skey)) and self._fusing()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best_fused = False
                for t in range(ncfg):
                    op[key] = t + 1
                    d = self._conv_desc(op, dgrad=dgrad)
                    d.res_ld = 0
                    rc = -1
                    if fused:
                        args = (C.byref(d), src, w.data_ptr(), self.ones.data_ptr(), self.zeros.data_ptr(), None, dst,
                                C.byref(self._bn_stats(op, skey)), _st())
                        rc = lib.gv_conv2d_fwd_bnstats(*args)
                        if rc == 0:
                            e0.record()
                            for _ in range(iters):
                                lib.gv_conv2d_fwd_bnstats(*args)
                            e1.record()
                            e1.synchronize()
                            t_ms = e0.elapsed_time(e1) / iters
                            if t_ms < best_ms:
                                best, best_ms, best_fused = t + 1, t_ms, True
                    # the plain launch of the same tile — of a fusable layer too: plus what its separate sums pass costs
                    # (measured once).  A folding epilogue can cost more than the pass it saves (Conv2d_1a's strip kernel:
                    # 0.27 ms folded against 0.12 + 0.08), so the layer then keeps the separate pass (op["_nofuse_*"]).
                    rc = lib.gv_conv2d_time(C.byref(d), src, w.data_ptr(), self.ones.data_ptr(), self.zeros.data_ptr(),
                                            dst, iters, C.byref(ms), _st())
                    extra = op.get("_sums_ms_b" if dgrad else "_sums_ms_f", 0.0) if fused else 0.0
                    if rc == 0 and ms.value + extra < best_ms:
                        best, best_ms, best_fused = t + 1, ms.value + extra, False
                op[key] = best
                op["_" + key + "_ms"] = best_ms
                if fused:
                    op["_nofuse_b" if dgrad else "_nofuse_f"] = not best_fused
            # the parity classes of a stride-2 layer's data gradient: one tile choice each
            for c_ in op.get("s2", ()) if self.s2_classes else ():
                fused = bool(op.get("st_b")) and self._fusing()
                best, best_ms = 0, float("inf")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for t in range(ncfg):
                    c_["tile"] = t + 1
                    dc = self._conv_desc_s2(op, c_, False)
                    src, dst = self._ptr(y, True), self._ptr(x, True)
                    if fused:
                        args = (C.byref(dc), src, c_["w"].data_ptr(), self.ones.data_ptr(), self.zeros.data_ptr(), None, dst,
                                C.byref(self._bn_stats(op, "st_b")), _st())
                        if lib.gv_conv2d_fwd_bnstats(*args) == 0:
                            e0.record()
                            for _ in range(iters):
                                lib.gv_conv2d_fwd_bnstats(*args)
                            e1.record()
                            e1.synchronize()
                            if e0.elapsed_time(e1) / iters < best_ms:
                                best, best_ms = t + 1, e0.elapsed_time(e1) / iters
                        continue                              # (classes must all fold, or none: only folding tiles compete)
                    rc = lib.gv_conv2d_time(C.byref(dc), src, c_["w"].data_ptr(), self.ones.data_ptr(),
                                            self.zeros.data_ptr(), dst, iters, C.byref(ms), _st())
                    if rc == 0 and ms.value < best_ms:
                        best, best_ms = t + 1, ms.value
                c_["tile"] = best
                c_["ms"] = best_ms
            if op.get("s2") and self.s2_classes:              # four class launches or the one zero-dilated launch?
                op["_s2_use"] = sum(c_["ms"] for c_ in op["s2"]) < op.get("_tile_d_ms", float("inf"))
            # filter gradient (16-bit storage): tile and pixel-split choice, timed with events on the launch stream
            nw = lib.gv_conv2d_wgrad_num_cfgs(self.dt)
            if nw:
                dw = torch.empty_like(self._dw(op))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best, best_ms = 0, float("inf")
                for t in range(nw + 1):
                    op["tile_w"] = t
                    d = self._conv_desc(op, wgrad=True)
                    args = (d, self._ptr(x), self._ptr(y, True), y.ld, dw.data_ptr())
                    if self._wgrad(*args) != 0:
                        continue
                    e0.record()
                    for _ in range(iters):
                        self._wgrad(*args)
                    e1.record()
                    e1.synchronize()
                    if e0.elapsed_time(e1) < best_ms:
                        best, best_ms = t, e0.elapsed_time(e1)
                op["tile_w"] = best

    def repack(self, sync=True):
        """Refresh the packed filters from the trainable HWIO variables (after an optimizer step)."""
        lib = self.lib
        if self.es == 2:                                  # 16-bit storage: every filter, both forms, ONE launch
            if self._pack_jobs is None:
                jobs, blocks = [], []
                for op in self.plan.ops:
                    if op["kind"] != "conv":
                        continue
                    kh, kw, cin, total = op["kh"], op["kw"], op["x"].c, op["y"].c
                    kpad_f = (kh * kw * cin + 31) // 32 * 32          # row length of the forward image
                    fused = bool(op.get("members"))
                    for wname, col, cout in self._members(op):
                        w = self.params[wname]
                        assert tuple(w.shape) == (kh, kw, cin, cout) and w.stride(3) == 1
                        w_ld = w.stride(2) if fused else 0            # members of a fused block: row stride = its width
                        for flipped, dst in ((0, op["w_fwd"]), (1, op["w_dgrad"])):
                            if dst is None:
                                continue
                            rows, k = (cin, kh * kw * cout) if flipped else (cout, kh * kw * cin)
                            nblk = (rows * ((k + 31) // 32 * 32) + 255) // 256
                            # forward image: member rows [col, col + cout) are contiguous; data-gradient image of a
                            # fused filter: member columns inside rows of kh*kw*total (k_off / k_total)
                            out = dst.data_ptr() + (0 if flipped else col * kpad_f * 2)
                            jobs.append(_lib.PackJob(w.data_ptr(), out, kh, kw, cin, cout, flipped, len(blocks),
                                                     col if flipped and fused else 0, total if flipped and fused else 0,
                                                     w_ld))
                            blocks.extend([len(jobs) - 1] * nblk)
                        for c_ in op.get("s2", ()):                   # the taps of one parity class, flipped
                            kc = c_["th"] * c_["tw"] * cout
                            nblk = (cin * ((kc + 31) // 32 * 32) + 255) // 256
                            jobs.append(_lib.PackJob(w.data_ptr(), c_["w"].data_ptr(), c_["th"], c_["tw"], cin, cout, 1,
                                                     len(blocks), 0, 0, 0, 2, c_["r0"], c_["s0"], kw))
                            blocks.extend([len(jobs) - 1] * nblk)
                raw = b"".join(bytes(j) for j in jobs)
                self._pack_jobs = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device), len(jobs),
                                   torch.tensor(blocks, dtype=torch.int32, device=self.device))
            jd, nj, bj = self._pack_jobs
            _lib.check(lib.gv_pack_filters_batched(jd.data_ptr(), nj, bj.data_ptr(), bj.numel(), self.dt, _st()),
                       "gv_pack_filters_batched")
            if sync:
                torch.cuda.synchronize(self.device)
            self._packed_dirty = False
            return
        keep = []
        for op in self.plan.ops:
            if op["kind"] != "conv":
                continue
            w = op["w_fused"] if op.get("members") else self.params[op["name"] + "/weights"]
            kh, kw, cin, cout = w.shape
            _lib.check(lib.gv_pack_filter_hwio(w.data_ptr(), kh, kw, cin, cout, op["w_fwd"].data_ptr(),
                                               self.dt, self.math_mode, _st()), "gv_pack_filter_hwio")
            if op["w_dgrad"] is not None:
                wt = torch.flip(w, (0, 1)).permute(0, 1, 3, 2).contiguous()      # [kh,kw,cout,cin]
                keep.append(wt)
                _lib.check(lib.gv_pack_filter_hwio(wt.data_ptr(), kh, kw, cout, cin, op["w_dgrad"].data_ptr(),
                                                   self.dt, self.math_mode, _st()), "gv_pack_filter_hwio")
        if sync:                                          # (the flipped copies in `keep` die with this frame)
            torch.cuda.synchronize(self.device)
        self._packed_dirty = False

    # -- forward (train mode) ----------------------------------------------------------------------------
    def forward(self, views, labels=None, g_scheme=None, g_weight=None, check=True):
        """Phase 1 + phase 2 of train.py:264-288.  Returns (scores [V], shape_descriptor, logits, loss)."""
        assert self.Vh == self.V, "a view-sharded engine is driven by sharding.ShardedTrainGVCNN"
        self.forward_backbone(views)
        self.score_partial()
        return self.forward_head(labels, g_scheme, g_weight, check)

    def forward_backbone(self, views):
        """Train-mode backbone over this engine's views (partial_run #1 without the scorer)."""
        lib = self.lib
        assert tuple(views.shape) == (self.N, self.V, self.H, self.W, 3) and views.is_cuda
        self._x = views.to(self.tdt).contiguous()
        self._x32 = views.float().contiguous()            # (what the first convolution's forward launch reads)
        if self._packed_dirty:
            self.repack()
        if self._zacc:
            self._accum_f.zero_()                         # every layer's forward sums: one fill
        self._phase_begin()
        self._run_pass(list(self.plan.ops), forward=True)
        self._phase_end()

    # -- pass loops: BatchNorm layers that become ready together share ONE statistics all-reduce ------------------------
    def _bn_sync_call(self, t):
        self.bn_sync_calls += 1
        self.bn_sync(t)

    def _run_pass(self, ops, forward, progress=None):
        """Run `ops` in order (forward ops, or the backward ops of the reversed list).  With a statistics exchange
        (bn_sync: shape-sharded / hybrid data parallelism) a run of CONSECUTIVE BatchNorm ops none of which reads what
        another one of the run writes — the member BatchNorms of a fused sibling GEMM — executes as: every member's
        sums, ONE all-reduce over their (adjacent) accumulators, every member's apply; one collective instead of one per
        layer and direction."""
        fn = self._forward_op if forward else self._backward_op
        kind = "a" if forward else "g"
        coalesce = (self.bn_sync is not None and self._zacc and not self.frozen_bn and self._lane_streams is None
                    and self.coalesce_bn_sync)
        overlap = lambda t, u: t.vbuf == u.vbuf and t.off < u.off + u.c and u.off < t.off + t.c
        i = 0
        while i < len(ops):
            op = ops[i]
            if not forward and (op["y"].vbuf < 0 or not self._has_grad(op["y"])):
                i += 1
                continue                                  # nothing downstream of the final tap reaches it
            run = [op]
            if coalesce and op["kind"] == "bn":
                j = i + 1
                while j < len(ops) and ops[j]["kind"] == "bn":
                    o = ops[j]
                    if not forward and (o["y"].vbuf < 0 or not self._has_grad(o["y"])):
                        break
                    # (forward: o reads x, the run wrote its y's; backward: o reads dy, the run wrote its dx's)
                    if any(overlap(o["x"] if forward else o["y"], r["y"] if forward else r["x"]) for r in run):
                        break
                    run.append(o)
                    j += 1
            if len(run) == 1:
                if forward:
                    self._on_lane(op, kind, (op["x"], op.get("res")), (op["y"],), lambda: fn(op, self._zacc))
                else:
                    outs = (op["x"], op.get("res"))
                    self._on_lane(op, kind, (op["y"],) + outs, outs, lambda: fn(op, self._zacc))
                    if progress is not None and op["kind"] == "conv" and op.get("g_lo") is not None:
                        progress(op["g_lo"])
                i += 1
                continue
            for o in run:
                fn(o, self._zacc, "sums")
            key = "acc_f" if forward else "acc_b"
            base = self._accum_f if forward else self._accum_b
            lo = min(o[key].storage_offset() for o in run)
            hi = max(o[key].storage_offset() + o[key].numel() for o in run)
            self._bn_sync_call(base[lo:hi])               # (the members' accumulators are adjacent: allocated in op order)
            for o in run:
                fn(o, self._zacc, "apply")
            i += len(run)

    def _forward_op(self, op, zeroed=False, part="all"):
        """One op of the train-mode forward pass (conv -> z, BatchNorm on batch statistics (+ReLU), pool).
        zeroed: the op's own fp64 accumulator was zero-filled by the caller (the pass loops do that for all layers at
        once); otherwise the sums call clears it itself, so calling an op on its own is always safe.
        part (BatchNorm only): "sums" = the statistics pass alone, "apply" = finalize + apply alone — the pass loops use
        the two halves to all-reduce the sums of several layers in ONE collective (shape-sharded training)."""
        lib, V = self.lib, self.V
        x, y = op["x"], op["y"]
        if op["kind"] == "conv":
            d = self._conv_desc(op)
            shift = self.params[op["bias"]] if op["bias"] else self.zeros
            res = op["res"]
            op["_st_f_done"] = False
            xin = self._x32.data_ptr() + 4 * x.off if (x.vbuf < 0 and self.es == 2) else self._ptr(x)
            if op.get("st_f") and zeroed and self._fusing() and not op.get("_nofuse_f"):   # the BatchNorm sums of z in this launch's epilogue
                # (a ResNet unit's conv3: the sums are those of shortcut + residual, the tensor the next pre-activation reads)
                rc = lib.gv_conv2d_fwd_bnstats(C.byref(d), xin, op["w_fwd"].data_ptr(), self.ones.data_ptr(),
                                               shift.data_ptr(), self._ptr(res) if res is not None else None, self._ptr(y),
                                               C.byref(self._bn_stats(op, "st_f")), _st())
                if rc != _lib.GV_E_UNSUPPORTED:               # (unsupported tile / geometry: the plain launch below)
                    _lib.check(rc, "conv + BN sums " + op["name"])
                    op["_st_f_done"] = True
                    return
            _lib.check(lib.gv_conv2d_fwd(C.byref(d), xin, op["w_fwd"].data_ptr(), self.ones.data_ptr(),
                                         shift.data_ptr(), self._ptr(res) if res is not None else None,
                                         self._ptr(y), None, None, None, _st()), "conv " + op["name"])
        elif op["kind"] == "bn":
            st = op["stat"]
            gamma = self.params[op["name"] + "/gamma"] if op["has_gamma"] else None
            beta = self.params[op["name"] + "/beta"]
            hw = x.h * x.w
            acc = op["acc_f"] if self._zacc else self.accum
            zf = _lib.GV_ACCUM_ZEROED if zeroed else 0
            if part == "apply":
                pass
            elif self.frozen_bn:                          # sums that finalize to (moving_mean, moving_variance)
                mm = self.params[op["name"] + "/moving_mean"].double()
                mv = self.params[op["name"] + "/moving_variance"].double()
                cnt = self._count(hw).double().view(V, 1)
                a = acc[:2 * V * x.c].view(V, x.c, 2)
                a[..., 0] = cnt * mm
                a[..., 1] = cnt * (mv + mm * mm)
            elif not (zeroed and op.get("fused_f") is not None and op["fused_f"].get("_st_f_done")):
                _lib.check(lib.gv_bn_sums_grouped_t(self._ptr(x), x.nb, hw, x.c, x.ld, V, acc.data_ptr(), self.dt | zf,
                                                    _st()), "bn sums " + op["name"])
            if part == "sums":
                return
            if part == "all" and self.bn_sync is not None and not self.frozen_bn:   # shape-sharded: reduce the sums first
                self._bn_sync_call(acc[:2 * V * x.c])
            if self._pool_fused(op):                      # normalised after the max pool that follows (see the pool op)
                return
            _lib.check(lib.gv_bn_finalize_apply_grouped_t(
                acc.data_ptr(), self._count(hw).data_ptr(), gamma.data_ptr() if gamma is not None else None,
                beta.data_ptr(), float(op["eps"]), self._ptr(x), x.nb, hw, x.c, x.ld, V, int(op["relu"]), self._ptr(y),
                y.ld, st["mean"].data_ptr(), st["var"].data_ptr(), st["inv"].data_ptr(), st["scale"].data_ptr(),
                st["shift"].data_ptr(), self.dt, _st()), "bn finalize + apply " + op["name"])
        else:
            d = _lib.PoolDesc(x.nb, x.h, x.w, x.c, x.ld, op["k"], op["k"], op["stride"], op["pad_t"],
                              op["pad_l"], y.h, y.w, y.ld, op["mode"], self.dt)
            if self._pool_fused(op):
                # pool z (BN + ReLU with a positive scale are monotone: same winner, same value after normalising), then
                # BatchNorm + ReLU on the POOLED tensor with the statistics of the whole z
                b = op["bn_before"]
                bx, st = b["x"], b["stat"]
                dz_ = _lib.PoolDesc(bx.nb, bx.h, bx.w, bx.c, bx.ld, 3, 3, 2, 0, 0, y.h, y.w, bx.c, _lib.GV_POOL_MAX, self.dt)
                if "argmax" not in op:
                    op["argmax"] = torch.empty(y.nb * y.h * y.w * y.c, dtype=torch.uint8, device=self.device)
                _lib.check(lib.gv_pool2d_fwd_argmax(C.byref(dz_), self._ptr(bx), op["pz"].data_ptr(), op["argmax"].data_ptr(),
                                                    _st()), "pool z + argmax " + op["name"])
                accf = b["acc_f"] if self._zacc else self.accum
                _lib.check(lib.gv_bn_finalize_apply_grouped_t(
                    accf.data_ptr(), self._count(bx.h * bx.w).data_ptr(), None, self.params[b["name"] + "/beta"].data_ptr(),
                    float(b["eps"]), op["pz"].data_ptr(), y.nb, y.h * y.w, bx.c, bx.c, V, 1, self._ptr(y), y.ld,
                    st["mean"].data_ptr(), st["var"].data_ptr(), st["inv"].data_ptr(), st["scale"].data_ptr(),
                    st["shift"].data_ptr(), self.dt, _st()), "bn finalize + apply (pooled) " + b["name"])
                return
            if op["mode"] == _lib.GV_POOL_MAX and self.pool_argmax:
                # the winning tap of every window is recorded (one byte per output element): the backward pass routes dy
                # by it and never re-reads x
                if "argmax" not in op:
                    op["argmax"] = torch.empty(y.nb * y.h * y.w * y.c, dtype=torch.uint8, device=self.device)
                _lib.check(lib.gv_pool2d_fwd_argmax(C.byref(d), self._ptr(x), self._ptr(y), op["argmax"].data_ptr(), _st()),
                           "pool+argmax " + op["name"])
            else:
                _lib.check(lib.gv_pool2d_fwd(C.byref(d), self._ptr(x), self._ptr(y), _st()), "pool " + op["name"])

    def score_partial(self):
        """Scorer responses r_img [N*V] of this engine's views (model.py:144-145); no gradient flows through the
        scorer: the scores leave the graph in partial_run #1."""
        r = self.raw
        _lib.check(self.lib.gv_view_score_partial(self._ptr(r), r.nb, r.h * r.w, r.c, r.ld,
                                                  self.score_kernel.data_ptr(), self.score_bias.data_ptr(), self.V,
                                                  _lib.GV_ORDER_SHAPE_MAJOR, self.r_img.data_ptr(), self.dt,
                                                  _st()), "score")
        return self.r_img

    def forward_head(self, labels=None, g_scheme=None, g_weight=None, check=True, F=None, r_img=None,
                     scores_ready=False):
        """Scores -> scheme/weight -> view pooling + fusion -> classifier -> loss, over self.Vh views.
        F [N, Vh, h, w, C] / r_img [N*Vh]: the gathered descriptors / scorer responses of a view-sharded job
        (default: this engine's own taps)."""
        lib, V = self.lib, self.Vh
        r_img = self.r_img if r_img is None else r_img
        assert r_img.numel() == self.N * V
        self._F = F
        f = self.final
        E = f.h * f.w * f.c
        F_ptr = self._ptr(f) if F is None else F.data_ptr()
        if self.per_shape:
            assert g_scheme is None, "per-shape grouping derives its schemes on the device"
            _lib.check(lib.gv_view_score_per_shape(r_img.data_ptr(), self.N * V, self.scores_ps.data_ptr(), _st()),
                       "score per shape")
            _lib.check(lib.gv_group_assign_per_shape(self.scores_ps.data_ptr(), self.N, V, self.G, self.num_bins,
                                                     self.weight_mode, self.gidx_ps.data_ptr(),
                                                     self.scheme_ps.data_ptr(), self.weight_ps.data_ptr(),
                                                     self.status.data_ptr(), _st()), "assign per shape")
            if check:
                st_ = int(self.status.item())
                if st_:
                    _raise_for_status(st_, self.gidx_ps.reshape(-1), self.G)
            _lib.check(lib.gv_view_pool_fuse_fwd_per_shape(F_ptr, V, self.N, E, E, V * E, self.scheme_ps.data_ptr(),
                                                           self.G, self.weight_ps.data_ptr(), self.pool_mode,
                                                           self.empty_fill, None, self.S.data_ptr(), self.dt,
                                                           _st()), "pool_fuse per shape")
            return self._classify_and_loss(labels, self.scores_ps)
        if not scores_ready:                              # (shape-sharded jobs finalise the scores over the global batch)
            _lib.check(lib.gv_view_score_finalize(r_img.data_ptr(), self.N, V, _lib.GV_ORDER_SHAPE_MAJOR,
                                                  self.scores.data_ptr(), _st()), "score finalize")
        if g_scheme is None:
            _lib.check(lib.gv_group_assign(self.scores.data_ptr(), V, self.G, self.num_bins, self.gidx.data_ptr(),
                                           self.scheme.data_ptr(), self.weight.data_ptr(), self.status.data_ptr(),
                                           _st()), "assign")
            if check:
                _raise_for_status(int(self.status.item()), self.gidx, self.G)
        else:
            self.scheme.copy_(torch.as_tensor(np.asarray(g_scheme), dtype=torch.int32))
            self.weight.copy_(torch.as_tensor(np.asarray(g_weight), dtype=torch.float32))
        _lib.check(lib.gv_view_pool_fuse_fwd(F_ptr, V, self.N, E, E, V * E, self.scheme.data_ptr(), self.G,
                                             self.weight.data_ptr(), self.pool_mode, self.empty_fill, None,
                                             self.S.data_ptr(), self.dt, _st()), "pool_fuse")
        return self._classify_and_loss(labels, self.scores)

    def _classify_and_loss(self, labels, scores):
        lib, f = self.lib, self.final
        _lib.check(lib.gv_global_avg_pool(self.S.data_ptr(), self.N, f.h * f.w, f.c, f.c, self.gap.data_ptr(),
                                          self.dt, _st()), "gap")
        kn, bn = self.cls_names
        _lib.check(lib.gv_dense_fwd(self.gap.data_ptr(), self.N, f.c, self.params[kn].data_ptr(),
                                    self.params[bn].data_ptr(), self.num_classes, self.logits.data_ptr(), _st()),
                   "dense")
        if labels is not None:
            self._labels = labels.to(device=self.device, dtype=torch.int64).contiguous()
            _lib.check(lib.gv_softmax_ce(self.logits.data_ptr(), self._labels.data_ptr(), self.N, self.num_classes,
                                         self.loss.data_ptr(), self.dlogits.data_ptr(), _st()), "softmax_ce")
            if self.shape_world > 1:                      # the loss is the mean over the GLOBAL batch
                _lib.check(lib.gv_scale(self.dlogits.data_ptr(), self.dlogits.numel(), 1.0 / self.shape_world, _st()),
                           "gv_scale")
        return scores, self.S, self.logits, self.loss

    # -- backward ----------------------------------------------------------------------------------------
    def backward(self):
        """Gradients of the mean CE loss (forward(labels=...) must have run).  Fills self.grads."""
        self.backward_head()
        return self.backward_backbone()

    def backward_head(self, dF=None):
        """Classifier, GAP, group fusion and view pooling backward.  The descriptor gradient is accumulated into
        this engine's final tap, or (view-sharded job) into the zeroed tensor dF [N, Vh, h, w, C]."""
        lib, V = self.lib, self.Vh
        self._flat_g.zero_()
        self._written = set()
        if not self._lazy:
            for g in self.grad:
                if g is not None:
                    g.zero_()
        elif dF is None and self._claim(self.final):
            self._zero_grad_of(self.final)                    # the head adds into the final tap's gradient
        f = self.final
        E = f.h * f.w * f.c
        kn, bn = self.cls_names
        _lib.check(lib.gv_dense_bwd(self.gap.data_ptr(), self.dlogits.data_ptr(), self.params[kn].data_ptr(), self.N,
                                    f.c, self.num_classes, self.dgap.data_ptr(), self.grads[kn].data_ptr(),
                                    self.grads[bn].data_ptr(), _st()), "dense_bwd")
        self.dS.zero_()
        _lib.check(lib.gv_global_avg_pool_bwd(self.dgap.data_ptr(), self.N, f.h * f.w, f.c, self.dS.data_ptr(), f.c,
                                              _st()), "gap_bwd")
        F_ptr = self._ptr(f) if self._F is None else self._F.data_ptr()
        dF_ptr = self._ptr(f, grad=True) if dF is None else dF.data_ptr()
        scheme, weight = (self.scheme_ps, self.weight_ps) if self.per_shape else (self.scheme, self.weight)
        _lib.check(lib.gv_view_pool_fuse_bwd_t(F_ptr, self.dS.data_ptr(), V, self.N, E, E, V * E, scheme.data_ptr(),
                                               self.G, weight.data_ptr(), self.pool_mode, dF_ptr, int(self.per_shape),
                                               self.dt, _st()), "pool_fuse_bwd")

    def final_grad(self):
        """[N, V, h, w, C] view of the gradient buffer of the final tap (allocated on first use)."""
        f = self.final
        self._ptr(f, grad=True)
        self._claim(f)                                        # (the caller writes all of it)
        return self.view(f, grad=True).view(self.N, self.V, f.h, f.w, f.c)

    def backward_backbone(self, progress=None):
        """Backbone backward from the gradient held in the final tap's gradient buffer.
        progress(lo): called after a convolution's backward has been ENQUEUED with the flat-buffer offset from which on
        every filter gradient is final, i.e. self._flat_g[lo:self._n_wd] will not be written again in this pass (the
        filters are laid out in layer order and the pass runs from the last layer to the first) — the hook a data-parallel
        wrapper uses to start reducing gradients while the rest of the backward pass still runs.  Only on the single
        launch stream (with branch lanes the enqueue order is not the execution order)."""
        if self._zacc:
            self._accum_b.zero_()                         # every layer's backward sums: one fill
        if progress is not None and (self._lane_streams is not None or not self._g_monotone):
            progress = None
        self._phase_begin()
        self._run_pass(list(reversed(self.plan.ops)), forward=False, progress=progress)
        self._phase_end()
        return self.grads

    def _backward_op(self, op, zeroed=False, part="all"):
        """Backward of one op: reads the gradient of its output, ACCUMULATES into the gradient of its input(s) and
        of its variables.  part (BatchNorm only): "sums" / "apply", see _forward_op."""
        lib, V = self.lib, self.V
        x, y = op["x"], op["y"]
        if op["kind"] == "bn":
            st = op["stat"]
            hw = x.h * x.w
            gamma = self.params[op["name"] + "/gamma"] if op["has_gamma"] else None
            dbeta = self.grads[op["name"] + "/beta"].data_ptr()
            dgamma = self.grads[op["name"] + "/gamma"].data_ptr() if gamma is not None else None
            yptr = self._ptr(y) if op["relu"] and not self._lazy else None
            sc = st["scale"].data_ptr() if op["relu"] and self._lazy else None      # mask from z*scale + shift > 0
            sh = st["shift"].data_ptr() if op["relu"] and self._lazy else None
            accb = op["acc_b"] if self._zacc else self.accum
            zf = _lib.GV_ACCUM_ZEROED if zeroed else 0
            if self._pool_fused(op):
                # BatchNorm -> max pool pair: only a window's winner carries a gradient, so the sums run over the POOLED
                # tensors (the pool's output gradient, the pooled z); the pool's backward kernel then gathers every input
                # pixel's gradient, masks it and finishes dz = A*g + B*z + C
                p_ = op["pool_after"]
                py = p_["y"]
                if part != "apply":
                    _lib.check(lib.gv_bn_relu_bwd_sums_grouped_t(
                        self._ptr(py, True), py.ld, None, py.ld, p_["pz"].data_ptr(), x.c, st["mean"].data_ptr(),
                        st["inv"].data_ptr(), py.nb, py.h * py.w, x.c, V, accb.data_ptr(), st["scale"].data_ptr(),
                        st["shift"].data_ptr(), self.dt | zf, _st()), "bn_bwd sums (pooled) " + op["name"])
                if part == "sums":
                    return
                if part == "all" and self.bn_sync is not None:
                    self._bn_sync_call(accb[:2 * V * x.c])
                assert self._claim(x), "the BatchNorm input of a fused pair has one reader"
                ca, cb, cc = op["coef"]
                _lib.check(lib.gv_bn_bwd_coeffs_t(accb.data_ptr(), self._count(hw).data_ptr(), st["mean"].data_ptr(),
                                                  st["inv"].data_ptr(), None, x.c, V, 0, ca.data_ptr(), cb.data_ptr(),
                                                  cc.data_ptr(), dbeta, None, _st()), "bn_bwd coefficients " + op["name"])
                dp = _lib.PoolDesc(x.nb, x.h, x.w, x.c, x.ld, 3, 3, 2, 0, 0, py.h, py.w, py.ld, _lib.GV_POOL_MAX, self.dt)
                _lib.check(lib.gv_pool2d_bwd_argmax_bn(C.byref(dp), p_["argmax"].data_ptr(), self._ptr(py, True), py.ld,
                                                       self._ptr(x), x.ld, V, ca.data_ptr(), cb.data_ptr(), cc.data_ptr(),
                                                       st["scale"].data_ptr(), st["shift"].data_ptr(), self._ptr(x, True),
                                                       x.ld, _st()), "pool_bwd + bn_bwd apply " + op["name"])
                return
            # (sum g, sum g*z) already added by the data-gradient launch that wrote the final dy?
            raw = zeroed and op.get("fused_b") is not None and op["fused_b"].get("_st_b_done", False)
            if not raw and part != "apply":
                _lib.check(lib.gv_bn_relu_bwd_sums_grouped_t(
                    self._ptr(y, True), y.ld, yptr, y.ld, self._ptr(x), x.ld, st["mean"].data_ptr(),
                    st["inv"].data_ptr(), x.nb, hw, x.c, V, accb.data_ptr(), sc, sh, self.dt | zf, _st()),
                    "bn_bwd sums " + op["name"])
            if part == "sums":
                return
            acc = 0 if self._claim(x) else 1
            if part == "all" and self.bn_sync is not None:
                self._bn_sync_call(accb[:2 * V * x.c])
            if self.frozen_bn:                            # beta / gamma take their gradients, the statistics terms vanish
                ab = accb[:2 * V * x.c].view(V, x.c, 2)
                self.grads[op["name"] + "/beta"] += ab[..., 0].sum(0).float()
                if gamma is not None:
                    self.grads[op["name"] + "/gamma"] += ab[..., 1].sum(0).float()
                ab.zero_()
                dbeta = dgamma = None
            def apply(flag):
                return lib.gv_bn_relu_bwd_apply_grouped_t(
                    self._ptr(y, True), y.ld, yptr, y.ld, self._ptr(x), x.ld, st["mean"].data_ptr(),
                    st["inv"].data_ptr(), gamma.data_ptr() if gamma is not None else None,
                    self._count(hw).data_ptr(), x.nb, hw, x.c, V, accb.data_ptr(), self._ptr(x, True), x.ld,
                    dbeta, dgamma, sc, sh, acc if self._lazy else 1, self.dt | flag, _st())
            rc = apply(_lib.GV_ACCUM_RAW_Z if raw else 0)
            if raw and rc == _lib.GV_E_UNSUPPORTED and self.bn_sync is None:
                # the apply kernel cannot convert (sum g, sum g*z) for this geometry (its non-vector path): nothing was
                # written — take the sums from the separate pass after all, and remember it for the next steps
                accb[:2 * V * x.c].zero_()
                _lib.check(lib.gv_bn_relu_bwd_sums_grouped_t(
                    self._ptr(y, True), y.ld, yptr, y.ld, self._ptr(x), x.ld, st["mean"].data_ptr(),
                    st["inv"].data_ptr(), x.nb, hw, x.c, V, accb.data_ptr(), sc, sh, self.dt | _lib.GV_ACCUM_ZEROED, _st()),
                    "bn_bwd sums " + op["name"])
                op["fused_b"]["_nofuse_b"] = True
                rc = apply(0)
            _lib.check(rc, "bn_bwd apply " + op["name"])
        elif op["kind"] == "conv":
            dz = self._ptr(y, True)
            if op["bias"]:
                src = self._bias_sources()[op["bias"]]
                if src is not None:                           # known without a pass over dy: zero, or other biases' gradients
                    for other in src:                         # (theirs are final: their ops ran earlier in this pass)
                        self.grads[op["bias"]].add_(self.grads[other])
                else:
                    _lib.check(lib.gv_bias_grad_t(dz, y.ld, y.npix, y.c, self.accum.data_ptr(),
                                                  self.grads[op["bias"]].data_ptr(), self.dt, _st()), "bias_grad")
            if op["res"] is not None:
                # y = conv(x) + r: the gradient of r receives dy.  Where this op is the FIRST contributor to it (always, in
                # ResNet-v2: a unit's conv3 is the last reader of its shortcut), the gradient of r simply IS dy's buffer
                # from here on — nothing reads dy after this op, and the later contributions (the unit's pre-activation,
                # the subsampling pool, the shortcut convolution's own backward) add to / read it in place.  Otherwise
                # (or with the A/B switch off) dy is added into r's own buffer: 16 copy passes of 100 - 800 MB per step.
                r = op["res"]
                first = self._claim(r)
                if first and self._can_alias_grad(r, y):
                    self._ptr(y, True)
                    self.grad[r.vbuf] = self.grad[y.vbuf]
                else:
                    if self.grad[r.vbuf] is not None and self.grad[r.vbuf] is self.grad[y.vbuf]:
                        self.grad[r.vbuf] = None                  # (aliased by an earlier pass: its own buffer again)
                    if first:
                        self._zero_grad_of(r)
                    _lib.check(lib.gv_accumulate_t(dz, y.ld, self._ptr(r, True), r.ld, y.npix, y.c, self.dt, _st()),
                               "res grad")
            d = self._conv_desc(op, wgrad=True)
            _lib.check(self._wgrad(d, self._ptr(x), dz, y.ld, self._dw(op).data_ptr()), "wgrad " + op["name"])
            # (a fused sibling GEMM: its gradient block IS the members' gradients, column by column)
            if x.vbuf >= 0:
                dd = self._conv_desc(op, dgrad=True)
                dx = self._ptr(x, True)
                store = self._claim(x)                        # first contribution: no residual read, plain store
                if store:
                    dd.res_ld = 0
                op["_st_b_done"] = False
                if op.get("s2") and self.s2_classes and op.get("_s2_use", x.npix >= 100000):
                    # four parity classes instead of one launch over the zero-dilated dZ (where that is faster: four
                    # launches of a quarter of the pixels each fill the chip only on the larger maps — autotune() measures
                    # both forms, the default goes by size); the BatchNorm sums of the
                    # layers that produced x ride on all four (each adds the pixels it writes) or on none
                    fuse = bool(op.get("st_b")) and zeroed and self._fusing() and op.get("_s2_stats_ok", True)
                    # The four launches are independent (disjoint pixels of dX, commutative exact sums) and each covers a
                    # quarter of the pixels — under one wave of workgroups on the 12x12 maps — so they run side by side on
                    # three extra streams, forked from and joined back into the launch stream (eager steps only: the
                    # multi-stream capture problem of enable_lanes() applies).
                    side = self._s2_streams() if (self.s2_concurrent and self._lane_streams is None and
                                                  not torch.cuda.is_current_stream_capturing()) else None
                    main = torch.cuda.current_stream(self.device)
                    if side is not None:
                        ev0 = self._s2_events[0]
                        ev0.record(main)
                    for ci, c_ in enumerate(op["s2"]):
                        stream = main if side is None or ci == 0 else side[ci - 1]
                        if stream is not main:
                            stream.wait_event(ev0)
                        with torch.cuda.stream(stream):
                            self._s2_class(op, ci, c_, dz, dx, store, fuse)
                        if stream is not main:
                            self._s2_events[ci].record(stream)
                    if side is not None:
                        for ci in range(1, len(op["s2"])):
                            main.wait_event(self._s2_events[ci])
                    ok = fuse and op.get("_s2_stats_ok", True)
                    if fuse and not ok:                       # a class declined the sums after others had added theirs
                        for b in op["st_b"]:
                            b["acc_b"].zero_()
                    op["_st_b_done"] = ok
                    return
                if op.get("st_b") and zeroed and self._fusing() and not op.get("_nofuse_b"):
                    # this launch writes the FINAL gradient of x: the backward sums of the BatchNorm layers that
                    # produced x leave its epilogue
                    rc = lib.gv_conv2d_fwd_bnstats(C.byref(dd), dz, op["w_dgrad"].data_ptr(), self.ones.data_ptr(),
                                                   self.zeros.data_ptr(), None if store else dx, dx,
                                                   C.byref(self._bn_stats(op, "st_b")), _st())
                    if rc != _lib.GV_E_UNSUPPORTED:
                        _lib.check(rc, "dgrad + BN sums " + op["name"])
                        op["_st_b_done"] = True
                        return
                _lib.check(lib.gv_conv2d_fwd(C.byref(dd), dz, op["w_dgrad"].data_ptr(), self.ones.data_ptr(),
                                             self.zeros.data_ptr(), None if store else dx, dx, None, None, None,
                                             _st()), "dgrad " + op["name"])
        else:
            if self._pool_fused(op):                          # its backward is part of the BatchNorm's (see there): the gradient
                self._claim(x)                                # of x is never materialised, only marked as produced
                return
            d = _lib.PoolDesc(x.nb, x.h, x.w, x.c, x.ld, op["k"], op["k"], op["stride"], op["pad_t"],
                              op["pad_l"], y.h, y.w, y.ld, op["mode"], self.dt)
            if self._claim(x):                                # first contribution: the gather kernels store
                d.mode |= _lib.GV_POOL_BWD_STORE
            if self.pool_argmax and "argmax" in op:          # (a record left by an earlier pool_argmax = True pass is stale)
                _lib.check(lib.gv_pool2d_bwd_argmax(C.byref(d), op["argmax"].data_ptr(), self._ptr(y, True), y.ld,
                                                    self._ptr(x, True), x.ld, _st()), "pool_bwd (argmax) " + op["name"])
            else:
                _lib.check(lib.gv_pool2d_bwd(C.byref(d), self._ptr(x), self._ptr(y, True), y.ld, self._ptr(x, True),
                                             x.ld, _st()), "pool_bwd " + op["name"])

    def apply_momentum(self, lr, mu=0.9, weight_decay=0.0):
        """tf.train.MomentumOptimizer(lr, 0.9); the slim L2 term (wd * w) applies to conv weights only."""
        for lo, hi, wd in ((0, self._n_wd, weight_decay), (self._n_wd, self._flat_p.numel(), 0.0)):
            if hi > lo:
                _lib.check(self.lib.gv_sgd_momentum(self._flat_p.data_ptr() + 4 * lo, self._flat_g.data_ptr() + 4 * lo,
                                                    self._flat_m.data_ptr() + 4 * lo, hi - lo, float(lr), float(mu),
                                                    float(wd), _st()), "sgd")
        self._packed_dirty = True

    def update_moving_averages(self, decay=None):
        """The batch-norm UPDATE_OPS of train.py:178-186: V sequential moving-average updates per BN layer from the
        batch statistics of the last forward.  decay defaults to the backbone's arg-scope value."""
        if decay is None:
            decay = 0.9997 if self.backbone == "inception_v3" else 0.997
        if self._moving_jobs is None:                     # every layer in ONE launch: a table of device pointers
            jobs, blocks = [], []
            for op in self.plan.ops:
                if op["kind"] != "bn":
                    continue
                st, x = op["stat"], op["x"]
                jobs.append(_lib.BnMovingJob(st["mean"].data_ptr(), st["var"].data_ptr(),
                                             self._count(x.h * x.w).data_ptr(),
                                             self.params[op["name"] + "/moving_mean"].data_ptr(),
                                             self.params[op["name"] + "/moving_variance"].data_ptr(), x.c, len(blocks)))
                blocks.extend([len(jobs) - 1] * ((x.c + 255) // 256))
            self._moving_jobs = (torch.frombuffer(bytearray(b"".join(bytes(j) for j in jobs)), dtype=torch.uint8)
                                 .to(self.device), len(jobs), torch.tensor(blocks, dtype=torch.int32, device=self.device),
                                 self.shape_world)
        assert self._moving_jobs[3] == self.shape_world   # (the pixel counts are part of the table)
        jd, nj, bj, _ = self._moving_jobs
        _lib.check(self.lib.gv_bn_update_moving_batched(jd.data_ptr(), nj, bj.data_ptr(), bj.numel(), self.V,
                                                        float(decay), _st()), "bn_update_moving_batched")

    def recalibrate_moving_averages(self, batches):
        """Replace every BatchNorm's moving statistics by the AVERAGE of the per-view batch statistics of `batches`
        (an iterable of view tensors [N, V, H, W, 3]) under the current variables: forward only, nothing is trained.
        This is what the V sequential moving-average updates per step (update_moving_averages, train.py:178-186) converge
        to once the variables stand still — the reference gets there by its arg-scope decay of 0.9997 over tens of
        thousands of steps.  After a SHORT run (or with a small decay) the moving statistics are an exponential window
        over the last few, noisy steps of a still-moving network, the last view's weighted most; eval-mode accuracy then
        swings between runs although the train-mode network classifies well (tools/convergence_diag.py,
        tests/test_gpu_convergence.py).  Not a call the reference has; a checkpoint written after it is an ordinary one."""
        if self.frozen_bn:
            raise RuntimeError("recalibrate_moving_averages: frozen_bn computes no batch statistics to average")
        if self.bn_sync is not None or self.shape_world > 1:
            # (a shard's batch statistics average only its own images / views: ranks would write different moving statistics)
            raise RuntimeError("recalibrate_moving_averages is a single-rank call: run it on an unsharded engine")
        bns = [op for op in self.plan.ops if op["kind"] == "bn"]
        m_sum = [torch.zeros(op["x"].c, dtype=torch.float64, device=self.device) for op in bns]
        v_sum = [torch.zeros(op["x"].c, dtype=torch.float64, device=self.device) for op in bns]
        n = 0
        for views in batches:
            self.forward_backbone(views)
            for i, op in enumerate(bns):
                cnt = self.N * op["x"].h * op["x"].w * self.shape_world
                m_sum[i] += op["stat"]["mean"].double().mean(0)
                v_sum[i] += op["stat"]["var"].double().mean(0) * (cnt / max(cnt - 1, 1))   # the unbiased estimate, as the update
            n += 1
        if n == 0:
            raise ValueError("recalibrate_moving_averages needs at least one batch")
        for i, op in enumerate(bns):
            self.params[op["name"] + "/moving_mean"].copy_((m_sum[i] / n).float())
            self.params[op["name"] + "/moving_variance"].copy_((v_sum[i] / n).float())
        return n

    def train_step(self, views, labels, lr=1e-3, mu=0.9, weight_decay=0.0, update_moving=True):
        self.forward(views, labels, check=False)
        self.backward()
        if update_moving:
            self.update_moving_averages()
        self.apply_momentum(lr, mu, weight_decay)
        return self.loss
