"""Canonical layer tables (test helper, not product code).

Three sources are reduced to the SAME canonical form and compared key by key (tests/test_arch_golden.py):

  * the golden fixtures `tests/golden/arch_*.json` — the reference's own constructors executed under a recording
    `tensorflow` stand-in (tests/golden/make_arch_golden.py);
  * the CPU oracle (`oracle/backbone.py`), traced while it runs on a small random input;
  * the product's launch plan (`gvcnn-tf_amd/backbones.py: BackbonePlan`), read off its symbolic op list — both the
    fused plan (sibling 1x1 GEMMs, commuted pooled branch) and the un-fused one.

Canonical form: a dict of nodes keyed by a structural key.
  conv:<scope>      k, stride, pad_before(h,w), in(h,w,c), out(h,w,c), bias, bn (eps, scale) | None, relu, src
  pool nodes        keyed structurally `max[k3,s2]<src>` / `avg[k3,s1]<src>`: kind, k, stride, pad_before, in, out, src
  preact:<bn scope> eps, scale, src        (stand-alone slim.batch_norm + ReLU: resnet_v2.py:75)
  add:<unit scope>  shortcut (src key), residual (src key)
  concat keys       `cat(<leaf>|<leaf>|...)` with nested concats flattened in channel order
`src` is the key of the node that produces the input.  Scope names carry the reference's full variable-scope path.
"""


def same_pad_before(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2


def pool_key(kind, k, s, src):
    return "%s[k%d,s%d]<%s>" % (kind, k, s, src)


# ------------------------------------------------------------------------------------------------
# golden fixture -> canonical
# ------------------------------------------------------------------------------------------------
def from_golden(doc, roots=None):
    ops = doc["ops"]
    live = set(doc["live_ops_for_taps"]) if roots is None else None
    key_of = {doc["input"]["tensor"]: "input"}
    padded = {}                     # tensor id -> (src key, pad_before_h, pad_before_w, unpadded shape)
    shape_of = {doc["input"]["tensor"]: doc["input"]["shape"]}
    nodes = {}
    for i, o in enumerate(ops):
        if live is not None and i not in live:
            continue
        kind = o["op"]
        ins = o["inputs"]
        shape_of[o["out"]] = o["out_shape"]
        if kind == "pad":
            p = o["paddings"]
            assert p[0] == [0, 0] and p[3] == [0, 0] and o["mode"] == "CONSTANT" and o["constant_values"] == 0
            padded[o["out"]] = (key_of[ins[0]], p[1][0], p[2][0], shape_of[ins[0]], p[1][1], p[2][1])
        elif kind == "conv2d":
            t = ins[0]
            kh, kw = o["kernel"]
            sh, sw = o["stride"]
            assert sh == sw and o["rate"] == 1
            if t in padded:
                src, pt, pl, ishape, pb, pr = padded[t]
                assert o["padding"] == "VALID"
            else:
                src, ishape = key_of[t], o["in_shape"]
                if o["padding"] == "SAME":
                    pt, pl = same_pad_before(ishape[1], kh, sh), same_pad_before(ishape[2], kw, sw)
                else:
                    pt = pl = 0
            key = "conv:" + o["scope"]
            assert key not in nodes, key
            nodes[key] = dict(k=[kh, kw], stride=sh, pad_before=[pt, pl], **{"in": ishape[1:]}, out=o["out_shape"][1:],
                              bias=o["bias"], bn=None, relu=False, src=src,
                              l2=o["weights_regularizer"])
            if o["normalizer"] is None:
                nodes[key]["relu"] = o["activation"] == "relu"      # applied inside conv2d, recorded as a relu op too
            key_of[o["out"]] = key
        elif kind == "batch_norm":
            src = key_of[ins[0]]
            if src.startswith("conv:") and o["scope"] == src[5:] + "/BatchNorm":
                assert o["center"] and nodes[src]["bn"] is None
                nodes[src]["bn"] = dict(eps=o["epsilon"], scale=o["scale"], decay=o["decay"])
                key_of[o["out"]] = src
            else:
                key = "preact:" + o["scope"]
                nodes[key] = dict(eps=o["epsilon"], scale=o["scale"], decay=o["decay"], src=src,
                                  relu=o["activation"] == "relu")
                key_of[o["out"]] = key
        elif kind == "relu":
            src = key_of[ins[0]]
            if src.startswith("conv:"):
                nodes[src]["relu"] = True
            key_of[o["out"]] = src
        elif kind in ("max_pool2d", "avg_pool2d"):
            kh, kw = o["kernel"]
            sh, sw = o["stride"]
            assert kh == kw and sh == sw
            ishape = o["in_shape"]
            pb = same_pad_before(ishape[1], kh, sh) if o["padding"] == "SAME" else 0
            key = pool_key(kind[:3], kh, sh, key_of[ins[0]])
            nodes[key] = dict(kind=kind[:3], k=kh, stride=sh, pad_before=[pb, same_pad_before(ishape[2], kw, sw)
                                                                            if o["padding"] == "SAME" else 0],
                              **{"in": ishape[1:]}, out=o["out_shape"][1:], src=key_of[ins[0]], name=o["scope"])
            key_of[o["out"]] = key
        elif kind == "concat":
            assert o["axis"] == 3
            leaves = []
            for t in ins:
                k = key_of[t]
                leaves.extend(k[4:-1].split("|") if k.startswith("cat(") else [k])
            key_of[o["out"]] = "cat(" + "|".join(leaves) + ")"
        elif kind == "add":
            key = "add:" + o["scope"]
            nodes[key] = dict(shortcut=key_of[ins[0]], residual=key_of[ins[1]])
            key_of[o["out"]] = key
        else:
            raise AssertionError("op %r on the live path is not part of the canonical form" % kind)
    eps = {}
    for name, ep in doc["end_points"].items():
        if ep["tensor"] in key_of:
            eps[name] = dict(key=key_of[ep["tensor"]], shape=ep["shape"][1:])
    return nodes, eps


# ------------------------------------------------------------------------------------------------
# oracle (oracle/backbone.py), traced -> canonical
# ------------------------------------------------------------------------------------------------
def from_oracle(backbone, size):
    import numpy as np
    import torch
    from oracle import backbone as B

    nodes, key_of, keep = {}, {}, []
    last_name = {}

    def reg(t, key):
        key_of[id(t)] = key
        keep.append(t)                                   # ids stay unique while the tensor is alive
        return t

    orig = dict(conv2d=B.conv2d, bn=B.batch_norm_inference, maxp=B.max_pool2d, avgp=B.avg_pool2d_same3, get=B._get,
                relu=torch.relu, cat=torch.cat, bott=B.bottleneck)

    def _get(P, name, shape):
        last_name[name.rsplit("/", 1)[1]] = name
        return orig["get"](P, name, shape)

    def conv2d(x, w, stride=1, padding="SAME", bias=None):
        y = orig["conv2d"](x, w, stride, padding, bias)
        scope = last_name["weights"].rsplit("/", 1)[0]
        kh, kw = int(w.shape[0]), int(w.shape[1])
        if padding == "SAME":
            pt, pl = same_pad_before(x.shape[1], kh, stride), same_pad_before(x.shape[2], kw, stride)
        elif padding == "VALID":
            pt = pl = 0
        else:
            pt, pl = padding[0], padding[2]
        nodes["conv:" + scope] = dict(k=[kh, kw], stride=stride, pad_before=[pt, pl], **{"in": list(x.shape[1:])},
                                      out=list(y.shape[1:]), bias=bias is not None, bn=None, relu=False,
                                      src=key_of[id(x)])
        if bias is not None:
            assert last_name["biases"] == scope + "/biases"
        return reg(y, "conv:" + scope)

    def bn(x, mean, var, beta, gamma, eps):
        y = orig["bn"](x, mean, var, beta, gamma, eps)
        src = key_of[id(x)]
        scope = last_name["beta"].rsplit("/", 1)[0]
        assert last_name["moving_mean"] == scope + "/moving_mean" and last_name["moving_variance"] == scope + "/moving_variance"
        if gamma is not None:
            assert last_name["gamma"] == scope + "/gamma"
        if src.startswith("conv:") and scope == src[5:] + "/BatchNorm":
            nodes[src]["bn"] = dict(eps=eps, scale=gamma is not None)
            return reg(y, src)
        nodes["preact:" + scope] = dict(eps=eps, scale=gamma is not None, src=src, relu=False)
        return reg(y, "preact:" + scope)

    def relu(x):
        y = orig["relu"](x)
        src = key_of[id(x)]
        nodes[src]["relu"] = True
        return reg(y, src)

    def pool(kind):
        def f(x, k=3, stride=1, padding="SAME"):
            y = orig["maxp"](x, k, stride, padding) if kind == "max" else orig["avgp"](x)
            pb = same_pad_before(x.shape[1], k, stride) if padding == "SAME" else 0
            pl = same_pad_before(x.shape[2], k, stride) if padding == "SAME" else 0
            key = pool_key(kind, k, stride, key_of[id(x)])
            nodes[key] = dict(kind=kind, k=k, stride=stride, pad_before=[pb, pl], **{"in": list(x.shape[1:])},
                              out=list(y.shape[1:]), src=key_of[id(x)])
            return reg(y, key)
        return f

    def cat(ts, dim=0):
        assert dim == 3
        leaves = []
        for t in ts:
            k = key_of[id(t)]
            leaves.extend(k[4:-1].split("|") if k.startswith("cat(") else [k])
        return reg(orig["cat"](ts, dim=dim), "cat(" + "|".join(leaves) + ")")

    def bottleneck(P, x, scope, depth, depth_bottleneck, stride, mode):
        y = orig["bott"](P, x, scope, depth, depth_bottleneck, stride, mode)
        # the two operands of `shortcut + residual` (resnet_v2.py:91): the residual is the conv3 output; the shortcut
        # is identified numerically among the candidates the reference allows (resnet_v2.py:76-81)
        res_key = "conv:" + scope + "/conv3"
        res = next(t for t in keep if key_of[id(t)] == res_key)
        sc = (y - res).detach()
        cands = {key_of[id(x)]: x}
        if stride > 1:
            cands[pool_key("max", 1, stride, key_of[id(x)])] = x[:, ::stride, ::stride, :]
        if "conv:" + scope + "/shortcut" in nodes:
            cands["conv:" + scope + "/shortcut"] = next(t for t in keep if key_of[id(t)] == "conv:" + scope + "/shortcut")
        hit = [k for k, t in cands.items() if t.shape == sc.shape and
               float((t - sc).abs().max()) <= 1e-4 * max(float(sc.abs().max()), 1.0)]
        assert len(hit) == 1, (scope, hit)
        if hit[0].startswith("max[k1"):
            nodes[hit[0]] = dict(kind="max", k=1, stride=stride, pad_before=[0, 0], **{"in": list(x.shape[1:])},
                                 out=list(sc.shape[1:]), src=key_of[id(x)])
        nodes["add:" + scope] = dict(shortcut=hit[0], residual=res_key)
        return reg(y, "add:" + scope)

    P = B.init_params(B.trace_param_shapes(backbone, size, size), seed=5)     # before the tracing wrappers go in
    B.conv2d, B.batch_norm_inference, B._get = conv2d, bn, _get
    B.max_pool2d, B.avg_pool2d_same3, B.bottleneck = pool("max"), (lambda x: pool("avg")(x, 3, 1, "SAME")), bottleneck
    torch.relu, torch.cat = relu, cat
    try:
        x = torch.rand(1, size, size, 3, generator=torch.Generator().manual_seed(0)) - 0.5
        reg(x, "input")
        if backbone == "inception_v3":
            _, ep = B.inception_v3_base(x, P)
        else:
            _, ep = B.resnet_v2_50(x, P)
    finally:
        B.conv2d, B.batch_norm_inference, B._get = orig["conv2d"], orig["bn"], orig["get"]
        B.max_pool2d, B.avg_pool2d_same3, B.bottleneck = orig["maxp"], orig["avgp"], orig["bott"]
        torch.relu, torch.cat = orig["relu"], orig["cat"]
    eps = {name: dict(key=key_of[id(t)], shape=list(t.shape[1:])) for name, t in ep.items()}
    return nodes, eps, sorted(P)


# ------------------------------------------------------------------------------------------------
# product launch plan (symbolic op list of BackbonePlan) -> canonical
# ------------------------------------------------------------------------------------------------
def from_plan(backbone, size, fuse=True):
    from gvcnn_tf_amd import _lib, backbones as PB
    b = PB.BackbonePlan(1, size, size)
    if backbone == "inception_v3":
        PB.build_inception_v3(b, fuse_siblings=fuse)
    else:
        PB.build_resnet_v2_50(b)
    ss = {}                                                  # (scale_off) -> spec
    for kind, name, c, eps, has_gamma, so, ho in b.ss_specs:
        ss[so] = (kind, name, c, eps, has_gamma)
    filt = {name: (kh, kw, cin, cout) for name, kh, kw, cin, cout, off in b.filters}
    nodes = {}
    regions = {}                                             # vbuf -> [(lo, hi, key)]

    def write(t, key):
        regions.setdefault(t.vbuf, [])
        regions[t.vbuf] = [r for r in regions[t.vbuf] if r[1] <= t.off or r[0] >= t.off + t.c]
        regions[t.vbuf].append((t.off, t.off + t.c, key))

    def read(t):
        if t.vbuf < 0:
            return "input"
        rs = sorted(r for r in regions[t.vbuf] if r[0] < t.off + t.c and r[1] > t.off)
        assert rs and rs[0][0] == t.off and rs[-1][1] == t.off + t.c, (t, rs)
        for a, c in zip(rs, rs[1:]):
            assert a[1] == c[0], (t, rs)
        if len(rs) == 1:
            return rs[0][2]
        leaves = []
        for r in rs:
            leaves.extend(r[2][4:-1].split("|") if r[2].startswith("cat(") else [r[2]])
        return "cat(" + "|".join(leaves) + ")"

    def bn_of(off, c_lo=0):
        kind, name, c, eps, has_gamma = ss[off]
        return kind, name, eps, has_gamma

    pending_pooled = {}                                      # scratch key -> conv scope whose ReLU the pool applies
    for op in b.ops:
        x, y = op["x"], op["y"]
        if op["kind"] == "conv":
            members = op["name"].split("+")
            src = read(x)
            if len(members) == 1:
                scope = members[0]
                kind, name, eps, has_gamma = bn_of(op["scale_off"])
                assert filt[scope + "/weights"] == (op["kh"], op["kw"], x.c, op["cout"])
                node = dict(k=[op["kh"], op["kw"]], stride=op["stride"], pad_before=[op["pad_t"], op["pad_l"]],
                            **{"in": [x.h, x.w, x.c]}, out=[y.h, y.w, y.c], bias=kind == "bias",
                            bn=dict(eps=eps, scale=has_gamma) if kind == "bn" else None, relu=bool(op["relu"]), src=src)
                if kind == "bn":
                    assert name == scope + "/BatchNorm"
                else:
                    assert name == scope + "/biases"
                key = "conv:" + scope
                nodes[key] = node
                if op["res"] is not None:                    # shortcut + residual in the conv3 epilogue (resnet_v2.py:91)
                    unit = scope.rsplit("/", 1)[0]
                    nodes["add:" + unit] = dict(shortcut=read(op["res"]), residual=key)
                    key = "add:" + unit
                write(y, key)
                if op["y2"] is not None:                     # second output = the NEXT unit's preact (resnet_v2.py:75)
                    k2, name2, eps2, g2 = bn_of(op["scale2_off"])
                    nodes["preact:" + name2] = dict(eps=eps2, scale=g2, src=key, relu=True)
                    write(op["y2"], "preact:" + name2)
            else:                                            # GV_CONV_SPLIT: sibling 1x1 convs over one input
                assert (op["kh"], op["kw"], op["stride"], op["pad_t"], op["pad_l"]) == (1, 1, 1, 0, 0)
                col = 0
                for j, scope in enumerate(members):
                    kh, kw, cin, cout = filt[scope + "/weights"]
                    assert (kh, kw, cin) == (1, 1, x.c)
                    spec = [s for s in b.ss_specs if s[1] == scope + "/BatchNorm"]
                    assert len(spec) == 1 and spec[0][5] == op["scale_off"] + col
                    relu_here = bool(op["relu"]) and (op["relu_cols"] == 0 or col < op["relu_cols"])
                    nodes["conv:" + scope] = dict(k=[1, 1], stride=1, pad_before=[0, 0], **{"in": [x.h, x.w, x.c]},
                                                  out=[y.h, y.w, cout], bias=False,
                                                  bn=dict(eps=spec[0][3], scale=spec[0][4]), relu=relu_here, src=src)
                    if j == 0:
                        assert cout == op["split"]
                        write(y, "conv:" + scope)
                    else:
                        lo = col - op["split"]
                        write(op["y2"].channels(lo, lo + cout), "conv:" + scope)
                        if not relu_here:
                            pending_pooled["conv:" + scope] = scope
                    col += cout
                assert col == op["cout"]
        elif op["kind"] == "pool":
            src = read(x)
            op = dict(op, mode=op["mode"] & 0xff)                 # (storage-format flags are not architecture)
            if op["mode"] == _lib.GV_POOL_AVG_RELU:
                # relu(avgpool(BN(conv1x1(x)))) stands for relu(BN(conv1x1(avgpool(x)))) (inception_v3.py:152-154): put the
                # canonical node back in the reference order
                conv_key = src
                assert conv_key in pending_pooled and op["k"] == 3 and op["stride"] == 1 and op["pad_t"] == 1
                cn = nodes[conv_key]
                pk = pool_key("avg", 3, 1, cn["src"])
                nodes[pk] = dict(kind="avg", k=3, stride=1, pad_before=[1, 1], **{"in": cn["in"]}, out=cn["in"],
                                 src=cn["src"], name=op["name"])
                cn["src"], cn["relu"] = pk, True
                del pending_pooled[conv_key]
                write(y, conv_key)
            else:
                kind = "max" if op["mode"] == _lib.GV_POOL_MAX else "avg"
                key = pool_key(kind, op["k"], op["stride"], src)
                nodes[key] = dict(kind=kind, k=op["k"], stride=op["stride"], pad_before=[op["pad_t"], op["pad_l"]],
                                  **{"in": [x.h, x.w, x.c]}, out=[y.h, y.w, y.c], src=src, name=op["name"])
                write(y, key)
        elif op["kind"] == "ssa":
            kind, name, eps, has_gamma = bn_of(op["scale_off"])
            nodes["preact:" + name] = dict(eps=eps, scale=has_gamma, src=read(x), relu=bool(op["relu"]))
            write(y, "preact:" + name)
        else:
            raise AssertionError(op["kind"])
    assert not pending_pooled
    eps = {name: dict(key=read(t), shape=[t.h, t.w, t.c]) for name, t in b.end_points.items()}
    return nodes, eps, sorted(set(n for n, *_ in b.filters) | set(
        s[1] + "/" + leaf for s in b.ss_specs if s[0] == "bn"
        for leaf in (["beta", "moving_mean", "moving_variance"] + (["gamma"] if s[4] else []))) | set(
        s[1] for s in b.ss_specs if s[0] == "bias"))
