"""Pin the ARCHITECTURE of the hot path by EXECUTING the reference's own graph constructors.

Run only in the authoring container (needs /root/reference); the GPU box never sees the reference.
Output: tests/golden/arch_resnet_v2_50.json, arch_inception_v3.json, arch_gvcnn_graph.json (data only).

TensorFlow 1.x / tf.contrib.slim cannot be installed here, but the reference's `nets/*.py` are pure
graph *descriptions*: every line is a call into `tf` / `slim`.  This script installs a RECORDING stand-in
for the `tensorflow` module and then runs, unmodified and imported from /root/reference:

  * `nets.resnet_v2.resnet_v2_50` under `slim.arg_scope(resnet_v2.resnet_arg_scope())`
    — exactly the call of nets/model.py:137-141;
  * `nets.inception_v3.inception_v3` under `slim.arg_scope(inception_v3.inception_v3_arg_scope())`
    — exactly the (commented) call of nets/model.py:131-136;
  * `nets.model.gvcnn` and `nets.model.basic` (nets/model.py:105-206) on a 2-view input, which records the end
    points they fetch, the scorer chain, the pooling / fusion / classifier chains.

What the stand-in implements itself (a restatement of tf.contrib.framework / tf.contrib.layers 1.15
behaviour, nothing of the reference): `arg_scope` / `add_arg_scope` default-argument stacking,
`variable_scope` name nesting (string, default_name, re-entering a captured scope object),
`collect_named_outputs` / `convert_collection_to_dict`, the default arguments of `slim.conv2d`,
`slim.batch_norm`, `slim.max_pool2d`, `slim.avg_pool2d`, `slim.dropout`, and TF's SAME / VALID output-shape rule.
Every other `tf.*` call is recorded generically (name + argument tensors).  No arithmetic is performed:
tensors are (id, static shape) records.  The recorded table is therefore *the reference's own sequence of
layer calls with the arguments slim would have seen* — scope names, kernel sizes, strides, paddings, depths,
bias / BatchNorm flags (epsilon, scale, decay), activation, L2 weight decay, concat order, end-point names.
"""
import contextlib
import json
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------
# recorded tensors and ops
# ------------------------------------------------------------------------------------------------
class Rec:
    def __init__(self):
        self.ops = []
        self.variables = {}          # name -> shape (first creation wins; re-creation must agree = AUTO_REUSE)
        self.collections = {}        # collection name -> [(alias, tensor id)]
        self.n_tensors = 0
        self.scope = ""              # current variable scope name
        self.used_default_names = {}

    def tensor(self, shape, producer=None):
        t = T(self.n_tensors, shape)
        self.n_tensors += 1
        return t

    def op(self, kind, inputs, out_shape, **attrs):
        out = self.tensor(out_shape)
        rec = dict(op=kind, inputs=[t.id for t in inputs], out=out.id,
                   out_shape=list(out_shape) if out_shape is not None else None)
        rec.update(attrs)
        self.ops.append(rec)
        return out

    def variable(self, name, shape):
        shape = [int(s) for s in shape]
        if name in self.variables:
            assert self.variables[name] == shape, (name, self.variables[name], shape)
        else:
            self.variables[name] = shape


REC = Rec()


class Shape:
    def __init__(self, dims):
        self.dims = dims

    def as_list(self):
        return list(self.dims)

    def __getitem__(self, i):
        return self.dims[i]

    def __len__(self):
        return len(self.dims)

    @property
    def ndims(self):
        return len(self.dims)


class T:
    """A graph tensor: id + static shape.  Arithmetic on it records an op."""

    def __init__(self, tid, shape):
        self.id = tid
        self.shape = list(shape) if shape is not None else None

    def get_shape(self):
        return Shape(self.shape)

    def __add__(self, other):
        return REC.op("add", [self, other], self.shape, scope=REC.scope)

    __radd__ = __add__


def tensors_in(obj):
    out = []
    if isinstance(obj, T):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            out.extend(tensors_in(o))
    elif isinstance(obj, dict):
        for o in obj.values():
            out.extend(tensors_in(o))
    return out


def plain(obj):
    """JSON-able view of a call argument."""
    if isinstance(obj, T):
        return {"tensor": obj.id}
    if isinstance(obj, (list, tuple)):
        return [plain(o) for o in obj]
    if isinstance(obj, dict):
        return {str(k): plain(v) for k, v in obj.items()}
    if isinstance(obj, (int, float, str, bool)) or obj is None:
        return obj
    if isinstance(obj, Named):
        return obj.name
    if callable(obj):
        return getattr(obj, "__name__", repr(obj))
    return repr(obj)


class Named:
    """An opaque object with a name (initializers, regularizers, collection keys)."""

    def __init__(self, name, **kw):
        self.name = name
        self.__dict__.update(kw)

    def __repr__(self):
        return self.name


# ------------------------------------------------------------------------------------------------
# tf.variable_scope
# ------------------------------------------------------------------------------------------------
class VariableScope:
    def __init__(self, name):
        self.name = name

    @property
    def original_name_scope(self):
        return self.name + "/"


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, values=None, reuse=None, **kw):
    prev = REC.scope
    if isinstance(name_or_scope, VariableScope):
        new = name_or_scope.name                                  # re-entering a captured scope: not nested again
    else:
        if name_or_scope is None:
            base = default_name
            key = (prev, base)
            n = REC.used_default_names.get(key, 0)
            REC.used_default_names[key] = n + 1
            base = base if n == 0 else "%s_%d" % (base, n)        # tf's unique_name for default names
        else:
            base = name_or_scope
        new = base if not prev else prev + "/" + base
    REC.scope = new
    try:
        yield VariableScope(new)
    finally:
        REC.scope = prev
        # tf's VariableScopeStore.close_variable_subscopes: leaving a scope resets the default-name counters of its
        # sub-scopes, which is why a second AUTO_REUSE pass finds `.../unit_1/bottleneck_v2` again, not `_1`
        for key in [k for k in REC.used_default_names if k[0] == new or k[0].startswith(new + "/")]:
            del REC.used_default_names[key]


# ------------------------------------------------------------------------------------------------
# slim.arg_scope / add_arg_scope (tf.contrib.framework.python.ops.arg_scope)
# ------------------------------------------------------------------------------------------------
_ARG_STACK = [{}]


def _key(f):
    return getattr(f, "_key_op", None) or (f.__module__ + "." + f.__name__)


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
    if isinstance(list_ops_or_scope, dict):
        if kwargs:
            raise ValueError("When attempting to re-use a scope by suppling a dictionary, kwargs must be empty.")
        _ARG_STACK.append(dict(list_ops_or_scope))
        try:
            yield list_ops_or_scope
        finally:
            _ARG_STACK.pop()
        return
    if not isinstance(list_ops_or_scope, (list, tuple)):
        raise TypeError("list_ops_or_scope must either be a list/tuple or reused scope (i.e. dict)")
    cur = dict(_ARG_STACK[-1])
    for f in list_ops_or_scope:
        if not hasattr(f, "_key_op"):
            raise ValueError("%s is not decorated with @add_arg_scope" % f)
        k = _key(f)
        merged = dict(cur.get(k, {}))
        merged.update(kwargs)
        cur[k] = merged
    _ARG_STACK.append(cur)
    try:
        yield cur
    finally:
        _ARG_STACK.pop()


def add_arg_scope(func):
    key = func.__module__ + "." + func.__name__

    def with_args(*args, **kwargs):
        cur = _ARG_STACK[-1]
        merged = kwargs
        if key in cur:
            merged = dict(cur[key])
            merged.update(kwargs)
        return func(*args, **merged)
    with_args._key_op = key
    with_args.__name__ = func.__name__
    with_args.__module__ = func.__module__
    with_args.__doc__ = func.__doc__
    return with_args


# ------------------------------------------------------------------------------------------------
# collections (slim.utils)
# ------------------------------------------------------------------------------------------------
def collect_named_outputs(collections, alias, outputs):
    if collections:
        names = collections if isinstance(collections, (list, tuple)) else [collections]
        for c in names:
            REC.collections.setdefault(str(c), []).append((alias, outputs))
    return outputs


class EndPoints(dict):
    """A dict that remembers which keys were READ (the taps nets/model.py fetches)."""
    fetched = []

    def __getitem__(self, k):
        EndPoints.fetched.append(k)
        return dict.__getitem__(self, k)


def convert_collection_to_dict(collection, clear_collection=False):
    d = EndPoints()
    for alias, t in REC.collections.get(str(collection), []):
        d[alias] = t
    return d


def last_dimension(shape, min_rank=1):
    return shape[-1]


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
def _two(v):
    return [int(v), int(v)] if isinstance(v, int) else [int(v[0]), int(v[1])]


def _out_hw(size, k, s, padding):
    if size is None:
        return None
    if padding == "SAME":
        return -(-size // s)
    if padding == "VALID":
        return (size - k) // s + 1
    raise ValueError(padding)


def relu(x, name=None):
    return REC.op("relu", [x], x.shape, scope=REC.scope)


relu.__name__ = "relu"


@add_arg_scope
def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None,
               param_initializers=None, param_regularizers=None, updates_collections="update_ops",
               is_training=True, reuse=None, variables_collections=None, outputs_collections=None,
               trainable=True, batch_weights=None, fused=None, data_format="NHWC",
               zero_debias_moving_mean=False, scope=None, renorm=False, renorm_clipping=None,
               renorm_decay=0.99, adjustment=None):
    with variable_scope(scope, "BatchNorm", [inputs], reuse=reuse) as sc:
        c = inputs.shape[-1]
        if center:
            REC.variable(sc.name + "/beta", [c])
        if scale:
            REC.variable(sc.name + "/gamma", [c])
        REC.variable(sc.name + "/moving_mean", [c])
        REC.variable(sc.name + "/moving_variance", [c])
        out = REC.op("batch_norm", [inputs], inputs.shape, scope=sc.name, decay=decay, center=bool(center),
                     scale=bool(scale), epsilon=epsilon, is_training=plain(is_training), fused=plain(fused),
                     activation=plain(activation_fn))
        if activation_fn is not None:
            out = activation_fn(out)
        return collect_named_outputs(outputs_collections, sc.name, out)


@add_arg_scope
def conv2d(inputs, num_outputs, kernel_size, stride=1, padding="SAME", data_format=None, rate=1,
           activation_fn=relu, normalizer_fn=None, normalizer_params=None, weights_initializer=None,
           weights_regularizer=None, biases_initializer="zeros", biases_regularizer=None, reuse=None,
           variables_collections=None, outputs_collections=None, trainable=True, scope=None):
    with variable_scope(scope, "Conv", [inputs], reuse=reuse) as sc:
        kh, kw = _two(kernel_size)
        sh, sw = _two(stride)
        n, h, w, cin = inputs.shape
        out_shape = [n, _out_hw(h, kh, sh, padding), _out_hw(w, kw, sw, padding), int(num_outputs)]
        REC.variable(sc.name + "/weights", [kh, kw, cin, num_outputs])
        has_bias = normalizer_fn is None and biases_initializer is not None
        if has_bias:
            REC.variable(sc.name + "/biases", [num_outputs])
        out = REC.op("conv2d", [inputs], out_shape, scope=sc.name, kernel=[kh, kw], stride=[sh, sw],
                     padding=padding, rate=plain(rate), in_shape=list(inputs.shape), depth=int(num_outputs),
                     bias=bool(has_bias), normalizer=plain(normalizer_fn), activation=plain(activation_fn),
                     weights_regularizer=plain(weights_regularizer),
                     weights_initializer=plain(weights_initializer))
        if normalizer_fn is not None:
            out = normalizer_fn(out, **(normalizer_params or {}))
        if activation_fn is not None:
            out = activation_fn(out)
        return collect_named_outputs(outputs_collections, sc.name, out)


def _pool(kind, default_name):
    def pool(inputs, kernel_size, stride=2, padding="VALID", data_format="NHWC", outputs_collections=None,
             scope=None):
        name = scope if scope is not None else default_name
        full = (REC.scope + "/" if REC.scope else "") + name      # a name_scope, not a variable scope
        kh, kw = _two(kernel_size)
        sh, sw = _two(stride)
        n, h, w, c = inputs.shape
        out_shape = [n, _out_hw(h, kh, sh, padding), _out_hw(w, kw, sw, padding), c]
        out = REC.op(kind, [inputs], out_shape, scope=full, kernel=[kh, kw], stride=[sh, sw], padding=padding,
                     in_shape=list(inputs.shape))
        return collect_named_outputs(outputs_collections, full, out)
    pool.__name__ = kind
    pool.__module__ = __name__
    return add_arg_scope(pool)


max_pool2d = _pool("max_pool2d", "MaxPool2D")
avg_pool2d = _pool("avg_pool2d", "AvgPool2D")


@add_arg_scope
def dropout(inputs, keep_prob=0.5, noise_shape=None, is_training=True, outputs_collections=None, scope=None,
            seed=None):
    name = (REC.scope + "/" if REC.scope else "") + (scope or "Dropout")
    return REC.op("dropout", [inputs], inputs.shape, scope=name, keep_prob=plain(keep_prob),
                  is_training=plain(is_training))


@add_arg_scope
def fully_connected(*a, **k):
    raise NotImplementedError("slim.fully_connected is never called on the path")


def softmax(logits, scope=None):
    return REC.op("softmax", [logits], logits.shape, scope=(REC.scope + "/" if REC.scope else "") + (scope or "softmax"))


def concat(*args, **kwargs):
    """tf.concat(values, axis, name='concat') — the reference passes both by keyword."""
    values = kwargs.get("values", args[0] if args else None)
    axis = kwargs.get("axis", args[1] if len(args) > 1 else None)
    shp = list(values[0].shape)
    shp[axis] = sum(v.shape[axis] for v in values)
    return REC.op("concat", list(values), shp, scope=REC.scope, axis=axis,
                  input_depths=[v.shape[axis] for v in values])


def pad(tensor, paddings, mode="CONSTANT", name=None, constant_values=0):
    shp = [None if d is None else d + p[0] + p[1] for d, p in zip(tensor.shape, paddings)]
    return REC.op("pad", [tensor], shp, scope=REC.scope, paddings=[list(p) for p in paddings],
                  mode=mode, constant_values=constant_values)


def squeeze(x, axis=None, name=None, **kw):
    axis = axis if axis is not None else kw.get("squeeze_dims")
    shp = None
    if x.shape is not None and axis is not None:
        shp = [d for i, d in enumerate(x.shape) if i not in axis] if isinstance(axis, (list, tuple)) else \
            [d for i, d in enumerate(x.shape) if i != axis]
    return REC.op("squeeze", [x], shp, scope=REC.scope, axis=plain(axis), name=name)


def reduce_mean(x, axis=None, keepdims=None, name=None, keep_dims=None, **kw):
    keep = bool(keepdims or keep_dims)
    shp = None
    if isinstance(x, T) and x.shape is not None and axis is not None:
        ax = [axis] if isinstance(axis, int) else list(axis)
        shp = [1 if i in ax else d for i, d in enumerate(x.shape)] if keep else \
            [d for i, d in enumerate(x.shape) if i not in ax]
    elif isinstance(x, T) and axis is None:
        shp = []
    return REC.op("tf.reduce_mean", tensors_in(x), shp, axis=plain(axis), keep_dims=keep, name=name)


def transpose(x, perm=None, name=None):
    return REC.op("transpose", [x], [x.shape[p] for p in perm], perm=list(perm))


def gather(params, indices, **kw):
    ts = tensors_in(params) + tensors_in(indices)
    if isinstance(params, T) and isinstance(indices, int):
        return REC.op("gather", ts, params.shape[1:], index=indices)
    return REC.op("gather", ts, None, params=plain(params), indices=plain(indices))


def unstack(value, num=None, axis=0, name=None):
    n = value.shape[axis]
    return [REC.op("unstack[%d]" % i, [value], value.shape[:axis] + value.shape[axis + 1:], index=i)
            for i in range(n)]


def cond(pred, true_fn=None, false_fn=None, **kw):
    a = true_fn()
    b = false_fn()
    shp = b.shape if isinstance(b, T) else None                  # both branches have the false branch's rank
    if shp is not None:
        shp = [None] + list(shp[1:])                              # gathered views: leading dim is data dependent
    return REC.op("cond", [pred] + tensors_in(a) + tensors_in(b), shp, true=plain(a), false=plain(b))


def _stacked_shape(obj):
    if isinstance(obj, T):
        return obj.shape
    if isinstance(obj, (list, tuple)) and obj and all(isinstance(o, T) for o in obj) and obj[0].shape is not None:
        return [len(obj)] + list(obj[0].shape)                     # a Python list of tensors is auto-stacked by tf
    return None


# static-shape rules of the few tf ops nets/model.py:44-102,163-164 needs (only so that Keras Dense knows its fan-in)
def _shape_rule(name, args, kwargs):
    base = name.split(".")[-1]
    if base in ("ones_like", "abs", "log", "sigmoid"):
        return _stacked_shape(args[0])
    if base in ("multiply", "div", "add", "subtract"):
        shapes = [a.shape for a in args if isinstance(a, T) and a.shape is not None]
        return max(shapes, key=len) if shapes else None
    if base == "add_n":
        return _stacked_shape(args[0][0]) if args and args[0] else None
    if base in ("reduce_max", "reduce_sum"):
        shp = _stacked_shape(args[0])
        axis = kwargs.get("axis", args[1] if len(args) > 1 else None)
        if shp is None:
            return None
        return [] if axis is None else [d for i, d in enumerate(shp) if i != axis]
    return None


def generic(name):
    def f(*args, **kwargs):
        ts = tensors_in(list(args)) + tensors_in(kwargs)
        return REC.op(name, ts, _shape_rule(name, args, kwargs), args=plain(list(args)), kwargs=plain(kwargs))
    f.__name__ = name
    return f


class KerasLayer:
    counters = {}

    def __init__(self, kind, base, **cfg):
        n = KerasLayer.counters.get(base, 0)
        KerasLayer.counters[base] = n + 1
        self.kind, self.cfg = kind, cfg
        self.name = base if n == 0 else "%s_%d" % (base, n)       # Keras auto names: dense, dense_1, ...

    def __call__(self, x):
        if self.kind == "GlobalAveragePooling2D":
            shp = [x.shape[0], x.shape[3]] if x.shape is not None else None
            return REC.op("keras.GlobalAveragePooling2D", [x], shp, name=self.name)
        units = self.cfg["units"]
        cin = x.shape[-1]
        REC.variable(self.name + "/kernel", [cin, units])
        REC.variable(self.name + "/bias", [units])
        return REC.op("keras.Dense", [x], [x.shape[0], units], name=self.name, units=units)


class GenericModule(types.ModuleType):
    """Attribute access falls back to a generic recording function named <module>.<attr>."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        f = generic(self.__name__.replace("tensorflow", "tf") + "." + item)
        setattr(self, item, f)
        return f


def install_stub():
    tf = GenericModule("tensorflow")
    tf.variable_scope = variable_scope
    tf.concat = concat
    tf.pad = pad
    tf.squeeze = squeeze
    tf.transpose = transpose
    tf.reduce_mean = reduce_mean
    tf.gather = gather
    tf.unstack = unstack
    tf.cond = cond
    tf.truncated_normal_initializer = lambda mean=0.0, stddev=1.0, **k: Named("truncated_normal(%g,%g)" % (mean, stddev))
    tf.GraphKeys = types.SimpleNamespace(UPDATE_OPS="update_ops")
    tf.nn = GenericModule("tensorflow.nn")
    tf.nn.relu = relu
    tf.math = GenericModule("tensorflow.math")
    tf.compat = types.SimpleNamespace(v1=types.SimpleNamespace(AUTO_REUSE="AUTO_REUSE"))
    tf.AUTO_REUSE = "AUTO_REUSE"
    tf.keras = types.SimpleNamespace(layers=types.SimpleNamespace(
        GlobalAveragePooling2D=lambda: KerasLayer("GlobalAveragePooling2D", "global_average_pooling2d"),
        Dense=lambda units, **k: KerasLayer("Dense", "dense", units=units)))
    slim = types.SimpleNamespace(
        arg_scope=arg_scope, add_arg_scope=add_arg_scope, conv2d=conv2d, batch_norm=batch_norm,
        max_pool2d=max_pool2d, avg_pool2d=avg_pool2d, dropout=dropout, fully_connected=fully_connected,
        softmax=softmax,
        l2_regularizer=lambda scale, scope=None: Named("l2_regularizer(%g)" % scale, scale=scale),
        variance_scaling_initializer=lambda factor=2.0, mode="FAN_IN", uniform=False, **k: Named(
            "variance_scaling_initializer(factor=%g,mode=%s,uniform=%s)" % (factor, mode, uniform)),
        utils=types.SimpleNamespace(collect_named_outputs=collect_named_outputs,
                                    convert_collection_to_dict=convert_collection_to_dict,
                                    last_dimension=last_dimension))
    tf.contrib = types.SimpleNamespace(slim=slim)
    sys.modules["tensorflow"] = tf
    return tf, slim


def reset():
    global REC
    REC.__init__()
    del _ARG_STACK[1:]
    KerasLayer.counters.clear()
    EndPoints.fetched = []


def live_ops(ops, roots):
    """Indices of the ops that are ancestors of the tensors in `roots` (what a session.run of them executes)."""
    by_out = {o["out"]: i for i, o in enumerate(ops)}
    seen, stack = set(), [r for r in roots]
    while stack:
        t = stack.pop()
        if t in by_out and by_out[t] not in seen:
            i = by_out[t]
            seen.add(i)
            stack.extend(ops[i]["inputs"])
    return sorted(seen)


def dump(path, **doc):
    with open(path, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    tf, slim = install_stub()
    sys.path.insert(0, REF)
    from nets import inception_v3, model, resnet_v2

    # ---- ResNet-v2-50 exactly as nets/model.py:137-141 calls it ---------------------------------
    for size in (224,):
        reset()
        x = REC.tensor([None, size, size, 3])
        with slim.arg_scope(resnet_v2.resnet_arg_scope()):
            net, end_points = resnet_v2.resnet_v2_50(x, num_classes=40, is_training=False, reuse=tf.compat.v1.AUTO_REUSE)
        taps = {"raw": "resnet_v2_50/block3", "final": "resnet_v2_50/block4"}                    # model.py:144,149
        live = live_ops(REC.ops, [end_points[t].id for t in taps.values()])
        dump(os.path.join(HERE, "arch_resnet_v2_50.json"),
             source="nets/resnet_v2.py:230-248 resnet_v2_50(num_classes=40, is_training=False, reuse=AUTO_REUSE) under "
                    "resnet_arg_scope() (nets/resnet_utils.py:198-248), called as nets/model.py:137-141",
             input={"tensor": x.id, "shape": x.shape}, ops=REC.ops, variables=REC.variables,
             end_points={k: {"tensor": v.id, "shape": v.shape} for k, v in end_points.items()},
             taps=taps, live_ops_for_taps=live)

    # ---- Inception-v3 exactly as the commented call of nets/model.py:131-136 --------------------
    for size in (224, 299):
        reset()
        x = REC.tensor([None, size, size, 3])
        with slim.arg_scope(inception_v3.inception_v3_arg_scope()):
            logits, end_points = inception_v3.inception_v3(x, num_classes=40, is_training=False,
                                                           dropout_keep_prob=0.8, reuse=tf.compat.v1.AUTO_REUSE)
        taps = {"final": "Mixed_7c"}                                                                 # model.py:193
        live = live_ops(REC.ops, [end_points["Mixed_7c"].id])
        dump(os.path.join(HERE, "arch_inception_v3_%d.json" % size),
             source="nets/inception_v3.py:413-531 inception_v3(num_classes=40, is_training=False, dropout_keep_prob=0.8, "
                    "reuse=AUTO_REUSE) under inception_v3_arg_scope() (nets/inception_utils.py:30-78), the call "
                    "commented out at nets/model.py:131-136",
             input={"tensor": x.id, "shape": x.shape}, ops=REC.ops, variables=REC.variables,
             end_points={k: {"tensor": v.id, "shape": v.shape} for k, v in end_points.items()},
             taps=taps, live_ops_for_taps=live)

    # ---- nets/model.py gvcnn / basic on a 2-view input ------------------------------------------
    reset()
    N, V, G, C = None, 2, 3, 40
    x = REC.tensor([N, V, 224, 224, 3])
    scheme = REC.tensor([G, V])
    weight = REC.tensor([G])
    scores, S, logits = model.gvcnn(x, C, scheme, weight, is_training=False, dropout_keep_prob=1.0)
    first_backbone_op = next(i for i, o in enumerate(REC.ops) if o["op"] == "conv2d")
    compact = [o for o in REC.ops if not str(o.get("scope", "")).startswith("resnet_v2_50")
               or o["op"] in ("keras.Dense",)]
    dump(os.path.join(HERE, "arch_gvcnn_graph.json"),
         source="nets/model.py:105-166 gvcnn(inputs[N,2,224,224,3], 40, group_scheme[3,2], group_weight[3], "
                "is_training=False, dropout_keep_prob=1.0): every op OUTSIDE the backbone scopes, in call order",
         inputs={"inputs": x.id, "group_scheme": scheme.id, "group_weight": weight.id},
         outputs={"view_discrimination_scores": [t.id for t in scores], "shape_descriptor": S.id, "logits": logits.id},
         end_points_fetched=EndPoints.fetched, ops_outside_backbone=compact,
         keras_variables={k: v for k, v in REC.variables.items() if not k.startswith("resnet_v2_50")},
         backbone_variables_shared_by_views=sorted(k for k in REC.variables if k.startswith("resnet_v2_50"))[:3] + ["..."],
         n_backbone_variables=sum(1 for k in REC.variables if k.startswith("resnet_v2_50")),
         n_conv2d_calls=sum(1 for o in REC.ops if o["op"] == "conv2d"), first_backbone_op=first_backbone_op)

    reset()
    x = REC.tensor([N, V, 224, 224, 3])
    S, logits = model.basic(x, C, is_training=False, dropout_keep_prob=1.0)
    compact = [o for o in REC.ops if not str(o.get("scope", "")).startswith("resnet_v2_50")]
    dump(os.path.join(HERE, "arch_basic_graph.json"),
         source="nets/model.py:169-206 basic(inputs[N,2,224,224,3], 40, is_training=False, dropout_keep_prob=1.0)",
         inputs={"inputs": x.id}, outputs={"shape_descriptor": S.id, "logits": logits.id},
         end_points_fetched=EndPoints.fetched, ops_outside_backbone=compact,
         keras_variables={k: v for k, v in REC.variables.items() if not k.startswith("resnet_v2_50")})


if __name__ == "__main__":
    main()
