"""Oracle: the grouping module of the reference, restated in numpy (fp32).

TEST INFRASTRUCTURE — see oracle/__init__.py.  Every function cites the
reference lines it follows (paths relative to /root/reference).

Layout conventions follow the reference: view descriptors are NHWC
`[N, h, w, C]` per view, stacked view-major `[V, N, h, w, C]`.
"""
import numpy as np

F32 = np.float32


def group_scheme(view_discrimination_score, num_group, num_views, num_bins=10):
    """nets/model.py:16-25.

    `view_discrimination_score` is a 1-element sequence wrapping the V scores
    (the reference indexes `[0]`, model.py:22).  The bin is
    `int(score * 10)` with the product taken in fp32 and truncated toward
    zero (model.py:23; the literal 10 is hard-coded there).  `num_bins` is the
    build's generalisation: `num_bins=10` is the reference.
    Raises IndexError when the bin is >= num_group, like the numpy indexing
    in the reference does.
    """
    schemes = np.zeros((num_group, num_views), dtype=np.int64)
    scores = np.asarray(view_discrimination_score[0], dtype=F32)
    for idx in range(len(scores)):
        b = int(F32(scores[idx]) * F32(num_bins))
        if b >= num_group or b < -num_group:
            raise IndexError(
                "index %d is out of bounds for axis 0 with size %d" % (b, num_group))
        schemes[b, idx] = 1
    return schemes


def group_index(scores, num_bins=10):
    """Bin index per view: the `int(score*10)` of model.py:23, vectorised."""
    scores = np.asarray(scores, dtype=F32)
    return (scores * F32(num_bins)).astype(F32).astype(np.int32)  # trunc toward 0


def group_weight(g_schemes):
    """nets/model.py:28-41: w_g = 1 + #views in group g (fp32)."""
    g_schemes = np.asarray(g_schemes)
    num_group, num_views = g_schemes.shape
    weights = np.zeros((num_group,), dtype=F32)
    for i in range(num_group):
        s = 1
        for j in range(num_views):
            if g_schemes[i][j] == 1:
                s += int(g_schemes[i][j])
        weights[i] = s
    return weights


def view_pooling(final_view_descriptors, group_scheme, pool="max", empty_fill=1.0):
    """nets/model.py:44-74.

    final_view_descriptors: sequence of V arrays [N,h,w,C] (or one stacked
    [V,N,h,w,C] array).  Per group g: indices = where(scheme[g]) (model.py:66);
    non-empty -> reduce_max over the gathered views, empty -> reduce_max of
    ones_like(all views) == ones (model.py:63,68-72).
    pool="mean", empty_fill=0 is the variant in unit_test.py:21,30.
    """
    stacked = np.asarray(np.stack(list(final_view_descriptors), axis=0))
    group_scheme = np.asarray(group_scheme)
    out = {}
    for g in range(group_scheme.shape[0]):
        ind = np.nonzero(group_scheme[g])[0]
        if ind.size > 0:
            sel = stacked[ind]
            out[g] = sel.max(axis=0) if pool == "max" else _mean0(sel)
        else:
            out[g] = np.full(stacked.shape[1:], empty_fill, dtype=stacked.dtype)
    return out


def _mean0(x):
    if np.issubdtype(x.dtype, np.integer):
        # tf.reduce_mean on int32 truncates toward zero (unit_test.py:30 runs on int32)
        s = x.sum(axis=0, dtype=np.int64)
        return np.trunc(s / x.shape[0]).astype(x.dtype)
    acc = np.zeros(x.shape[1:], dtype=x.dtype)
    for v in range(x.shape[0]):          # fixed order, fp32 accumulate
        acc = acc + x[v]
    return (acc / x.dtype.type(x.shape[0])).astype(x.dtype)


def group_fusion(group_descriptors, group_weight):
    """nets/model.py:77-102: S = add_n(w_g * D_g) / reduce_sum(w)."""
    w = np.asarray(group_weight, dtype=F32)
    acc = None
    for key, value in group_descriptors.items():        # dict order = group order
        term = (w[key] * np.asarray(value, dtype=F32)).astype(F32)
        acc = term if acc is None else (acc + term).astype(F32)
    denom = F32(0)
    for x in w:
        denom = F32(denom + x)
    return (acc / denom).astype(F32)


def global_average_pool(x):
    """tf.keras.layers.GlobalAveragePooling2D (model.py:144,163): mean over h,w."""
    x = np.asarray(x, dtype=F32)
    return x.mean(axis=(1, 2), dtype=np.float64).astype(F32)


def view_score(raw_descriptor, kernel, bias):
    """nets/model.py:144-147 for ONE view.

    raw_descriptor [N,h,w,Cr]; kernel [Cr,1] / [Cr]; bias scalar.
    r_n = GAP(raw)[n] . k + b ; r = mean_n r_n ; score = sigmoid(log(|r|)).

    The dot product and the batch mean are accumulated in fp64 and rounded to fp32 once; TensorFlow's Dense / reduce_mean
    accumulate in fp32 in an unspecified order.  The two differ by a few fp32 ulps of r — far inside the path's 1e-3
    tolerance and irrelevant to the binning tests, which feed identical fp32 scores (SURVEY hard part (v)).
    """
    gap = global_average_pool(raw_descriptor).astype(np.float64)
    r_n = gap @ np.asarray(kernel, dtype=np.float64).reshape(-1) + float(bias)
    r = F32(r_n.mean())
    return score_from_r(r)


def score_from_r(r):
    """sigmoid(log(abs(r))) in fp32 (model.py:147); r == 0 -> 0."""
    r = np.abs(np.asarray(r, dtype=F32))
    with np.errstate(divide="ignore"):
        lg = np.log(r).astype(F32)
    return (F32(1) / (F32(1) + np.exp(-lg).astype(F32))).astype(F32)


def dense(x, kernel, bias):
    """tf.keras.layers.Dense(C) (model.py:164): x[N,F] @ kernel[F,C] + bias[C].  (fp64 accumulation, one rounding to
    fp32: TF accumulates in fp32 — a difference of fp32 ulps, inside the path's tolerance.)"""
    y = np.asarray(x, dtype=np.float64) @ np.asarray(kernel, dtype=np.float64)
    return (y + np.asarray(bias, dtype=np.float64)).astype(F32)


def grouping_head(final_descs, scheme, weight, cls_kernel, cls_bias,
                  pool="max", empty_fill=1.0):
    """model.py:154-164: view_pooling -> group_fusion -> GAP -> Dense."""
    gd = view_pooling(final_descs, scheme, pool=pool, empty_fill=empty_fill)
    shape_desc = group_fusion(gd, weight)
    logits = dense(global_average_pool(shape_desc), cls_kernel, cls_bias)
    return shape_desc, logits


def basic_head(final_descs, cls_kernel, cls_bias):
    """model.py:202-204 (MVCNN baseline): max over views -> GAP -> Dense."""
    stacked = np.stack(list(final_descs), axis=0)
    shape_desc = stacked.max(axis=0)
    logits = dense(global_average_pool(shape_desc), cls_kernel, cls_bias)
    return shape_desc, logits


# --------------------------------------------------------------------------
# per-shape grouping: the paper's module — every shape scores, bins and fuses its own views.
# NOT computed by the reference (nets/model.py:146 averages the response over the batch); restated here from
# the same building blocks so the HIP kernels have a checker.  No reference fixture exists: parity unpinned.
# --------------------------------------------------------------------------
def group_weight_mean_score(scheme, scores):
    """weight[g] = mean score of the views in group g (fp32, view order), 0 for an empty group."""
    scheme = np.asarray(scheme)
    w = np.zeros(scheme.shape[0], dtype=F32)
    for g in range(scheme.shape[0]):
        acc, cnt = F32(0), 0
        for v in range(scheme.shape[1]):
            if scheme[g, v]:
                acc = F32(acc + F32(scores[v]))
                cnt += 1
        w[g] = F32(acc / F32(cnt)) if cnt else F32(0)
    return w


def per_shape_grouping(final_descs, r, num_group, num_bins=10, weight_mode="count", pool="max", empty_fill=1.0):
    """final_descs [N,V,h,w,C]; r [N,V] scorer responses.  Returns (scores [N,V], schemes [N,G,V],
    weights [N,G], shape_descriptor [N,h,w,C])."""
    Fd = np.asarray(final_descs, dtype=F32)
    r = np.asarray(r, dtype=F32)
    N, V = r.shape
    scores = score_from_r(r)
    schemes = np.zeros((N, num_group, V), dtype=np.int64)
    weights = np.zeros((N, num_group), dtype=F32)
    S = np.zeros((N,) + Fd.shape[2:], dtype=F32)
    for n in range(N):
        schemes[n] = group_scheme([scores[n]], num_group, V, num_bins)
        weights[n] = group_weight(schemes[n]) if weight_mode == "count" else group_weight_mean_score(schemes[n], scores[n])
        gd = view_pooling([Fd[n:n + 1, v] for v in range(V)], schemes[n], pool=pool, empty_fill=empty_fill)
        S[n] = group_fusion(gd, weights[n])[0] if float(weights[n].sum()) != 0.0 else 0.0
    return scores, schemes, weights, S
