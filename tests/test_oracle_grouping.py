"""Oracle vs the golden vectors produced by executing the reference's
nets/model.py:16-41 (tests/golden/make_golden.py) and vs the literal constants
of the reference's unit_test.py:17-18."""
import json
import os

import numpy as np
import pytest

from oracle import grouping as G
from oracle import naive

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "grouping_golden.json")))


def _scores(case):
    return np.array([np.frombuffer(bytes.fromhex(h), dtype=np.float32)[0]
                     for h in case["scores_f32_hex"]], dtype=np.float32)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_group_scheme_weight_golden(case):
    s = _scores(case)
    if case["error"] == "IndexError":
        with pytest.raises(IndexError):
            G.group_scheme([s], case["G"], case["V"])
        bad, _, _, _ = naive.group_assign(s, case["G"])
        assert bad == 1
        return
    sch = G.group_scheme([s], case["G"], case["V"])
    assert sch.dtype == np.int64                       # np.int -> int64 in the reference
    assert sch.tolist() == case["scheme"]              # bit-exact integer path
    w = G.group_weight(sch)
    assert w.dtype == np.float32 and w.tolist() == case["weight"]
    # second, independent CPU implementation (C)
    bad, gidx, sch_c, w_c = naive.group_assign(s, case["G"])
    assert bad == 0 and sch_c.tolist() == case["scheme"] and w_c.tolist() == case["weight"]
    assert gidx.tolist() == G.group_index(s).tolist()


def test_kat1_unit_test_constants():
    k = GOLD["kat1"]
    D = np.array(k["final_view_descriptors"], dtype=np.int32)        # [V=5, F=4]
    sch = np.array(k["group_scheme"])
    views = [D[v][None, None, None, :] for v in range(D.shape[0])]   # [N=1,h=1,w=1,C=4] per view
    # unit_test.py:21,30 — mean pool, zeros dummy, int32
    got = G.view_pooling(views, sch, pool="mean", empty_fill=0)
    for g, exp in k["unit_test_mean_int32"].items():
        assert got[int(g)].reshape(-1).tolist() == exp
    # nets/model.py:44-102 on the same constants — max pool, ones dummy, 1+count weights
    fviews = [v.astype(np.float32) for v in views]
    gd = G.view_pooling(fviews, sch)
    for g, exp in k["model_py_group_max"].items():
        assert gd[int(g)].reshape(-1).tolist() == exp
    w = G.group_weight(sch)
    assert w.tolist() == k["model_py_weight"] == k["model_py_weight_via_reference"]
    S = G.group_fusion(gd, w)
    np.testing.assert_allclose(S.reshape(-1), k["model_py_shape_descriptor"], rtol=1e-6)
    # C implementation agrees
    F = np.stack(fviews)                                              # [V,N,1,1,4]
    Dc, Sc = naive.view_pool_fuse(F, sch, w)
    np.testing.assert_allclose(Sc.reshape(-1), k["model_py_shape_descriptor"], rtol=1e-6)
    for g, exp in k["model_py_group_max"].items():
        assert Dc[int(g)].reshape(-1).tolist() == exp


def test_fusion_identities():
    """all views in one group -> ((1+V)*max_v F + (G-1)*1)/(G+V)."""
    rng = np.random.RandomState(0)
    V, N, Gn = 6, 2, 5
    F = [rng.randn(N, 3, 3, 8).astype(np.float32) for _ in range(V)]
    sch = np.zeros((Gn, V), dtype=np.int64)
    sch[2, :] = 1
    w = G.group_weight(sch)
    assert w.sum() == Gn + V
    S = G.group_fusion(G.view_pooling(F, sch), w)
    mx = np.stack(F).max(axis=0)
    np.testing.assert_allclose(S, ((1 + V) * mx + (Gn - 1)) / (Gn + V), rtol=1e-6)
    # basic == max over all views
    Sb, _ = G.basic_head(F, np.zeros((8, 3), np.float32), np.zeros(3, np.float32))
    np.testing.assert_array_equal(Sb, mx)


def test_score_formula():
    """sigmoid(log|r|) == |r|/(1+|r|) (model.py:147); r=0 -> 0."""
    r = np.array([0.0, 1e-3, -0.5, 1.0, 7.0, -123.0], dtype=np.float32)
    s = G.score_from_r(r)
    np.testing.assert_allclose(s, np.abs(r) / (1 + np.abs(r)), rtol=2e-6, atol=0)
    assert s[0] == 0.0
