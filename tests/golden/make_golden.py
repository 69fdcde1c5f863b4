"""Generate golden vectors by EXECUTING the reference's own host-numpy code.

Run only in the authoring container (needs /root/reference); the GPU box never
sees the reference.  Output: tests/golden/grouping_golden.json (data only).

What is executed: `nets/model.py:16-25` (`group_scheme`) and
`nets/model.py:28-41` (`group_weight`) of ace19-dev/gvcnn-tf, imported with a
stub `tensorflow` module (TensorFlow is not installable here and those two
functions do not touch it).  `np.int` (removed from NumPy >= 1.24, used at
model.py:21) is aliased to `int`.

Also recorded: the literal constants of the reference's scratch script
`unit_test.py:17-18` (KAT-1) with the values its printed dict takes, evaluated
by hand with integer (truncating) mean — the script needs TensorFlow to run.
"""
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "grouping_golden.json")


def load_reference_model():
    tf = MagicMock()
    tf.contrib.slim.add_arg_scope = lambda f: f
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, REF)
    np.int = int                     # model.py:21 uses the removed alias
    from nets import model
    return model


def f32_list(a):
    return [float(np.float32(x)) for x in a]


def main():
    model = load_reference_model()
    rng = np.random.RandomState(1234)
    cases = []

    def add(name, scores, G):
        scores = np.asarray(scores, dtype=np.float32)
        V = len(scores)
        rec = {"name": name, "G": G, "V": V, "scores_f32_hex": [np.float32(s).tobytes().hex() for s in scores],
               "scores": f32_list(scores)}
        try:
            sch = model.group_scheme([scores], G, V)
            w = model.group_weight(sch)
            rec.update(error=None, scheme=sch.astype(int).tolist(), weight=f32_list(w),
                       scheme_dtype=str(sch.dtype), weight_dtype=str(w.dtype))
        except IndexError as e:
            rec.update(error="IndexError", message=str(e))
        cases.append(rec)

    # KAT-2
    add("kat2", [0.05, 0.31, 0.349, 0.92, 0.5, 0.99], 10)
    # KAT-3 bin edges: fp32 product then int() truncation
    add("kat3_edges", [0.3, 0.29999998, 0.7, 0.70000005, 0.99999994, 0.0, 0.1, 0.2, 0.4, 0.6, 0.8, 0.9], 10)
    add("kat3_one", [1.0], 10)                       # -> IndexError (bin 10)
    add("small_G_ok", [0.05, 0.15, 0.25, 0.35, 0.45, 0.49], 5)
    add("small_G_overflow", [0.05, 0.15, 0.25, 0.35, 0.45, 0.95], 5)   # bin 9 >= 5 -> IndexError
    add("all_one_bin", [0.55] * 12, 10)
    add("single_view", [0.42], 10)
    add("G12_more_groups_than_bins", [0.05, 0.95, 0.5, 0.51], 12)
    # every multiple of 0.1 +/- 1 ulp
    edge = []
    for k in range(1, 10):
        c = np.float32(k / 10.0)
        edge += [np.nextafter(c, np.float32(0)), c, np.nextafter(c, np.float32(1))]
    add("ulp_edges", edge, 10)
    for V in (6, 12, 20):
        for rep in range(4):
            add("rand_V%d_%d" % (V, rep), rng.uniform(0, 0.99999, size=V).astype(np.float32), 10)

    kat1 = {
        "source": "unit_test.py:17-18 literal constants",
        "final_view_descriptors": [[8, 1, 220, 55], [3, 4, 3, -1], [54, 1, 6, -53], [-3, -4, 35, -1], [0, 34, 0, -23]],
        "group_scheme": [[0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0], [1, 0, 0, 1, 1], [0, 0, 0, 0, 0]],
        # unit_test.py:21,30 semantics: zeros dummy, reduce_mean on int32 (truncating)
        "unit_test_mean_int32": {"0": [3, 4, 3, -1], "1": [54, 1, 6, -53], "2": [0, 0, 0, 0],
                                 "3": [1, 10, 85, 10], "4": [0, 0, 0, 0]},
        # nets/model.py:44-102 semantics on the same data: max, ones dummy, w = 1+count
        "model_py_group_max": {"0": [3, 4, 3, -1], "1": [54, 1, 6, -53], "2": [1, 1, 1, 1],
                               "3": [8, 34, 220, 55], "4": [1, 1, 1, 1]},
        "model_py_weight": [2, 2, 1, 4, 1],
        "model_py_shape_descriptor": [14.8, 14.8, 90.0, 11.4],
    }
    # group weights of KAT-1's scheme through the reference's own group_weight
    kat1["model_py_weight_via_reference"] = f32_list(model.group_weight(np.array(kat1["group_scheme"])))

    with open(OUT, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "reference": "ace19-dev/gvcnn-tf nets/model.py:16-41 executed with numpy %s" % np.__version__,
                   "cases": cases, "kat1": kat1}, f, indent=1)
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
