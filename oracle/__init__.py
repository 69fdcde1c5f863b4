"""CPU oracle for the GVCNN hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference algorithm
(ace19-dev/gvcnn-tf: nets/model.py, nets/inception_v3.py, nets/resnet_v2.py,
nets/resnet_utils.py, nets/inception_utils.py).  It exists to check the HIP
path; it is never shipped, never imported by the product package
(`gvcnn-tf_amd/`), and never the thing that is measured.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import it.

Parity pinning status
---------------------
* integer path (`group_scheme`, `group_weight`): PINNED — checked against
  golden vectors produced by executing the reference's own
  `nets/model.py:16-41` in the authoring container
  (`tests/golden/make_golden.py`, fixtures in `tests/golden/`).
* float path (backbones, scorer, view pooling, fusion, classifier): PARITY
  UNPINNED — the arithmetic lives in TensorFlow 1.x (`tf.contrib.slim`,
  version not pinned by the reference; not installable here), and the
  reference holds no test that pins a float result.  The restatement is
  cross-checked by (i) the literal constants of the reference's
  `unit_test.py:17-18`, (ii) two independent CPU implementations
  (torch-functional `backbone.py` vs direct-loop C `naive.c`) agreeing,
  (iii) the shape comments written in `nets/inception_v3.py:96-386`,
  (iv) parameter counts.
"""
