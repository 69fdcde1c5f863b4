"""TensorFlow checkpoint-v2 bundle reader, host logic: round trip through the inverse writer with the
slim variable names of both backbones (prefix-compressed keys across several table blocks), CRC detection, dtype and
shape handling, name filtering.  TensorFlow itself is not available to produce a fixture (stated in the module)."""
import os

import numpy as np
import pytest

from gvcnn_tf_amd import tf_checkpoint as TC
from oracle import backbone as OB


@pytest.mark.parametrize("backbone", ["inception_v3", "resnet_v2_50"])
def test_roundtrip_with_slim_variable_names(tmp_path, backbone):
    shapes = OB.trace_param_shapes(backbone, 75, 75)
    rng = np.random.RandomState(0)
    # the real variable names (they exercise the key prefix compression), small payloads (the CRC is pure Python)
    tensors = {k: rng.randn(*[min(d, 5) for d in s]).astype(np.float32) for k, s in shapes.items()}
    tensors["global_step"] = np.array(1234, dtype=np.int64)                       # a scalar of another dtype
    prefix = os.path.join(tmp_path, "model.ckpt-1234")
    TC.write_checkpoint(prefix, tensors)
    got = TC.load_checkpoint(prefix, check_crc=True)
    assert set(got) == set(tensors) and len(got) > 250
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    some = sorted(shapes)[:3]
    sub = TC.load_checkpoint(prefix, names=set(some))
    assert sorted(sub) == some


def test_corruption_is_detected(tmp_path):
    prefix = os.path.join(tmp_path, "m")
    TC.write_checkpoint(prefix, {"a/weights": np.arange(12, dtype=np.float32).reshape(3, 4), "a/biases": np.ones(4, np.float32)})
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError):
        TC.load_checkpoint(prefix)
    TC.write_checkpoint(prefix, {"a/weights": np.arange(12, dtype=np.float32).reshape(3, 4)})
    dat = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    dat[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(dat))
    with pytest.raises(ValueError):
        TC.load_checkpoint(prefix, check_crc=True)
    open(prefix + ".index", "wb").write(b"not a table" * 10)
    with pytest.raises(ValueError):
        TC.load_checkpoint(prefix)
