"""TensorFlow checkpoint-v2 bundle reader, host logic: round trip through the inverse writer with the
slim variable names of both backbones (prefix-compressed keys across several table blocks), CRC detection, dtype and
shape handling, name filtering.  TensorFlow itself is not available to produce a fixture (stated in the module)."""
import os

import struct

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


# ------------------------------------------------------------------------------------------------
# Spec-level known answers: a checkpoint-v2 bundle assembled BYTE BY BYTE here from the published formats (LevelDB table
# format: doc/table_format.md; tensorflow/core/protobuf/tensor_bundle.proto; tensorflow/core/lib/hash/crc32c.h), without
# the package's writer, and read back by the package's reader.
# ------------------------------------------------------------------------------------------------
def _v(n):                                    # protobuf / LevelDB varint
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _crc32c_ref(data):                        # bitwise CRC-32C (Castagnoli, reflected 0x82F63B78): the slow textbook form
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def _masked(crc):                             # crc32c.h: Mask() = rotate right by 15, add 0xa282ead8
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_crc32c_and_varint_known_answers():
    from gvcnn_tf_amd.records import _crc32c, _varint
    assert _crc32c_ref(b"123456789") == 0xE3069283                     # the CRC-32C check value (RFC 3720 B.4)
    assert _crc32c(b"123456789") == 0xE3069283
    assert _crc32c(bytes(32)) == 0x8A9136AA and _crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4 vectors
    assert _masked(0xE3069283) == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF
    for n, enc in ((0, b"\x00"), (1, b"\x01"), (127, b"\x7f"), (128, b"\x80\x01"), (300, b"\xac\x02"),
                   (2 ** 32 - 1, b"\xff\xff\xff\xff\x0f"), (2 ** 35, b"\x80\x80\x80\x80\x80\x01"),
                   (2 ** 63 - 1, b"\xff" * 8 + b"\x7f"), (2 ** 64 - 1, b"\xff" * 9 + b"\x01")):
        assert _v(n) == enc
        assert _varint(enc + b"\x55", 0) == (n, len(enc))


def test_reader_on_a_hand_assembled_bundle(tmp_path):
    """Two data blocks with prefix-compressed keys and restart points every 2 entries, an index block, the 48-byte footer
    with the table magic, a BundleHeaderProto under the empty key, a float32 tensor, an int64 scalar (global_step), a second
    float32 tensor; a partitioned variable (BundleEntryProto.slices, field 7) is refused by name — slim's backbone
    variables are never partitioned, and the slice keys' OrderedCode encoding is not reproduced here."""
    import numpy as np
    from gvcnn_tf_amd import tf_checkpoint as tc

    def block(entries, restart_every=2):
        out, restarts, prev = bytearray(), [], b""
        for i, (k, v) in enumerate(entries):
            if i % restart_every == 0:
                restarts.append(len(out))
                shared = 0
            else:
                shared = 0
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            out += _v(shared) + _v(len(k) - shared) + _v(len(v)) + k[shared:] + v
            prev = k
        for r in restarts:
            out += struct.pack("<I", r)
        out += struct.pack("<I", len(restarts))
        return bytes(out)

    def shape_proto(dims):                        # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
        return b"".join(b"\x12" + _v(len(d)) + d for d in (b"\x08" + _v(n) for n in dims))

    def entry(dtype, dims, offset, size, crc, slices=()):
        # BundleEntryProto: dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32), slices=7
        e = b"\x08" + _v(dtype) + b"\x12" + _v(len(shape_proto(dims))) + shape_proto(dims) + b"\x18\x00" + \
            b"\x20" + _v(offset) + b"\x28" + _v(size) + b"\x35" + struct.pack("<I", crc)
        for sl in slices:
            e += b"\x3a" + _v(len(sl)) + sl
        return e

    w = np.arange(24, dtype=np.float32).reshape(2, 3, 4) * 0.5 - 3.0
    step = np.asarray(1234567890123, dtype=np.int64)
    big = np.arange(10, dtype=np.float32).reshape(5, 2)
    data = w.tobytes() + step.tobytes() + big[:3].tobytes() + big[3:].tobytes()
    off_w, off_s, off_b0, off_b1 = 0, 96, 104, 128
    header = b"\x08\x01" + b"\x10\x00" + b"\x1a\x02\x08\x01"            # num_shards=1, LITTLE endian, version{producer=1}

    full_key = b"Mixed/other"
    crc = lambda b: _masked(_crc32c_ref(b))
    entries = sorted([
        (b"", header),
        (b"InceptionV3/Conv2d_1a_3x3/weights", entry(1, w.shape, off_w, 96, crc(w.tobytes()))),
        (full_key, entry(1, big.shape, off_b0, 40, crc(big.tobytes()))),
        (b"global_step", entry(9, (), off_s, 8, crc(step.tobytes()))),
        (b"a/first", entry(1, (3, 2), off_b0, 24, crc(big[:3].tobytes()))),
        (b"a/firstborn", entry(1, (2, 2), off_b1, 16, crc(big[3:].tobytes()))),
    ])
    b1, b2 = block(entries[:3]), block(entries[3:])
    table = bytearray()
    handles = []
    for blk, last in ((b1, entries[2][0]), (b2, entries[-1][0])):
        handles.append((last, _v(len(table)) + _v(len(blk))))
        table += blk + b"\x00" + struct.pack("<I", _masked(_crc32c_ref(blk + b"\x00")))
    meta = block([])
    meta_h = _v(len(table)) + _v(len(meta))
    table += meta + b"\x00" + struct.pack("<I", _masked(_crc32c_ref(meta + b"\x00")))
    index = block(handles, restart_every=1)
    index_h = _v(len(table)) + _v(len(index))
    table += index + b"\x00" + struct.pack("<I", _masked(_crc32c_ref(index + b"\x00")))
    footer = meta_h + index_h
    footer += bytes(40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    assert len(footer) == 48
    table += footer
    prefix = str(tmp_path / "model.ckpt-7")
    open(prefix + ".index", "wb").write(bytes(table))
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    got = tc.load_checkpoint(prefix, check_crc=True)
    assert set(got) == {"InceptionV3/Conv2d_1a_3x3/weights", "global_step", "Mixed/other", "a/first", "a/firstborn"}
    np.testing.assert_array_equal(got["InceptionV3/Conv2d_1a_3x3/weights"], w)
    assert got["global_step"].dtype == np.int64 and int(got["global_step"]) == 1234567890123
    np.testing.assert_array_equal(got["Mixed/other"], big)
    np.testing.assert_array_equal(got["a/first"], big[:3])           # shared key prefix "a/first" (prefix compression)
    np.testing.assert_array_equal(got["a/firstborn"], big[3:])
    assert tc._parse_entry(entry(1, (5, 2), 0, 0, 0, slices=(b"\x0a\x04\x08\x00\x10\x03",)))["sliced"]
    # a flipped data byte is caught by the entry CRC, a flipped index byte by the block CRC
    bad = bytearray(data)
    bad[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        tc.load_checkpoint(prefix, check_crc=True)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    t2 = bytearray(table)
    t2[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(t2))
    with pytest.raises(ValueError):
        tc.load_checkpoint(prefix, check_crc=True)
