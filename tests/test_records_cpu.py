"""Container formats of the input pipeline, host logic: GZIP TFRecord framing with masked CRC32C,
the tf.Example subset of the dataset, the PNG decoder against an independent decoder (Pillow) for every colour type
and adaptive filtering, and the resize oracle's closed-form cases."""
import io
import os
import struct

import numpy as np
import pytest

from gvcnn_tf_amd import records as R
from oracle import preprocess as OP


def test_crc32c_known_answers():
    assert R._crc32c(b"123456789") == 0xE3069283                      # the CRC-32C check value
    assert R._crc32c(b"") == 0


def test_tfrecord_example_png_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    shapes = []
    for i in range(3):
        views = [rng.randint(0, 256, size=(9, 7, 3)).astype(np.uint8) for _ in range(4)]
        shapes.append((views, 5 + i))
    path = os.path.join(tmp_path, "data.record")
    R.write_tfrecords(path, [R.make_example([R.encode_png(v) for v in views], lab) for views, lab in shapes])
    with open(path, "rb") as f:
        assert f.read(2) == b"\x1f\x8b"                               # GZIP, like create_modelnet_tf_record.py:141
    got = list(R.read_tfrecords(path, check_crc=True))
    assert len(got) == 3
    for rec, (views, lab) in zip(got, shapes):
        ex = R.parse_example(rec)
        assert ex["image/label"] == [lab] and len(ex["image/encoded"]) == 4
        for e, v in zip(ex["image/encoded"], views):
            assert np.array_equal(R.decode_png(e), v)
    # a flipped payload byte is caught by the CRC
    raw = bytearray(b"".join(struct.pack("<Q", len(r)) + b"\0\0\0\0" + r + b"\0\0\0\0" for r in got[:1]))
    p2 = os.path.join(tmp_path, "bad.record")
    open(p2, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        list(R.read_tfrecords(p2, check_crc=True))


def test_example_negative_label_and_unpacked_ints():
    ex = R.parse_example(R.make_example([b"x"], -3))
    assert ex["image/label"] == [-3] and ex["image/encoded"] == [b"x"]


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P"])
def test_png_decoder_against_pillow(mode):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(len(mode))
    # smooth + noisy content so the encoder picks different filter types per line
    yy, xx = np.mgrid[0:37, 0:53]
    base = ((yy * 5 + xx * 3) % 256).astype(np.uint8)
    rgb = np.stack([base, base[::-1], rng.randint(0, 256, size=base.shape).astype(np.uint8)], axis=2)
    img = Image.fromarray(rgb, "RGB").convert(mode)
    buf = io.BytesIO()
    img.save(buf, format="PNG", optimize=True)
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    if mode in ("RGBA", "LA"):                                        # decode_png(channels=3) drops alpha
        want = np.asarray(img)[..., :3] if mode == "RGBA" else np.repeat(np.asarray(img)[..., :1], 3, axis=2)
    got = R.decode_png(buf.getvalue())
    assert got.shape == want.shape and np.array_equal(got, want)


def test_resize_oracle_closed_form_cases():
    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    assert np.array_equal(OP.resize_bilinear_legacy(img, 4, 6), img.astype(np.float32))       # identity
    up = OP.resize_bilinear_legacy(img, 8, 12)                                                 # exact 2x: src = dst/2
    assert np.array_equal(up[::2, ::2], img.astype(np.float32))
    np.testing.assert_allclose(up[1, 0], (img[0, 0].astype(np.float32) + img[1, 0]) / 2)
    np.testing.assert_allclose(up[7, 11], img[3, 5])                                           # clamped at the border
    down = OP.resize_bilinear_legacy(img, 2, 3)                                                # src = dst*2: pure sampling
    assert np.array_equal(down, img[::2, ::2].astype(np.float32))
    out = OP.preprocess_views(img[None], 4, 6, flip=[3], delta=[2.0])
    np.testing.assert_allclose(out[0], (img[::-1, ::-1].astype(np.float32) + 2.0) / 255.0 - 0.5, rtol=0, atol=1e-6)
