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


def _png_with_filters(img, filters):
    """PNG bytes of uint8 [h, w, 3] whose line y is written with filter type filters[y % len(filters)] (an independent,
    straight-from-the-specification encoder: Sub / Up / Average / Paeth need no library)."""
    import struct
    import zlib
    h, w, ch = img.shape
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    for y in range(h):
        ft = filters[y % len(filters)]
        cur, prev = rows[y], (rows[y - 1] if y else np.zeros(w * ch, np.int32))
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            p = np.zeros_like(cur)
        elif ft == 1:
            p = a
        elif ft == 2:
            p = prev
        elif ft == 3:
            p = (a + prev) >> 1
        else:
            pa, pb, pc = np.abs(prev - c), np.abs(a - c), np.abs(a + prev - 2 * c)
            p = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw += bytes([ft]) + ((cur - p) & 255).astype(np.uint8).tobytes()

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))


@pytest.mark.parametrize("filters", [(1,), (2,), (3,), (4,), (0, 1, 2, 3, 4), (4, 3, 1)])
def test_png_row_filters_through_the_native_helper(filters):
    """Every PNG filter type (the renders on disk use them all) through gv_png_unfilter — host code of the C-ABI
    library — against an encoder written from the specification; first line, first pixel and wrap-around cases included."""
    rng = np.random.RandomState(sum(filters))
    img = rng.randint(0, 256, size=(13, 17, 3)).astype(np.uint8)
    img[0, :3] = 255
    img[5] = 0
    assert np.array_equal(R.decode_png(_png_with_filters(img, filters)), img)


def test_view_batcher_shuffle_buffer_and_remainder_policy(tmp_path):
    """The streaming shuffle (train_data.py:123) emits every shape exactly once, deterministically per seed, and mixes
    the class-ordered records; the remainder policy is explicit."""
    import os
    V = 2
    rng = np.random.RandomState(0)
    shapes = [([rng.randint(0, 256, size=(6, 5, 3)).astype(np.uint8) for _ in range(V)], k) for k in range(23)]
    path = os.path.join(tmp_path, "s.record")
    R.write_tfrecords(path, [R.make_example([R.encode_png(v) for v in vs], lab) for vs, lab in shapes])
    plain = [lab for _, lab in R.ViewBatcher(path, V, 8, 8, 4, "cpu")._shapes()]
    assert plain == list(range(23))
    a = [lab for _, lab in R.ViewBatcher(path, V, 8, 8, 4, "cpu", seed=5, shuffle_buffer=8)._shapes()]
    b = [lab for _, lab in R.ViewBatcher(path, V, 8, 8, 4, "cpu", seed=5, shuffle_buffer=8)._shapes()]
    c = [lab for _, lab in R.ViewBatcher(path, V, 8, 8, 4, "cpu", seed=6, shuffle_buffer=8)._shapes()]
    assert sorted(a) == list(range(23)) and a == b and a != plain and a != c
    # a shape can leave the buffer only after it entered it: emitted position >= record position - buffer
    assert all(pos + 8 >= lab for pos, lab in enumerate(a))
    for v, lab in R.ViewBatcher(path, V, 8, 8, 4, "cpu", seed=5, shuffle_buffer=8)._shapes():
        assert np.array_equal(v, np.stack(shapes[lab][0]))
    with pytest.raises(ValueError):
        R.ViewBatcher(path, V, 8, 8, 4, "cpu", remainder="keep")


def test_decode_workers_return_the_same_shapes_in_order(tmp_path):
    """ViewBatcher(workers=K): the records are decoded by K spawned processes with a bounded window in flight and come
    back in file order — the decoded stream (and so every batch) equals the single-process one."""
    rng = np.random.RandomState(3)
    shapes = []
    for i in range(23):
        views = [rng.randint(0, 256, size=(11, 13, 3)).astype(np.uint8) for _ in range(3)]
        shapes.append((views, i % 5))
    path = os.path.join(tmp_path, "w.record")
    R.write_tfrecords(path, [R.make_example([R.encode_png(v) for v in views], lab) for views, lab in shapes])
    serial = list(R.ViewBatcher(path, 3, 8, 8, 4, "cpu")._decoded())
    vb = R.ViewBatcher(path, 3, 8, 8, 4, "cpu", workers=3)
    try:
        pooled = list(vb._decoded())
    finally:
        vb.close()
    assert len(serial) == len(pooled) == 23
    for (a, la), (b, lb), (views, lab) in zip(serial, pooled, shapes):
        assert la == lb == lab and np.array_equal(a, b) and np.array_equal(a, np.stack(views))


def test_zero_copy_views_stay_valid_for_a_whole_batch(tmp_path):
    """What __iter__ asks of the decode ring (zero_copy=True): a decoded shape comes out as a VIEW of its shared-memory slot
    and must keep its pixels until the batch it belongs to is complete — N shapes are held un-copied, then compared, for
    every batch of a file longer than the ring (more shapes than 4 * workers + N + 1 slots, so slots ARE reused)."""
    rng = np.random.RandomState(9)
    N, V = 4, 2
    shapes = [([rng.randint(0, 256, size=(9, 7, 3)).astype(np.uint8) for _ in range(V)], i) for i in range(61)]
    path = os.path.join(tmp_path, "z.record")
    R.write_tfrecords(path, [R.make_example([R.encode_png(v) for v in vs], lab) for vs, lab in shapes])
    vb = R.ViewBatcher(path, V, 8, 8, N, "cpu", workers=2)              # ring: 8 + 5 slots
    try:
        held, seen = [], 0
        for views, lab in vb._shapes(zero_copy=True):
            held.append((views, lab))
            if len(held) == N:
                for v, l in held:                                       # only now "copied", like _batch does
                    assert l == seen and np.array_equal(v, np.stack(shapes[l][0]))
                    seen += 1
                held = []
        for v, l in held:
            assert l == seen and np.array_equal(v, np.stack(shapes[l][0]))
            seen += 1
        assert seen == 61
    finally:
        vb.close()


def test_a_list_of_record_files_is_interleaved_the_same_with_and_without_workers(tmp_path):
    """ViewBatcher([files]): one record of each file in turn, files of different lengths (train_data.py:22-24 reads its
    files in parallel); reader threads + decode workers return exactly the single-process stream."""
    rng = np.random.RandomState(5)
    shapes = [([rng.randint(0, 256, size=(7, 9, 3)).astype(np.uint8) for _ in range(2)], i) for i in range(19)]
    cuts = [(0, 4), (4, 13), (13, 19)]
    paths = []
    for k, (a, b) in enumerate(cuts):
        paths.append(os.path.join(tmp_path, "f%d.record" % k))
        R.write_tfrecords(paths[-1], [R.make_example([R.encode_png(v) for v in vs], lab) for vs, lab in shapes[a:b]])
    serial = list(R.ViewBatcher(paths, 2, 8, 8, 4, "cpu")._decoded())
    want = []                                              # round-robin over the three files until each runs out
    its = [iter(range(a, b)) for a, b in cuts]
    while its:
        for it in list(its):
            i = next(it, None)
            if i is None:
                its.remove(it)
            else:
                want.append(i)
    assert [lab for _, lab in serial] == want
    vb = R.ViewBatcher(paths, 2, 8, 8, 4, "cpu", workers=3)
    try:
        pooled = list(vb._decoded())
    finally:
        vb.close()
    assert len(pooled) == 19
    for (a, la), (b, lb) in zip(serial, pooled):
        assert la == lb and np.array_equal(a, b) and np.array_equal(a, np.stack(shapes[la][0]))
