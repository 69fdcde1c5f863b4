#!/usr/bin/env python3
"""Can the input pipeline (GZIP TFRecord -> tf.Example -> PNG decode -> resize / normalise) feed the
kernels?  Synthetic ModelNet-like renders (gray shaded blobs on white, 256x256 PNG with adaptive filters), V views per
shape; views/s of ViewBatcher end to end (host decode + H2D + gv_preprocess_views on the device) per worker count.
    python tools/pipeline_bench.py [--shapes 1536] [--files 1 4] [--views 12] [--sizes 224 299] [--workers 0 4 8 16 32 64]"""
import argparse
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def render(rng, size=256):
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    img = np.full((size, size), 255.0, np.float32)
    for _ in range(6):
        cx, cy, r = rng.uniform(60, 196), rng.uniform(60, 196), rng.uniform(20, 70)
        d = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
        shade = 60 + 150 * np.clip(1 - d / r, 0, 1) + rng.uniform(-3, 3, size=d.shape)
        img = np.where(d < r, shade, img)
    g = np.clip(img, 0, 255).astype(np.uint8)
    return np.stack([g, g, g], axis=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", type=int, default=1536)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--sizes", type=int, nargs="+", default=[224, 299])
    ap.add_argument("--workers", type=int, nargs="+", default=[0, 4, 8, 16, 32, 64])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--files", type=int, nargs="+", default=[1, 4], help="split the set into this many record files")
    a = ap.parse_args()
    import torch
    import gvcnn_tf_amd                              # noqa: F401  (same import order as every other entry point)
    from gvcnn_tf_amd import records as R           # (the decoder pools are spawn pools: nothing inherits a GPU context)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    rng = np.random.RandomState(0)
    base = [render(rng) for _ in range(24)]
    encoded = []
    for img in base:
        if Image is not None:                       # adaptive filters, like the renders of the dataset tools
            buf = io.BytesIO()
            Image.fromarray(img, "RGB").save(buf, format="PNG")
            encoded.append(buf.getvalue())
        else:
            encoded.append(R.encode_png(img))
    tmp = tempfile.mkdtemp(prefix="gv_pipe_")
    recs = [R.make_example([encoded[(s * a.views + v) % len(encoded)] for v in range(a.views)], s % 40) for s in range(a.shapes)]
    sets = {}
    for nf in a.files:
        sets[nf] = []
        for f in range(nf):
            sets[nf].append(os.path.join(tmp, "synthetic_%d_of_%d.record" % (f, nf)))
            R.write_tfrecords(sets[nf][-1], recs[f::nf])
    print("synthetic set: %d shapes x %d views, 256x256 PNG (%s), %.1f KB per view, %.1f MB in all; host cores %d"
          % (a.shapes, a.views, "Pillow, adaptive filters" if Image is not None else "filter 0",
             sum(len(e) for e in encoded) / len(encoded) / 1e3, sum(os.path.getsize(p) for p in sets[a.files[0]]) / 1e6,
             os.cpu_count()))
    dev = "cuda:0" if torch.cuda.is_available() else "cpu"
    for size in a.sizes:
        for nf in a.files:
            for w in a.workers:
                vb = R.ViewBatcher(sets[nf] if nf > 1 else sets[nf][0], a.views, size, size, a.batch, dev, augment=True, workers=w)
                try:
                    n = 0
                    it = iter(vb)
                    first = next(it)                # pool start-up and what the workers decoded ahead meanwhile are not
                    second = next(it)               # the steady state: the clock starts behind the second batch and runs
                    t0 = time.time()                # over the remaining 46 (short samples measure the prefetch window)
                    for x, y in it:
                        n += x.shape[0] * x.shape[1]
                    if dev != "cpu":
                        torch.cuda.synchronize()
                    dt = time.time() - t0
                finally:
                    vb.close()
                print("  %dx%d  %d file(s)  workers %2d: %8.0f views/s  (%d views in %.2f s after the second batch)"
                      % (size, size, nf, w, n / dt, n, dt))


if __name__ == "__main__":
    main()
