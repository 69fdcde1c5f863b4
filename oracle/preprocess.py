"""Oracle for the input preprocessing — TEST INFRASTRUCTURE, see oracle/__init__.py.

train_data.py:63 `tf.image.resize` is, under TF 1.x, `resize_images` v1: bilinear, align_corners=False, the legacy
sampling src = dst * (in / out) without the half-pixel shift; corners are blended as
top + (bottom - top) * y_lerp with top = tl + (tr - tl) * x_lerp (resize_bilinear_op).  Then the flips and the
brightness delta of train_data.py:81-84 and `x * (1/255) - 0.5` (train_data.py:101).  numpy fp32.
TensorFlow is not available to pin this restatement: parity unpinned.
"""
import numpy as np

F32 = np.float32


def resize_bilinear_legacy(img, height, width):
    """img uint8/float [h0, w0, C] -> float32 [height, width, C]."""
    img = np.asarray(img).astype(F32)
    h0, w0 = img.shape[:2]
    sy, sx = F32(h0) / F32(height), F32(w0) / F32(width)
    fy = (np.arange(height, dtype=F32) * sy).astype(F32)
    fx = (np.arange(width, dtype=F32) * sx).astype(F32)
    y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, h0 - 1), np.minimum(x0 + 1, w0 - 1)
    ly, lx = (fy - y0.astype(F32))[:, None, None], (fx - x0.astype(F32))[None, :, None]
    tl, tr = img[y0][:, x0], img[y0][:, x1]
    bl, br = img[y1][:, x0], img[y1][:, x1]
    top = (tl + ((tr - tl) * lx).astype(F32)).astype(F32)
    bot = (bl + ((br - bl) * lx).astype(F32)).astype(F32)
    return (top + ((bot - top) * ly).astype(F32)).astype(F32)


def preprocess_views(src, height, width, flip=None, delta=None):
    """src uint8 [nimg, h0, w0, 3] -> float32 [nimg, height, width, 3]."""
    out = np.zeros((src.shape[0], height, width, 3), dtype=F32)
    for i in range(src.shape[0]):
        im = resize_bilinear_legacy(src[i], height, width)
        f = int(flip[i]) if flip is not None else 0
        if f & 1:
            im = im[:, ::-1]
        if f & 2:
            im = im[::-1]
        if delta is not None:
            im = (im + F32(delta[i])).astype(F32)
        out[i] = (im * F32(1.0 / 255.0)).astype(F32) + F32(-0.5)
    return out
