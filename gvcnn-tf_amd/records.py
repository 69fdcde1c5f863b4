"""Input pipeline of the reference, host side: the container formats, nothing more.

    read_tfrecords(path)          GZIP TFRecord framing (train_data.py:22-24, create_modelnet_tf_record.py:141-142)
    parse_example(record)         tf.Example with `image/encoded` (V PNG strings) and `image/label` (int64)
                                  (train_data.py:47-54, create_modelnet_tf_record.py:119-129)
    decode_png(data)              tf.image.decode_png(channels=3) (train_data.py:63): 8-bit gray / gray+alpha / RGB /
                                  RGBA / palette, non-interlaced -> uint8 [h, w, 3]
    ViewBatcher                   batches shapes, draws the augmentation decisions on the host RNG
                                  (train_data.py:81-84) and hands the decoded bytes to ONE device launch
                                  (gv_preprocess_views: resize + flips + brightness + x/255 - 0.5)

The writer halves (write_tfrecords / make_example / encode_png) exist for tests and for building fixtures; they
produce files the reference's own readers accept (masked CRC32C framing).  Pure Python + zlib: decoding stays on
host cores, the arithmetic on the pixels happens on the device.
"""
import gzip
import struct
import zlib

import numpy as np

# ------------------------------------------------------------------------------------------------
# TFRecord framing: [length u64][masked crc32c(length) u32][data][masked crc32c(data) u32]
# ------------------------------------------------------------------------------------------------
_CRC_TABLE = None


def _crc32c(data):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data):
    c = _crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def read_tfrecords(path, check_crc=False):
    """Yields the payload of every record of a (GZIP or plain) TFRecord file."""
    with open(path, "rb") as f:
        magic = f.read(2)
    opener = gzip.open if magic == b"\x1f\x8b" else open
    with opener(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("truncated TFRecord header")
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if check_crc and _masked_crc(head[:8]) != lcrc:
                raise ValueError("TFRecord length CRC mismatch")
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError("truncated TFRecord")
            if check_crc and _masked_crc(data) != struct.unpack("<I", tail)[0]:
                raise ValueError("TFRecord data CRC mismatch")
            yield data


def write_tfrecords(path, records, compress=True):
    opener = gzip.open if compress else open
    with opener(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", _masked_crc(head)) + data + struct.pack("<I", _masked_crc(data)))


# ------------------------------------------------------------------------------------------------
# tf.Example (protobuf wire format, only what the features of this dataset need)
# ------------------------------------------------------------------------------------------------
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; length-delimited values as memoryview slices."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, val


def parse_example(record):
    """tf.Example -> {feature name: list of bytes | list of int | list of float}."""
    out = {}
    buf = memoryview(record)
    for num, _, features in _fields(buf):                           # Example.features = 1
        if num != 1:
            continue
        for fnum, _, entry in _fields(features):                    # Features.feature (map entry) = 1
            if fnum != 1:
                continue
            name, feat = None, None
            for knum, _, val in _fields(entry):                     # key = 1, value = 2
                if knum == 1:
                    name = bytes(val).decode("utf8")
                elif knum == 2:
                    feat = val
            values = []
            for tnum, _, lst in _fields(feat):                      # bytes_list = 1, float_list = 2, int64_list = 3
                for vnum, wt, val in _fields(lst):
                    if tnum == 1:
                        values.append(bytes(val))
                    elif tnum == 3:
                        if wt == 2:                                 # packed
                            p = 0
                            while p < len(val):
                                v, p = _varint(val, p)
                                values.append(v - (1 << 64) if v >> 63 else v)
                        else:
                            values.append(val - (1 << 64) if val >> 63 else val)
                    elif tnum == 2:
                        raw = bytes(val)
                        values.extend(struct.unpack("<%df" % (len(raw) // 4), raw))
            out[name] = values
    return out


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(encoded_views, label):
    """tf.Example with `image/encoded` (list of PNG byte strings) and `image/label`."""
    bl = b"".join(_ld(1, v) for v in encoded_views)
    f_enc = _ld(1, b"image/encoded") + _ld(2, _ld(1, bl))
    il = _ld(1, _enc_varint(int(label)))                              # packed int64
    f_lab = _ld(1, b"image/label") + _ld(2, _ld(3, il))
    return _ld(1, _ld(1, f_enc) + _ld(1, f_lab))


# ------------------------------------------------------------------------------------------------
# PNG (tf.image.decode_png(channels=3)): 8-bit, non-interlaced
# ------------------------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def decode_png(data):
    if data[:8] != _PNG_SIG:
        raise ValueError("not a PNG")
    pos, idat, plte, trns = 8, [], None, None
    w = h = depth = ctype = None
    while pos < len(data):
        ln, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        pos += 12 + ln
        if kind == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if depth != 8 or interlace:
                raise ValueError("only 8-bit non-interlaced PNGs are supported")
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.ascontiguousarray(np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * ch))
    img = np.empty((h, w * ch), np.uint8)
    if not raw[:, 0].any():                     # filter 0 on every line: the bytes are the pixels
        img[:] = raw[:, 1:]
    else:                                       # Sub / Average / Paeth are sequential per byte: native helper
        from . import _lib
        _lib.check(_lib.load().gv_png_unfilter(raw.ctypes.data, h, w * ch, ch, img.ctypes.data), "gv_png_unfilter")
    img = img.reshape(h, w, ch)
    if ctype == 3:
        return plte[img[..., 0]]
    if ctype in (0, 4):
        return np.repeat(img[..., :1], 3, axis=2)
    return np.ascontiguousarray(img[..., :3])


def encode_png(img):
    """uint8 [h, w, 3] -> PNG bytes (filter 0 on every line)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, _ = img.shape

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    return (_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def _decode_record(rec, num_views):
    """One tf.Example -> (views uint8 [V, h0, w0, 3], label): the unit of work of ViewBatcher's decode workers."""
    ex = parse_example(rec)
    enc = ex["image/encoded"]
    if len(enc) != num_views:
        raise ValueError("record holds %d views, expected %d" % (len(enc), num_views))
    return np.stack([decode_png(e) for e in enc]), int(ex["image/label"][0])


_SHM_CACHE = {}


def _decode_record_into(rec, num_views, shm_name, offset, nbytes):
    """Worker side of the shared-memory hand-over: decode one record straight into its slot of the parent's shared
    segment (2.4 MB of pixels per 12-view shape do not travel through a pipe); returns (shape, label)."""
    views, label = _decode_record(rec, num_views)
    if views.nbytes != nbytes:
        raise ValueError("views of %d bytes, the batcher's slots hold %d (all shapes of a file share one size)"
                         % (views.nbytes, nbytes))
    shm = _SHM_CACHE.get(shm_name)
    if shm is None:
        from multiprocessing import shared_memory
        shm = _SHM_CACHE[shm_name] = shared_memory.SharedMemory(name=shm_name)
    np.ndarray(views.shape, np.uint8, buffer=shm.buf, offset=offset)[...] = views
    return views.shape, label


# ------------------------------------------------------------------------------------------------
# batches for the engine
# ------------------------------------------------------------------------------------------------
class ViewBatcher:
    """TFRecord shapes -> (views [N, V, H, W, 3] fp32 on the device, labels [N]).  path: one GZIP TFRecord file or a list
    of them (record order: _records()).  All views must share one decoded size (the ModelNet renders do).  augment=True draws the flips and the brightness delta of
    train_data.py:81-84 per view from `rng`.

    shuffle_buffer > 0: the streaming shuffle of tf.data (train_data.py:123 uses 1000 + 3 * batch_size): a buffer of that
    many shapes is filled, each new shape replaces a uniformly drawn one which is emitted, the buffer is drained in
    random order at the end — the reference's records are written class by class, so without it every batch is
    class-pure.  remainder: what happens to the last N_total % N shapes — "warn" (default) drops them with a warning and
    counts them in `dropped`; "pad" repeats the last shape up to N, pads the labels with -1 (ignored by
    gv_eval_metrics) and sets `last_valid` to the real count, for Evaluator.add_batch(..., valid=batcher.last_valid)
    (the reference's eval loop counts that short batch, eval.py:204; note that its batch-mean view scores are then taken
    over the real shapes only, here over the padded batch); "error" raises.

    workers > 0: the PNG inflate + unfilter of the records runs in that many decoder PROCESSES (tf.data's
    num_parallel_calls, train_data.py:95-104) — one Python process decodes ~0.5 k views/s, one GPU consumes 15 - 85 k.
    The outer GZIP stream is read by this process (it is one serial stream), records are dealt to the pool with a bounded
    window in flight and come back IN ORDER, so batches are identical to the workers = 0 ones.  The pool is a `spawn`
    pool: its processes never inherit an initialised GPU runtime (create the batcher — or at least start iterating —
    wherever convenient; nothing is forked)."""

    def __init__(self, path, num_views, height, width, batch_size, device, augment=False, seed=0, shuffle_buffer=0,
                 remainder="warn", workers=0):
        if remainder not in ("warn", "pad", "error"):
            raise ValueError("remainder must be 'warn', 'pad' or 'error'")
        self.path, self.V, self.H, self.W, self.N = path, num_views, height, width, batch_size
        self.device, self.augment = device, augment
        self.rng = np.random.RandomState(seed)
        self.shuffle_buffer, self.remainder = int(shuffle_buffer), remainder
        self.dropped, self.last_valid = 0, batch_size
        self.workers = int(workers)
        self._pool = None
        self._shm = None

    def close(self):
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None
        if getattr(self, "_shm", None) is not None:
            try:
                self._shm.close()
                self._shm.unlink()
            except Exception:
                pass
            self._shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _records(self):
        """The serialized records.  One file: in file order.  A LIST of files: interleaved round-robin, one record of each
        file in turn until a file ends (it then leaves the rotation) — the deterministic form of what
        tf.data.TFRecordDataset(files, num_parallel_reads=...) does in train_data.py:22-24.  With workers > 0 every file
        is inflated by its own reader THREAD (zlib releases the GIL: one GZIP stream inflates at ~17 k views/s on the GPU
        box's host, four at ~35 k), the order is the same."""
        paths = [self.path] if isinstance(self.path, (str, bytes)) or hasattr(self.path, "__fspath__") else list(self.path)
        if len(paths) == 1 and self.workers <= 0:
            yield from read_tfrecords(paths[0])
            return
        # (with decode workers ONE file goes through a reader thread as well: inflating it in the consumer's thread held the
        # whole pipeline at the ~17 k views/s of one GZIP stream minus everything else that thread does)
        if self.workers <= 0:
            its = [read_tfrecords(p) for p in paths]
            while its:
                for it in list(its):
                    rec = next(it, None)
                    if rec is None:
                        its.remove(it)
                    else:
                        yield rec
            return
        import queue
        import threading
        stop = threading.Event()
        qs = [queue.Queue(maxsize=64) for _ in paths]

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.2)
                    return
                except queue.Full:
                    pass

        def run(path, q):
            try:
                for rec in read_tfrecords(path):
                    if stop.is_set():
                        return
                    put(q, rec)
                put(q, None)
            except BaseException as e:                             # surfaces in the consumer
                put(q, e)
        threads = [threading.Thread(target=run, args=(p, q), daemon=True) for p, q in zip(paths, qs)]
        for t in threads:
            t.start()
        try:
            active = list(qs)
            while active:
                for q in list(active):
                    rec = q.get()
                    if rec is None:
                        active.remove(q)
                    elif isinstance(rec, BaseException):
                        raise rec
                    else:
                        yield rec
        finally:
            stop.set()

    def _decoded(self, zero_copy=False):
        """(views, label) per record in _records() order; workers > 0: decoded by the pool, a bounded window ahead.
        zero_copy (what __iter__ asks for): the views are VIEWS of the shared-memory ring, valid until N + 1 further
        items have been taken — a caller that keeps them longer (list(...)) must leave it off."""
        if self.workers <= 0:
            for rec in self._records():
                yield _decode_record(rec, self.V)
            return
        if self._pool is None:
            import multiprocessing as mp
            self._pool = mp.get_context("spawn").Pool(self.workers)
        from collections import deque
        from multiprocessing import shared_memory
        window, pending = 4 * self.workers, deque()
        # Without a shuffle buffer a decoded shape is handed out as a VIEW of its shared-memory slot (no 2.4 MB copy in the
        # consumer's thread, which at ~1 ms per shape capped the pipeline near 12 k views/s whatever the workers did): the
        # consumer copies it into the pinned staging buffer when its batch is complete, at most N - 1 records later, so a
        # slot may only be written again N + 1 hand-outs after its own — the ring has that many slots more than the window.
        zero_copy = bool(zero_copy) and self.shuffle_buffer <= 0
        nslots = window + (self.N + 1 if zero_copy else 0)
        records = self._records()
        first = next(records, None)
        if first is None:
            return
        views0, label0 = _decode_record(first, self.V)             # the slot size: every shape of a file has this one
        nbytes = views0.nbytes
        if getattr(self, "_shm", None) is None or self._shm.size < nslots * nbytes:
            if getattr(self, "_shm", None) is not None:
                self._shm.close()
                self._shm.unlink()
            self._shm = shared_memory.SharedMemory(create=True, size=nslots * nbytes)
        yield views0, label0

        def take(slot, res):
            shape, label = res.get(timeout=300)                       # (a worker that died — e.g. SIGBUS on a full /dev/shm — raises here)
            view = np.ndarray(shape, np.uint8, buffer=self._shm.buf, offset=slot * nbytes)
            return (view if zero_copy else view.copy()), label
        slot = 0
        try:
            for rec in records:
                if len(pending) >= window:                         # the slot about to be reused must have been read out
                    yield take(*pending.popleft())
                pending.append((slot, self._pool.apply_async(_decode_record_into, (rec, self.V, self._shm.name,
                                                                                  slot * nbytes, nbytes))))
                slot = (slot + 1) % nslots
            while pending:
                yield take(*pending.popleft())
        finally:
            # an abandoned iteration: the tasks still in flight write into shared-memory slots the next __iter__ hands out
            # again from slot 0 — wait for them (bounded: a dead worker must not block the caller forever)
            for _, res in pending:
                try:
                    res.wait(60)
                except Exception:
                    pass

    def _shapes(self, zero_copy=False):
        """(views uint8 [V, h0, w0, 3], label) per record, through the shuffle buffer."""
        buf = []
        for item in self._decoded(zero_copy):
            if self.shuffle_buffer <= 0:
                yield item
            elif len(buf) < self.shuffle_buffer:
                buf.append(item)
            else:
                k = int(self.rng.randint(0, len(buf)))
                out, buf[k] = buf[k], item
                yield out
        while buf:
            yield buf.pop(int(self.rng.randint(0, len(buf))))

    def _batch(self, imgs, labels):
        import torch
        from . import _lib
        from .model import _st
        nimg, h0, w0 = self.N * self.V, imgs[0].shape[1], imgs[0].shape[2]
        # the decoded bytes go through ONE page-locked staging buffer (two, alternating: the copy of batch k may still be
        # in flight when batch k+1 is assembled) and an asynchronous copy; on a CPU "device" (tests) plain stacking
        if str(self.device).startswith("cuda"):
            shape = (self.N, self.V, h0, w0, 3)
            if getattr(self, "_pin", None) is None or tuple(self._pin[0].shape) != shape:
                self._pin, self._pin_k = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)], 0
                self._pin_ev = [None, None]
            stage = self._pin[self._pin_k]
            # the host may only write into a staging buffer once the asynchronous copy that last read it has finished
            # (a consumer that does not synchronise every step lets the decode pool run two batches ahead of the stream)
            if self._pin_ev[self._pin_k] is not None:
                self._pin_ev[self._pin_k].synchronize()
            view = stage.numpy()
            if len(imgs) >= 8:                                     # 2.4 MB memcpys release the GIL: four threads share them
                if getattr(self, "_copiers", None) is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._copiers = ThreadPoolExecutor(4)

                def put(lo):
                    for i in range(lo, len(imgs), 4):
                        view[i] = imgs[i]
                list(self._copiers.map(put, range(4)))
            else:
                for i, im in enumerate(imgs):
                    view[i] = im
            src = stage.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pin_ev[self._pin_k] = ev
            self._pin_k ^= 1
        else:
            src = torch.from_numpy(np.stack(imgs)).to(self.device)       # [N, V, h0, w0, 3] uint8
        dst = torch.empty((self.N, self.V, self.H, self.W, 3), dtype=torch.float32, device=self.device)
        flip = delta = None
        if self.augment:
            flip = torch.from_numpy(self.rng.randint(0, 4, size=nimg).astype(np.int32)).to(self.device)
            delta = torch.from_numpy(self.rng.uniform(-1.1, 1.1, size=nimg).astype(np.float32)).to(self.device)
        _lib.check(_lib.load().gv_preprocess_views(src.data_ptr(), nimg, h0, w0, self.H, self.W,
                                                   flip.data_ptr() if flip is not None else None,
                                                   delta.data_ptr() if delta is not None else None, dst.data_ptr(),
                                                   _st()), "gv_preprocess_views")
        return dst, torch.tensor(labels, dtype=torch.int64)

    def __iter__(self):
        import warnings
        imgs, labels = [], []
        self.dropped, self.last_valid = 0, self.N
        for img, label in self._shapes(zero_copy=True):
            imgs.append(img)
            labels.append(label)
            if len(imgs) == self.N:
                yield self._batch(imgs, labels)
                imgs, labels = [], []
        if imgs:
            if self.remainder == "error":
                raise ValueError("%d trailing shapes do not fill a batch of %d" % (len(imgs), self.N))
            if self.remainder == "pad":
                self.last_valid = len(imgs)
                while len(imgs) < self.N:
                    imgs.append(imgs[self.last_valid - 1])
                    labels.append(-1)
                yield self._batch(imgs, labels)
            else:
                self.dropped = len(imgs)
                warnings.warn("ViewBatcher: the last %d shapes of %s do not fill a batch of %d and were dropped"
                              % (len(imgs), self.path, self.N))
