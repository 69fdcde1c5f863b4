"""Reader for TensorFlow checkpoint-v2 bundles (load TF-slim `inception_v3` / `resnet_v2_50` weights).

A checkpoint `prefix` is `prefix.index` + `prefix.data-0000k-of-0000n` (tensorflow/core/util/tensor_bundle):

  * `.index` is a LevelDB-format table (tensorflow/core/lib/io/table*): data blocks of prefix-compressed
    key/value entries with a restart array, each block followed by a 1-byte compression type and a masked CRC32C;
    an index block mapping last-keys to block handles; a 48-byte footer (metaindex handle, index handle, padding,
    magic 0xdb4775248b80fb57).  Key "" holds the BundleHeaderProto, every other key is a variable name whose value
    is a BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}.
  * `.data-*` holds the raw little-endian tensor bytes at (shard_id, offset, size).

`load_checkpoint(prefix)` returns {variable name: numpy array} — exactly the dict `GVCNN(backbone_params=...)` /
`configure(backbone_params=...)` take, since the engine looks variables up by their slim names.
TensorFlow is not available in this environment: the reader follows the published format and is exercised by
`write_checkpoint` (the inverse: prefix-compressed keys, restart arrays, several data blocks, CRCs; used by the tests) — it has NOT been run against a file written by TensorFlow.
Snappy-compressed index blocks (not what the bundle writer emits) are rejected.
"""
import os
import struct

import numpy as np

from .records import _crc32c, _fields, _varint

_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 14: None, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_OF = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _read_block(buf, offset, size, check_crc):
    data = buf[offset:offset + size]
    ctype = buf[offset + size]
    if check_crc:
        (crc,) = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])
        if _mask(_crc32c(bytes(data) + bytes([ctype]))) != crc:
            raise ValueError("checkpoint index: block CRC mismatch")
    if ctype != 0:
        raise ValueError("checkpoint index: compressed blocks (type %d) are not supported" % ctype)
    return data


def _block_entries(block):
    """(key, value) pairs of one table block."""
    (num_restarts,) = struct.unpack("<I", block[-4:])
    end = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_index(path, check_crc=True):
    """{key bytes: value memoryview} of a `.index` table."""
    buf = memoryview(open(path, "rb").read())
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad magic)" % path)
    footer = buf[-48:]
    _, _, p = _handle(footer, 0)                        # metaindex (unused)
    ioff, isize, _ = _handle(footer, p)
    out = {}
    for _, h in _block_entries(_read_block(buf, ioff, isize, check_crc)):
        boff, bsize, _ = _handle(h, 0)
        for k, v in _block_entries(_read_block(buf, boff, bsize, check_crc)):
            out[k] = v
    return out


def _parse_entry(val):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for num, wt, v in _fields(val):
        if num == 1:
            e["dtype"] = v
        elif num == 2:                                   # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            for dnum, _, dim in _fields(v):
                if dnum == 2:
                    size = 0
                    for snum, _, sv in _fields(dim):
                        if snum == 1:
                            size = sv
                    e["shape"].append(size)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif num == 7:
            e["sliced"] = True
    return e


def load_checkpoint(prefix, names=None, check_crc=False):
    """{name: ndarray} of the variables of checkpoint `prefix` (optionally only `names`).  Partitioned (sliced)
    variables are not supported (slim's backbone variables are not partitioned)."""
    index = read_index(prefix + ".index", check_crc=True)
    header = index.get(b"")
    num_shards = 1
    if header is not None:
        for num, _, v in _fields(header):
            if num == 1:
                num_shards = v
            elif num == 2 and v != 0:
                raise ValueError("big-endian checkpoints are not supported")
    shards = {}
    out = {}
    for key, val in index.items():
        if key == b"":
            continue
        name = key.decode("utf8")
        if names is not None and name not in names:
            continue
        e = _parse_entry(val)
        if e["sliced"]:
            raise ValueError("%s is a partitioned variable" % name)
        dt = _DTYPES.get(e["dtype"])
        if dt is None:
            continue                                     # strings / resources: not tensors the engine can use
        if e["shard_id"] not in shards:
            shards[e["shard_id"]] = np.memmap("%s.data-%05d-of-%05d" % (prefix, e["shard_id"], num_shards), mode="r")
        raw = shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]]
        if check_crc and e["crc32c"] is not None and _mask(_crc32c(bytes(raw))) != e["crc32c"]:
            raise ValueError("%s: data CRC mismatch" % name)
        out[name] = np.frombuffer(bytes(raw), dtype=dt).reshape(e["shape"]).copy()
    return out


# ------------------------------------------------------------------------------------------------
# writer (tests, fixtures): one shard, one entry per restart, uncompressed blocks
# ------------------------------------------------------------------------------------------------
def _enc(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc((num << 3) | 2) + _enc(len(payload)) + payload


def _block(pairs, restart_interval=16):
    """LevelDB block: keys prefix-compressed against their predecessor, a restart (full key) every
    `restart_interval` entries, restart offsets + count at the end."""
    body, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                shared += 1
        body += _enc(shared) + _enc(len(k) - shared) + _enc(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts.append(0)
    body += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(body)


def write_checkpoint(prefix, tensors, block_entries=24):
    """Writes {name: ndarray} as a one-shard checkpoint-v2 bundle (the inverse of load_checkpoint)."""
    data = bytearray()
    entries = [(b"", _enc((1 << 3) | 0) + _enc(1) + _ld(3, _enc((1 << 3) | 0) + _enc(1)))]    # num_shards=1, version{producer=1}
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        raw = a.tobytes()                                 # C order
        shape = b"".join(_ld(2, _enc((1 << 3) | 0) + _enc(int(d))) for d in a.shape)
        val = (_enc((1 << 3) | 0) + _enc(_DT_OF[a.dtype]) + _ld(2, shape) + _enc((4 << 3) | 0) + _enc(len(data)) +
               _enc((5 << 3) | 0) + _enc(len(raw)) + _enc((6 << 3) | 5) + struct.pack("<I", _mask(_crc32c(raw))))
        entries.append((name.encode("utf8"), val))
        data += raw
    out, index_pairs = bytearray(), []

    def emit(block):
        off = len(out)
        out.extend(block + b"\x00" + struct.pack("<I", _mask(_crc32c(block + b"\x00"))))
        return _enc(off) + _enc(len(block))
    for i in range(0, len(entries), block_entries):
        chunk = entries[i:i + block_entries]
        index_pairs.append((chunk[-1][0], emit(_block(chunk))))
    meta = emit(_block([]))
    idx = emit(_block(index_pairs))
    footer = meta + idx
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    open(prefix + ".index", "wb").write(bytes(out))
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
