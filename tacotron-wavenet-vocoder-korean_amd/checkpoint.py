"""TensorFlow checkpoint bundles (`model.ckpt-N.index` + `model.ckpt-N.data-00000-of-00001` + the `checkpoint` state file)
read and written WITHOUT TensorFlow -- the data format on either side of the path.

The reference stores and restores its weights through `tf.train.Saver` (train_vocoder.py:133,176 `save`, utils/__init__.py:62-90
`save` / `load`, generate.py:157-161, synthesizer.py `Synthesizer.load`).  This module restates the on-disk format of
TensorFlow 1.x "V2" checkpoints from its published definition:

* `.index` is a LevelDB-style sorted string table (tensorflow/core/lib/io/table*): prefix-compressed key/value blocks with a
  restart array, a 5-byte trailer per block (compression type, masked CRC-32C), an index block, and a 48-byte footer that
  ends in the magic 0xdb4775248b80fb57.  Key "" holds a BundleHeaderProto, every other key is a variable name whose value is
  a BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (masked)} (tensor_bundle.proto).
* `.data-SSSSS-of-NNNNN` holds the raw little-endian tensor bytes at [offset, offset+size).

PARITY UNPINNED: TensorFlow is absent from this image and /root/reference ships no checkpoint, so the reader has only been
exercised on files this module's own writer produced (plus the published CRC-32C / Snappy known answers).  Treat the first
load of a real `model.ckpt-*` as the test that pins it; `verify=True` makes every checksum mismatch an error.
"""
import os
import struct
import warnings

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_SIZE = 262144             # TF table_options.h default
RESTART_INTERVAL = 16
MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
      9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
DT_OF = {v: k for k, v in DT.items()}


class CheckpointError(ValueError):
    pass


# ---------------------------------------------------------------- checksums
def crc32c(data, crc=0):
    """CRC-32C through the library's host helper (twv_crc32c); bytes-like or contiguous ndarray"""
    import ctypes as C
    from . import _lib
    buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return int(_lib.lib().twv_crc32c(C.c_void_p(buf.ctypes.data), buf.size, crc))


def crc_mask(c):
    return (((c >> 15) | (c << 17)) + MASK_DELTA) & 0xffffffff


def crc_unmask(m):
    r = (m - MASK_DELTA) & 0xffffffff
    return ((r >> 17) | (r << 15)) & 0xffffffff


# ---------------------------------------------------------------- varints / protobuf wire format
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(b, i):
    r, s = 0, 0
    while True:
        if i >= len(b):
            raise CheckpointError("truncated varint")
        c = b[i]
        i += 1
        r |= (c & 0x7f) << s
        if not c & 0x80:
            return r, i
        s += 7
        if s > 70:
            raise CheckpointError("varint too long")


def _proto_fields(b):
    """yield (field number, wire type, value) of one serialized message; value = int (varint/fixed) or bytes"""
    i = 0
    while i < len(b):
        tag, i = _get_varint(b, i)
        f, w = tag >> 3, tag & 7
        if w == 0:
            v, i = _get_varint(b, i)
        elif w == 1:
            v = struct.unpack_from("<Q", b, i)[0]; i += 8
        elif w == 2:
            n, i = _get_varint(b, i)
            v = bytes(b[i:i + n]); i += n
        elif w == 5:
            v = struct.unpack_from("<I", b, i)[0]; i += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % w)
        yield f, w, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(b):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for f, w, v in _proto_fields(b):
        if f == 1:
            e["dtype"] = v
        elif f == 2:                                   # TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    e["shape"].append(size)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] += 1
    return e


def _parse_header(b):
    h = {"num_shards": 0, "endianness": 0}
    for f, _, v in _proto_fields(b):
        if f == 1:
            h["num_shards"] = v
        elif f == 2:
            h["endianness"] = v
    return h


def _entry_bytes(dtype, shape, offset, size, crc_masked, shard_id=0):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(n)) for n in shape))
    out = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(dims)) + dims
    if shard_id:
        out += b"\x18" + _put_varint(shard_id)
    if offset:
        out += b"\x20" + _put_varint(offset)            # proto3: zero-valued scalars are not serialized
    if size:
        out += b"\x28" + _put_varint(size)
    return out + b"\x35" + struct.pack("<I", crc_masked)


# ---------------------------------------------------------------- Snappy (raw format), decompression only
def snappy_uncompress(b):
    n, i = _get_varint(b, 0)
    out = bytearray()
    while i < len(b):
        tag = b[i]; i += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(b[i:i + nb], "little"); i += nb
            ln += 1
            out += b[i:i + ln]; i += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | b[i]; i += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = b[i] | (b[i + 1] << 8); i += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(b[i:i + 4], "little"); i += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):                              # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch (%d != %d)" % (len(out), n))
    return bytes(out)


# ---------------------------------------------------------------- table reader
def _read_block(f, offset, size, verify):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise CheckpointError("truncated table block at %d" % offset)
    body, ctype, stored = raw[:size], raw[size], struct.unpack_from("<I", raw, size + 1)[0]
    if verify is not None:
        actual = crc32c(raw[:size + 1])
        if crc_unmask(stored) != actual:
            _complain("block at offset %d: crc32c mismatch" % offset, verify)
    if ctype == 1:
        body = snappy_uncompress(body)
    elif ctype != 0:
        raise CheckpointError("unknown block compression type %d" % ctype)
    return body


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError("table block too small")
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    if end < 0:
        raise CheckpointError("bad restart array")
    i, key = 0, b""
    while i < end:
        shared, i = _get_varint(block, i)
        non_shared, i = _get_varint(block, i)
        vlen, i = _get_varint(block, i)
        key = key[:shared] + bytes(block[i:i + non_shared]); i += non_shared
        yield key, bytes(block[i:i + vlen]); i += vlen


def _complain(msg, verify):
    if verify:
        raise CheckpointError(msg)
    warnings.warn("checkpoint: " + msg)


def read_table(path, verify=False):
    """all (key, value) pairs of one .index file, in key order.  verify: True = checksum mismatches raise, False = warn,
    None = skip the checks"""
    with open(path, "rb") as f:
        f.seek(0, 2)
        total = f.tell()
        if total < FOOTER_LEN:
            raise CheckpointError("%s is too short to be a checkpoint index" % path)
        f.seek(total - FOOTER_LEN)
        footer = f.read(FOOTER_LEN)
        if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
            raise CheckpointError("%s: bad table magic (not a TensorFlow V2 checkpoint index)" % path)
        i = 0
        _, i = _get_varint(footer, i)                   # metaindex handle
        _, i = _get_varint(footer, i)
        ioff, i = _get_varint(footer, i)
        isize, i = _get_varint(footer, i)
        out = []
        for _, handle in _block_entries(_read_block(f, ioff, isize, verify)):
            boff, j = _get_varint(handle, 0)
            bsize, j = _get_varint(handle, j)
            out.extend(_block_entries(_read_block(f, boff, bsize, verify)))
    return out


# ---------------------------------------------------------------- bundle reader
def read_bundle(prefix, verify=False, names=None):
    """{variable name: ndarray} of the bundle `prefix` (e.g. logdir/model.ckpt-1000); `names` restricts what is loaded"""
    entries = read_table(prefix + ".index", verify)
    if not entries or entries[0][0] != b"":
        raise CheckpointError("%s.index has no bundle header" % prefix)
    hdr = _parse_header(entries[0][1])
    if hdr["endianness"] != 0:
        raise CheckpointError("big-endian bundles are not supported")
    nshards = max(1, hdr["num_shards"])
    files, out = {}, {}
    try:
        for key, val in entries[1:]:
            name = key.decode("utf-8")
            if names is not None and name not in names:
                continue
            e = _parse_entry(val)
            if e["slices"]:
                raise CheckpointError("%s: partitioned (sliced) variables are not supported" % name)
            if e["dtype"] not in DT:
                if names is None:
                    continue                             # e.g. DT_STRING bookkeeping entries: not weights
                raise CheckpointError("%s: unsupported dtype %d" % (name, e["dtype"]))
            dt = DT[e["dtype"]]
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if count * dt.itemsize != e["size"]:
                raise CheckpointError("%s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, nshards), "rb")
            files[sid].seek(e["offset"])
            raw = files[sid].read(e["size"])
            if len(raw) != e["size"]:
                raise CheckpointError("%s: data file is truncated" % name)
            if verify is not None and e["crc32c"] is not None and crc_unmask(e["crc32c"]) != crc32c(raw):
                _complain("%s: crc32c mismatch" % name, verify)
            out[name] = np.frombuffer(raw, dt).reshape(e["shape"]).copy()
    finally:
        for fh in files.values():
            fh.close()
    return out


# ---------------------------------------------------------------- table / bundle writer
class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.ri, self.buf, self.restarts, self.count, self.last = restart_interval, bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.count < self.ri:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=BLOCK_SIZE):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order; blocks are stored uncompressed"""
    with open(path, "wb") as f:
        def emit(block):
            off = f.tell()
            f.write(block + b"\x00" + struct.pack("<I", crc_mask(crc32c(block + b"\x00"))))
            return _put_varint(off) + _put_varint(len(block))

        index, data, prev = _BlockBuilder(1), _BlockBuilder(RESTART_INTERVAL), None
        for key, value in items:
            if prev is not None and key <= prev:
                raise CheckpointError("table keys must be strictly increasing")
            data.add(key, value)
            prev = key
            if data.size() >= block_size:
                index.add(prev, emit(data.finish()))
                data = _BlockBuilder(RESTART_INTERVAL)
        if not data.empty():
            index.add(prev, emit(data.finish()))
        meta = emit(_BlockBuilder(RESTART_INTERVAL).finish())
        idx = emit(index.finish())
        handles = meta + idx
        f.write(handles + b"\x00" * (40 - len(handles)) + struct.pack("<Q", TABLE_MAGIC))


def write_bundle(prefix, tensors, num_shards=1):
    """bundle of {name: array}; names are stored in sorted order like tf.train.Saver does.  num_shards > 1 deals the tensors
    round-robin over `.data-0000k-of-0000n` files (what a sharded Saver / a merged multi-device save looks like on disk)"""
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    num_shards = max(1, int(num_shards))
    items = [(b"", b"\x08" + _put_varint(num_shards) + b"\x1a\x02\x08\x01")]   # BundleHeaderProto{num_shards, little endian, version{producer 1}}
    files = [open("%s.data-%05d-of-%05d" % (prefix, k, num_shards), "wb") for k in range(num_shards)]
    offs = [0] * num_shards
    try:
        for i, name in enumerate(sorted(tensors, key=lambda s: s.encode("utf-8"))):
            if not name:
                raise CheckpointError("empty variable name")
            a = np.asarray(tensors[name])
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            if np.dtype(dt) not in DT_OF:
                raise CheckpointError("%s: dtype %s cannot be stored" % (name, a.dtype))
            raw = np.ascontiguousarray(a, dtype=dt).tobytes()
            k = i % num_shards
            files[k].write(raw)
            items.append((name.encode("utf-8"), _entry_bytes(DT_OF[np.dtype(dt)], a.shape, offs[k], len(raw), crc_mask(crc32c(raw)), k)))
            offs[k] += len(raw)
    finally:
        for fh in files:
            fh.close()
    write_table(prefix + ".index", items)


# ---------------------------------------------------------------- the `checkpoint` state file (tf.train.get_checkpoint_state)
def write_checkpoint_state(logdir, model_checkpoint_path, all_paths=None):
    base = os.path.basename(model_checkpoint_path)
    paths = [os.path.basename(p) for p in (all_paths or [model_checkpoint_path])]
    with open(os.path.join(logdir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % base)
        for p in paths:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)


def latest_checkpoint(logdir):
    """tf.train.get_checkpoint_state(logdir).model_checkpoint_path (utils/__init__.py:78-85), or None"""
    state = os.path.join(logdir, "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as f:
        for line in f:
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                return p if os.path.isabs(p) else os.path.join(logdir, p)
    return None


def most_recent_checkpoint(checkpoint_dir):
    """tacotron/__init__.py:11-20: the largest N among `*.ckpt-N.data-*` files -> `<dir>/model.ckpt-N`"""
    import glob
    steps = [int(os.path.basename(p).split("-")[1].split(".")[0]) for p in glob.glob("%s/*.ckpt-*.data-*" % checkpoint_dir)]
    if not steps:
        raise CheckpointError("no *.ckpt-*.data-* file in %s" % checkpoint_dir)
    return os.path.join(checkpoint_dir, "model.ckpt-%d" % max(steps))


def checkpoint_step(path):
    """utils/__init__.py:82: the global step is parsed from the file name"""
    return int(os.path.basename(path).split("-")[-1])


def resolve(path):
    """a bundle prefix from a logdir (via its `checkpoint` file), a prefix, or a path to one of the bundle's files"""
    if os.path.isdir(path):
        p = latest_checkpoint(path)
        if p is None:
            raise CheckpointError("no `checkpoint` state file in %s" % path)
        return p
    for suf in (".index", ".data-00000-of-00001", ".meta"):
        if path.endswith(suf):
            return path[:-len(suf)]
    return path


# ---------------------------------------------------------------- variable-name mapping
EMA_SUFFIX = "/ExponentialMovingAverage"     # tf.train.ExponentialMovingAverage shadow variables (model.py:30,346)


def wavenet_tensors(variables, specs, use_ema=False):
    """the tensors `weights.tensor_specs` names, from a bundle's variables.  generate.py:157-158 restores every global variable
    whose name has no 'queue' BY ITS OWN NAME -- i.e. the raw weights, not the EMA shadows the training graph also keeps;
    use_ema=True picks the shadows instead (the usual choice for this model family, not what the reference does)."""
    out, missing = {}, []
    for name, shape in specs:
        src = name + EMA_SUFFIX if use_ema else name
        if src not in variables:
            missing.append(src)
            continue
        out[name] = np.asarray(variables[src], np.float32)
    if missing:
        raise CheckpointError("checkpoint lacks %d tensors, first: %s" % (len(missing), ", ".join(missing[:4])))
    return out


BN_PARTS = ("gamma", "beta", "moving_mean", "moving_variance")


def tacotron_tensors(variables, specs, scope="model/inference/"):
    """the tensors `tacotron.tacotron_specs` names: the graph's variables live under 'model/inference/' (tacotron.py:36,
    synthesizer.py:52); each `batch_normalization` spec is the stack of that layer's four TF variables."""
    out, missing = {}, []
    for name, shape in specs:
        if name.endswith("batch_normalization"):
            parts = [scope + name + "/" + p for p in BN_PARTS]
            if all(p in variables for p in parts):
                out[name] = np.stack([np.asarray(variables[p], np.float32) for p in parts])
            else:
                missing += [p for p in parts if p not in variables]
        elif scope + name in variables:
            out[name] = np.asarray(variables[scope + name], np.float32)
        else:
            missing.append(scope + name)
    if missing:
        raise CheckpointError("checkpoint lacks %d tensors, first: %s" % (len(missing), ", ".join(missing[:4])))
    return out


def tacotron_variable_names(spec_names, scope="model/inference/"):
    """the bundle variables `tacotron_tensors` reads for these specs (for read_bundle(names=...): Adam slots etc. are not loaded)"""
    out = set()
    for name in spec_names:
        if name.endswith("batch_normalization"):
            out.update(scope + name + "/" + p for p in BN_PARTS)
        else:
            out.add(scope + name)
    return out


# ---------------------------------------------------------------- name-tolerant restore
# TensorFlow numbers layers that are created without an explicit name (tf.layers.dense -> "dense", "dense_1", ...;
# tf.layers.conv1d -> "conv1d", "conv1d_1", ...) per GRAPH, in creation order.  The names weights.py / tacotron.py list are this
# project's reading of that rule for the reference graph; a checkpoint written by a graph that created one more (or one fewer)
# unnamed layer first -- another TF version, a train-mode graph, an extra tower -- carries the same tensors under shifted
# suffixes.  When an exact name is missing, the restore below falls back to (scope path with the auto-suffix removed, shape,
# creation order) and reports every remapping it made.
_AUTO_BASES = ("dense", "conv1d", "conv2d_transpose", "batch_normalization", "gru_cell", "embedding", "multi_rnn_cell", "output_projection_wrapper",
               "attention_wrapper", "bahdanau_monotonic_attention", "memory_layer")


def _split_auto(component):
    """('dense', 3) for 'dense_3', ('dense', 0) for 'dense'; (component, None) when it is not an auto-numbered layer name"""
    for base in _AUTO_BASES:
        if component == base:
            return base, 0
        if component.startswith(base + "_") and component[len(base) + 1:].isdigit():
            return base, int(component[len(base) + 1:])
    return component, None


def _stem(name):
    """(path with auto-suffixes removed, tuple of the removed indices)"""
    comps, idx = [], []
    for c in name.split("/"):
        base, i = _split_auto(c)
        comps.append(base)
        if i is not None:
            idx.append(i)
    return "/".join(comps), tuple(idx)


def bundle_shapes(prefix, verify=None):
    """{variable name: shape tuple} of a bundle, from its index alone (no tensor data is read)"""
    out = {}
    for key, val in read_table(prefix + ".index", verify)[1:]:
        e = _parse_entry(val)
        if e["dtype"] in DT and not e["slices"]:
            out[key.decode("utf-8")] = tuple(int(v) for v in e["shape"])
    return out


def remap_names(wanted, available, strict=False, notes=None):
    """wanted: ordered [(variable name, shape)] the graph expects; available: {variable name: shape} of the checkpoint.
    Returns {wanted name: checkpoint name}.  Tensors are grouped by (scope path without auto-suffixes, shape).
    * A group whose wanted names ALL exist in the checkpoint with the right shape is taken by name -- what tf.train.Saver.restore does
      (generate.py:157-161, synthesizer.py:69-70: by name only).  When the checkpoint holds a different NUMBER of variables of that
      scope path and shape than the graph (extra same-shape auto-numbered layers the inference graph does not build -- or one extra
      layer created first, which shifts the numbering so that the names exist and hold the neighbours' tensors), the match is still
      made, as TensorFlow would, and a line is appended to `notes`; with strict=True such a group is left unmatched instead.
    * Otherwise the whole group is matched in creation order (ascending auto-suffix) -- a shifted numbering makes SOME names collide
      with their neighbours' ("dense_2" of the checkpoint is the graph's "dense_1"), so partial exact matches inside a group would
      silently load the wrong tensors -- and only when the group sizes agree.
    Anything else stays unmatched and the caller reports it (`notes` says which groups were ambiguous)."""
    out = {}
    groups_w, groups_a = {}, {}
    for name, shape in wanted:
        st, idx = _stem(name)
        groups_w.setdefault((st, tuple(shape)), []).append((idx, name))
    for name, shape in available.items():
        st, idx = _stem(name)
        groups_a.setdefault((st, tuple(shape)), []).append((idx, name))
    for key, ws in groups_w.items():
        cand = groups_a.get(key, [])
        exact = all(n in available and tuple(available[n]) == key[1] for _, n in ws)
        if len(cand) != len(ws):
            what = "%s %s: the graph wants %d, the checkpoint holds %d" % (key[0], list(key[1]), len(ws), len(cand))
            if exact and not strict:
                if notes is not None:
                    notes.append("matched by name although the group sizes differ (as tf.train.Saver does; an extra layer created first "
                                 "would have shifted the auto-numbering): " + what)
                for _, n in ws:
                    out[n] = n
            elif notes is not None and cand:
                notes.append("ambiguous group size, left unmatched: " + what)
            continue
        if exact:
            for _, n in ws:
                out[n] = n
            continue
        for (_, wn), (_, an) in zip(sorted(ws), sorted(cand)):
            out[wn] = an
    return out


def restore_variables(prefix, wanted, verify=True, log=None, strict=False):
    """{wanted name: ndarray} from the bundle `prefix` (generate.py:157-161 / synthesizer.py:69-70 Saver.restore): by exact variable
    name where the checkpoint has it, else through remap_names; every remapping and every by-name match inside a group of a different
    size is reported through `log` (default: print).  strict=True refuses such groups (remap_names).
    Raises CheckpointError listing what could not be placed."""
    wanted = [(n, tuple(s)) for n, s in wanted]
    shapes = bundle_shapes(prefix, verify=None)
    notes = []
    mapping = remap_names(wanted, shapes, strict=strict, notes=notes)
    say = log or print
    missing = [n for n, _ in wanted if n not in mapping]
    if missing:
        amb = [m for m in notes if m.startswith("ambiguous")]
        raise CheckpointError("checkpoint lacks %d tensors (no variable of the same scope path and shape%s), first: %s%s"
                              % (len(missing), " in a group of the same size" if amb else " either", ", ".join(missing[:4]),
                                 ("; " + "; ".join(amb[:3])) if amb else ""))
    for m in notes:
        say("checkpoint: WARNING " + m)
    moved = [(w, a) for w, a in mapping.items() if w != a]
    if moved:
        say("checkpoint: %d variables restored under a different auto-generated name:" % len(moved))
        for w, a in sorted(moved):
            say("    %s  <-  %s" % (w, a))
    data = read_bundle(prefix, verify=verify, names=set(mapping.values()))
    return {w: data[a] for w, a in mapping.items()}


def tacotron_variable_specs(specs, scope="model/inference/"):
    """[(bundle variable name, shape)] behind `tacotron.tacotron_specs` entries (batch-norm entries are four variables)"""
    out = []
    for name, shape in specs:
        if name.endswith("batch_normalization"):
            out += [(scope + name + "/" + p, (shape[1],)) for p in BN_PARTS]
        else:
            out.append((scope + name, tuple(shape)))
    return out


def all_checkpoint_paths(logdir):
    """all_model_checkpoint_paths of the `checkpoint` state file, oldest first (tf.train.Saver's max_to_keep bookkeeping)"""
    state = os.path.join(logdir, "checkpoint")
    out = []
    if os.path.exists(state):
        with open(state) as f:
            for line in f:
                if line.startswith("all_model_checkpoint_paths:"):
                    q = line.split(":", 1)[1].strip().strip('"')
                    out.append(q if os.path.isabs(q) else os.path.join(logdir, q))
    return out


def delete_bundle(prefix):
    for f in [prefix + ".index"] + [q for q in __import__("glob").glob(prefix + ".data-*")]:
        if os.path.exists(f):
            os.remove(f)


def tacotron_variables(tensors, scope="model/inference/"):
    """inverse of tacotron_tensors (for writing a bundle)"""
    out = {}
    for name, a in tensors.items():
        if name.endswith("batch_normalization"):
            for i, p in enumerate(BN_PARTS):
                out[scope + name + "/" + p] = np.asarray(a[i], np.float32)
        else:
            out[scope + name] = np.asarray(a, np.float32)
    return out
