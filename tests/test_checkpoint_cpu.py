"""TensorFlow V2 checkpoint bundles without TensorFlow (twvk_amd.checkpoint): published known answers for the checksum,
hand-assembled Snappy / protobuf bytes, and round trips through the module's own writer.  PARITY UNPINNED against a
TensorFlow-written file (none exists in this image or in the reference tree)."""
import os
import struct
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import twvk_amd                                   # noqa: E402
from twvk_amd import checkpoint as ck            # noqa: E402


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 + the classic check value
    assert ck.crc32c(b"123456789") == 0xE3069283
    assert ck.crc32c(bytes(32)) == 0x8A9136AA
    assert ck.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert ck.crc32c(bytes(range(32))) == 0x46DD794E
    assert ck.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    data = np.random.RandomState(0).bytes(100003)
    for cut in (0, 1, 7, 8, 9, 4096, 100002):
        assert ck.crc32c(data[cut:], ck.crc32c(data[:cut])) == ck.crc32c(data)      # incremental == one shot, any alignment
    assert ck.crc32c(b"") == 0


def test_crc_mask_round_trip():
    assert ck.crc_mask(0) == 0xa282ead8
    for c in (0, 1, 0xffffffff, 0xE3069283, 0x12345678):
        m = ck.crc_mask(c)
        assert 0 <= m <= 0xffffffff and ck.crc_unmask(m) == c
    assert ck.crc_mask(0xE3069283) != 0xE3069283


def test_snappy_hand_assembled_streams():
    # literal "abcd", then copy-1 (offset 4, length 8): tag = ((8-4)<<2)|1, offset byte 4
    assert ck.snappy_uncompress(bytes([12, 3 << 2]) + b"abcd" + bytes([(4 << 2) | 1, 4])) == b"abcd" * 3
    # overlapping copy-2: offset 1, length 10 -> run of the last byte
    assert ck.snappy_uncompress(bytes([11, 0]) + b"x" + bytes([(9 << 2) | 2, 1, 0])) == b"x" * 11
    # long literal: tag 60 -> one length byte (len-1)
    lit = bytes(range(200))
    assert ck.snappy_uncompress(bytes([200, 1, 60 << 2, 199]) + lit) == lit
    # copy-4
    assert ck.snappy_uncompress(bytes([8, 3 << 2]) + b"wxyz" + bytes([(3 << 2) | 3, 4, 0, 0, 0])) == b"wxyzwxyz"
    with pytest.raises(ck.CheckpointError):
        ck.snappy_uncompress(bytes([4, (3 << 2) | 2, 9, 0]))                          # offset beyond the output
    with pytest.raises(ck.CheckpointError):
        ck.snappy_uncompress(bytes([9, 3 << 2]) + b"abcd")                            # length mismatch


def test_bundle_entry_protobuf_bytes():
    # BundleEntryProto{dtype: DT_FLOAT, shape{dim{size:2} dim{size:3}}, size: 24, crc32c: 0x04030201}, offset 0 omitted (proto3)
    want = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x28, 0x18, 0x35, 1, 2, 3, 4])
    assert ck._entry_bytes(1, (2, 3), 0, 24, 0x04030201) == want
    e = ck._parse_entry(want)
    assert (e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"], e["shard_id"]) == (1, [2, 3], 0, 24, 0x04030201, 0)
    # a scalar at a large offset: varints > 1 byte
    b = ck._entry_bytes(3, (), 300, 4, 7)
    assert b == bytes([0x08, 0x03, 0x12, 0x00, 0x20, 0xac, 0x02, 0x28, 0x04, 0x35, 7, 0, 0, 0])
    e = ck._parse_entry(b)
    assert (e["dtype"], e["shape"], e["offset"], e["size"]) == (3, [], 300, 4)
    assert ck._parse_header(b"\x08\x01\x1a\x02\x08\x01") == {"num_shards": 1, "endianness": 0}


def _items(n):
    return [(("wavenet/dilated_stack/layer%03d/dilation_layer/conv_filter/kernel" % i).encode(), os.urandom(1 + i % 37)) for i in range(n)]


@pytest.mark.parametrize("block_size", [64, 700, 262144])
def test_table_round_trip(tmp_path, block_size):
    items = [(b"", b"hdr")] + _items(300)
    p = str(tmp_path / "t.index")
    ck.write_table(p, items, block_size=block_size)
    raw = open(p, "rb").read()
    assert raw[-8:] == struct.pack("<Q", 0xdb4775248b80fb57) and raw[-8:] == bytes([0x57, 0xfb, 0x80, 0x8b, 0x24, 0x75, 0x47, 0xdb])
    assert ck.read_table(p, verify=True) == items
    with pytest.raises(ck.CheckpointError):
        ck.write_table(p, [(b"b", b""), (b"a", b"")])


def test_table_reader_handles_snappy_blocks_and_detects_corruption(tmp_path):
    """re-wrap every block of a written table as a Snappy stream (literals + one copy) with type byte 1"""
    items = [(b"", b"h")] + _items(40)
    p = str(tmp_path / "a.index")
    ck.write_table(p, items, block_size=512)
    raw = open(p, "rb").read()
    footer = raw[-48:]
    i = 0
    moff, i = ck._get_varint(footer, i); msz, i = ck._get_varint(footer, i)
    ioff, i = ck._get_varint(footer, i); isz, i = ck._get_varint(footer, i)

    def snap(b):                                                    # a valid (if pointless) Snappy encoding of b
        out = ck._put_varint(len(b))
        body, tail = (b[:-4], True) if len(b) >= 8 and b[-4:] == b[-8:-4] else (b, False)
        for o in range(0, len(body), 60):
            c = body[o:o + 60]
            out += bytes([(len(c) - 1) << 2]) + c
        if tail:
            out += bytes([(0 << 2) | 1, 4])                         # copy-1: length 4, offset 4
        return out

    out, handles = b"", []
    index_entries = list(ck._block_entries(raw[ioff:ioff + isz]))
    new_index = ck._BlockBuilder(1)
    for key, h in index_entries:
        o, j = ck._get_varint(h, 0); n, j = ck._get_varint(h, j)
        blk = snap(raw[o:o + n])
        new_index.add(key, ck._put_varint(len(out)) + ck._put_varint(len(blk)))
        out += blk + b"\x01" + struct.pack("<I", ck.crc_mask(ck.crc32c(blk + b"\x01")))
    meta = raw[moff:moff + msz + 5]
    mh = ck._put_varint(len(out)) + ck._put_varint(msz); out += meta
    ib = snap(new_index.finish())
    ih = ck._put_varint(len(out)) + ck._put_varint(len(ib))
    out += ib + b"\x01" + struct.pack("<I", ck.crc_mask(ck.crc32c(ib + b"\x01")))
    hs = mh + ih
    out += hs + b"\x00" * (40 - len(hs)) + footer[-8:]
    q = str(tmp_path / "b.index")
    open(q, "wb").write(out)
    assert ck.read_table(q, verify=True) == items
    bad = bytearray(out); bad[3] ^= 0x40
    open(q, "wb").write(bytes(bad))
    with pytest.raises(ck.CheckpointError):
        ck.read_table(q, verify=True)
    bad = bytearray(out); bad[-1] ^= 1
    open(q, "wb").write(bytes(bad))
    with pytest.raises(ck.CheckpointError):
        ck.read_table(q, verify=None)                               # bad magic is always an error


def test_bundle_round_trip_and_checksums(tmp_path):
    rng = np.random.RandomState(1)
    var = {"wavenet/conv1d/kernel": rng.randn(32, 1, 32).astype(np.float32),
           "wavenet/gc_embedding": rng.randn(2, 32).astype(np.float32),
           "global_step": np.asarray(123456, np.int32),
           "optimizer/beta1_power": np.asarray(0.5, np.float32),
           "counts": np.arange(7, dtype=np.int64),
           "empty": np.zeros((0, 4), np.float32),
           "wavenet/conv1d/kernel/ExponentialMovingAverage": rng.randn(32, 1, 32).astype(np.float32),
           "half": rng.randn(3).astype(np.float16), "flag": np.asarray([True, False])}
    prefix = str(tmp_path / "logdir" / "model.ckpt-123456")
    ck.write_bundle(prefix, var)
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(np.asarray(v).nbytes for v in var.values())
    got = ck.read_bundle(prefix, verify=True)
    assert sorted(got) == sorted(var)
    for k in var:
        assert got[k].dtype == np.asarray(var[k]).dtype and got[k].shape == np.asarray(var[k]).shape
        np.testing.assert_array_equal(got[k], var[k])
    keys = [k for k, _ in ck.read_table(prefix + ".index")]
    assert keys[0] == b"" and keys == sorted(keys)
    only = ck.read_bundle(prefix, names={"global_step"})
    assert list(only) == ["global_step"] and int(only["global_step"]) == 123456
    # flip one data byte: verify=True raises, verify=False warns, verify=None is silent
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); raw[len(raw) // 2] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ck.CheckpointError):
        ck.read_bundle(prefix, verify=True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ck.read_bundle(prefix, verify=False)
        assert any("crc32c" in str(x.message) for x in w)
    ck.read_bundle(prefix, verify=None)
    with pytest.raises(ck.CheckpointError):
        ck.write_bundle(prefix, {"s": np.asarray(["a"])})


def test_checkpoint_state_and_discovery(tmp_path):
    d = str(tmp_path)
    assert ck.latest_checkpoint(d) is None
    with pytest.raises(ck.CheckpointError):
        ck.resolve(d)
    for step in (1000, 24000, 3000):
        ck.write_bundle(os.path.join(d, "model.ckpt-%d" % step), {"global_step": np.asarray(step, np.int32)})
    ck.write_checkpoint_state(d, os.path.join(d, "model.ckpt-3000"), [os.path.join(d, "model.ckpt-%d" % s) for s in (1000, 24000, 3000)])
    txt = open(os.path.join(d, "checkpoint")).read().splitlines()
    assert txt[0] == 'model_checkpoint_path: "model.ckpt-3000"' and txt[1] == 'all_model_checkpoint_paths: "model.ckpt-1000"'
    assert ck.latest_checkpoint(d) == os.path.join(d, "model.ckpt-3000")               # the state file decides (utils/__init__.py:78)
    assert ck.most_recent_checkpoint(d) == os.path.join(d, "model.ckpt-24000")         # the largest step decides (tacotron/__init__.py:11)
    assert ck.checkpoint_step(ck.latest_checkpoint(d)) == 3000
    assert ck.resolve(d) == os.path.join(d, "model.ckpt-3000")
    assert ck.resolve(os.path.join(d, "model.ckpt-1000.index")) == os.path.join(d, "model.ckpt-1000")
    assert ck.resolve(os.path.join(d, "model.ckpt-1000.data-00000-of-00001")) == os.path.join(d, "model.ckpt-1000")


def test_wavenet_variable_selection(tmp_path):
    from twvk_amd import weights as W
    specs = W.tensor_specs(2, S=64, gc_cardinality=3)
    t = W.random_tensors(specs, seed=3)
    var = dict(t)
    var.update({k + ck.EMA_SUFFIX: v + 1 for k, v in t.items()})
    var["wavenet/queue/causal_queue"] = np.zeros((1, 32, 1), np.float32)               # state: never restored (generate.py:157)
    var["optimizer/wavenet/conv1d/kernel/Adam"] = np.zeros((32, 1, 32), np.float32)
    prefix = str(tmp_path / "model.ckpt-7")
    ck.write_bundle(prefix, var)
    got = ck.wavenet_tensors(ck.read_bundle(prefix, verify=True), specs)
    assert sorted(got) == sorted(t)
    np.testing.assert_array_equal(W.flatten(specs, got), W.flatten(specs, t))
    ema = ck.wavenet_tensors(ck.read_bundle(prefix), specs, use_ema=True)
    np.testing.assert_array_equal(ema["wavenet/gc_embedding"], t["wavenet/gc_embedding"] + 1)
    del var["wavenet/conv1d_2/bias"]
    ck.write_bundle(prefix, var)
    with pytest.raises(ck.CheckpointError, match="conv1d_2/bias"):
        ck.wavenet_tensors(ck.read_bundle(prefix), specs)


def test_tacotron_variable_names_round_trip(tmp_path):
    from twvk_amd.hparams import hparams
    from twvk_amd import tacotron as T
    specs = T.tacotron_specs(hparams, 2)
    rng = np.random.RandomState(0)
    t = {n: np.abs(rng.randn(*s)).astype(np.float32) for n, s in specs}
    var = ck.tacotron_variables(t)
    assert "model/inference/encoder_cbhg/conv_bank/conv1d_1/batch_normalization/moving_variance" in var
    assert "model/inference/embedding" in var and all(k.startswith("model/inference/") for k in var)
    prefix = str(tmp_path / "model.ckpt-1")
    ck.write_bundle(prefix, var)
    got = ck.tacotron_tensors(ck.read_bundle(prefix, verify=True), specs)
    np.testing.assert_array_equal(T.flatten(specs, got), T.flatten(specs, t))
    del var["model/inference/memory_layer/kernel"]
    with pytest.raises(ck.CheckpointError, match="memory_layer"):
        ck.tacotron_tensors(var, specs)


def test_sharded_bundle_and_unsupported_entries(tmp_path):
    """tensors dealt over three data shards read back; partitioned (sliced) variables and missing shards are reported, not mis-read"""
    rng = np.random.RandomState(3)
    var = {"v%02d" % i: rng.randn(3, i + 1).astype(np.float32) for i in range(7)}
    var["wide/shape"] = rng.randn(130, 3, 1, 2).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt-5")
    ck.write_bundle(prefix, var, num_shards=3)
    assert sorted(os.listdir(str(tmp_path))) == ["model.ckpt-5.data-0000%d-of-00003" % k for k in range(3)] + ["model.ckpt-5.index"]
    got = ck.read_bundle(prefix, verify=True)
    assert sorted(got) == sorted(var) and all(np.array_equal(got[k], var[k]) for k in var)
    e = ck._parse_entry(ck._entry_bytes(1, (2, 2), 16, 16, 5, shard_id=2))
    assert (e["shard_id"], e["offset"], e["size"], e["shape"]) == (2, 16, 16, [2, 2])
    os.remove(prefix + ".data-00001-of-00003")
    with pytest.raises(FileNotFoundError):
        ck.read_bundle(prefix)
    assert sorted(ck.read_bundle(prefix, names={"v00", "v03", "v06"})) == ["v00", "v03", "v06"]     # shard 0 only: still readable
    # a partitioned variable (BundleEntryProto.slices, field 7) is refused by name
    items = [(b"", b"\x08\x01\x1a\x02\x08\x01"), (b"part", ck._entry_bytes(1, (4,), 0, 16, 0) + b"\x3a\x00")]
    ck.write_table(str(tmp_path / "p.index"), items)
    open(str(tmp_path / "p.data-00000-of-00001"), "wb").write(bytes(16))
    with pytest.raises(ck.CheckpointError, match="partitioned"):
        ck.read_bundle(str(tmp_path / "p"))


def _shift_auto_names(variables, base, delta, scope="model/inference/"):
    """rename every top-level `<scope><base>[_N]/...` variable to `<base>_{N+delta}` (N = 0 for the bare name)"""
    out = {}
    for k, v in variables.items():
        rest = k[len(scope):] if k.startswith(scope) else None
        if rest is not None:
            head, _, tail = rest.partition("/")
            b, i = ck._split_auto(head)
            if b == base and i is not None:
                n = i + delta
                out[scope + (base if n == 0 else "%s_%d" % (base, n)) + "/" + tail] = v
                continue
        out[k] = v
    return out


def test_restore_tolerates_shifted_auto_generated_names(tmp_path, capsys):
    """the names TensorFlow auto-generates (dense, dense_1, ...) are this project's RECOLLECTION of its numbering
    rule.  A checkpoint whose graph created one more unnamed dense layer first carries the same tensors as dense_1 .. dense_6: the
    restore falls back to (scope path without the suffix, shape, creation order), lists what it remapped, and loads the same model."""
    from twvk_amd.hparams import hparams as hp
    from twvk_amd.tacotron import tacotron_specs
    specs = tacotron_specs(hp, 2)
    rng = np.random.RandomState(0)
    tensors = {n: rng.randn(*s).astype(np.float32) for n, s in specs}
    variables = ck.tacotron_variables(tensors)
    # plus what a training checkpoint also holds: Adam slots and the global step
    extra = {k + "/Adam": np.zeros_like(v) for k, v in list(variables.items())[:5]}
    extra["global_step"] = np.array(7, np.int64)
    shifted = _shift_auto_names(variables, "dense", +1)
    assert "model/inference/dense_6/kernel" in shifted and "model/inference/dense/kernel" not in shifted
    assert "model/inference/prenet/dense_1/kernel" in shifted           # explicitly named layers (modules.py:20) are not touched
    ck.write_bundle(str(tmp_path / "model.ckpt-1"), dict(shifted, **extra))
    wanted = ck.tacotron_variable_specs(specs)
    got = ck.tacotron_tensors(ck.restore_variables(str(tmp_path / "model.ckpt-1"), wanted), specs)
    for n, _ in specs:
        assert np.array_equal(got[n], tensors[n]), n
    report = capsys.readouterr().out
    assert "12 variables restored under a different auto-generated name" in report
    assert "model/inference/dense/kernel  <-  model/inference/dense_1/kernel" in report
    assert "model/inference/dense_5/bias  <-  model/inference/dense_6/bias" in report
    # the unshifted checkpoint restores silently
    ck.write_bundle(str(tmp_path / "model.ckpt-2"), variables)
    got2 = ck.restore_variables(str(tmp_path / "model.ckpt-2"), wanted)
    assert capsys.readouterr().out == "" and set(got2) == set(n for n, _ in wanted)
    # single-speaker graph (tacotron.py:97-104): its only unnamed dense layer is the linear projection, "dense"; a checkpoint that
    # calls it dense_3 still loads, one that lacks it does not
    specs1 = tacotron_specs(hp, 1)
    t1 = {n: rng.randn(*s).astype(np.float32) for n, s in specs1}
    v1 = _shift_auto_names(ck.tacotron_variables(t1), "dense", +3)
    ck.write_bundle(str(tmp_path / "model.ckpt-3"), v1)
    g1 = ck.tacotron_tensors(ck.restore_variables(str(tmp_path / "model.ckpt-3"), ck.tacotron_variable_specs(specs1), log=lambda m: None), specs1)
    assert np.array_equal(g1["dense/kernel"], t1["dense/kernel"])
    del v1["model/inference/dense_3/kernel"]
    ck.write_bundle(str(tmp_path / "model.ckpt-4"), v1)
    with pytest.raises(ck.CheckpointError, match="lacks 1 tensors"):
        ck.restore_variables(str(tmp_path / "model.ckpt-4"), ck.tacotron_variable_specs(specs1))
    # an ambiguous group (two candidates of the same stem and shape for one wanted tensor) is refused, not guessed
    v2 = dict(ck.tacotron_variables(t1))
    lin = v2.pop("model/inference/dense/kernel")
    v2["model/inference/dense_1/kernel"] = lin
    v2["model/inference/dense_2/kernel"] = lin + 1
    ck.write_bundle(str(tmp_path / "model.ckpt-5"), v2)
    with pytest.raises(ck.CheckpointError):
        ck.restore_variables(str(tmp_path / "model.ckpt-5"), ck.tacotron_variable_specs(specs1))


def test_restore_by_name_inside_a_group_of_another_size(tmp_path):
    """The checkpoint's graph created ONE MORE unnamed dense layer of the same shape (its `dense` is that extra layer, its `dense_1` the
    graph's `dense`).  tf.train.Saver.restore (generate.py:157-161, synthesizer.py:69-70) matches by NAME only and loads `dense` without
    a word; so does restore_variables -- but it says so -- and strict=True refuses the group with a message that names
    the ambiguity (ADVICE r04's concern: the name may hold a neighbour's tensor)."""
    from twvk_amd.hparams import hparams as hp
    from twvk_amd.tacotron import tacotron_specs
    specs1 = tacotron_specs(hp, 1)
    rng = np.random.RandomState(3)
    t1 = {n: rng.randn(*s).astype(np.float32) for n, s in specs1}
    v = dict(ck.tacotron_variables(t1))
    lin_k, lin_b = v["model/inference/dense/kernel"], v["model/inference/dense/bias"]
    v["model/inference/dense_1/kernel"], v["model/inference/dense_1/bias"] = lin_k * 0 + 7, lin_b * 0 + 7     # an extra same-shape layer
    ck.write_bundle(str(tmp_path / "model.ckpt-1"), v)
    said = []
    got = ck.tacotron_tensors(ck.restore_variables(str(tmp_path / "model.ckpt-1"), ck.tacotron_variable_specs(specs1), log=said.append), specs1)
    assert np.array_equal(got["dense/kernel"], t1["dense/kernel"])                  # by name, as TensorFlow does
    assert any("WARNING matched by name although the group sizes differ" in m and "the graph wants 1, the checkpoint holds 2" in m for m in said), said
    with pytest.raises(ck.CheckpointError, match="ambiguous group size"):
        ck.restore_variables(str(tmp_path / "model.ckpt-1"), ck.tacotron_variable_specs(specs1), strict=True)
    # a group of another size WITHOUT all the wanted names is refused either way, and the message says why
    v2 = dict(ck.tacotron_variables(t1))
    lin = v2.pop("model/inference/dense/kernel")
    v2["model/inference/dense_1/kernel"] = lin
    v2["model/inference/dense_2/kernel"] = lin + 1
    ck.write_bundle(str(tmp_path / "model.ckpt-2"), v2)
    with pytest.raises(ck.CheckpointError, match="ambiguous group size"):
        ck.restore_variables(str(tmp_path / "model.ckpt-2"), ck.tacotron_variable_specs(specs1))


def test_wavenet_restore_tolerates_shifted_conv1d_names(tmp_path):
    """generate.py:157-161: wavenet/conv1d (causal), conv1d_1, conv1d_2 (post-processing) are auto-numbered tf.layers.conv1d layers"""
    from twvk_amd import weights as W
    specs = W.tensor_specs(3)
    tensors = W.random_tensors(specs, seed=1, scale=0.1)
    shifted = {}
    for k, v in tensors.items():
        parts = k.split("/")
        b, i = ck._split_auto(parts[1])
        if b == "conv1d" and i is not None and len(parts) == 3:
            parts[1] = "conv1d_%d" % (i + 2)
        shifted["/".join(parts)] = v
    assert "wavenet/conv1d_2/kernel" in shifted and "wavenet/conv1d/kernel" not in shifted
    ck.write_bundle(str(tmp_path / "model.ckpt-9"), shifted)
    notes = []
    got = ck.wavenet_tensors(ck.restore_variables(str(tmp_path / "model.ckpt-9"), specs, log=notes.append), specs)
    for n, _ in specs:
        assert np.array_equal(got[n], tensors[n]), n
    assert any("wavenet/conv1d/kernel  <-  wavenet/conv1d_2/kernel" in m for m in notes)


@pytest.mark.parametrize("kind", ["wavenet", "tacotron"])
def test_the_saver_bundle_test_runs_on_a_bundle_of_the_same_layout(tmp_path, monkeypatch, kind):
    """tests/test_reference_goldens.py::test_reference_saver_bundle_reads_back_bit_for_bit cannot run here (it needs a bundle written by
    TensorFlow: scripts/make_reference_goldens.py --only ckpt).  So that it does not rot unexecuted, the same test function is driven
    here over a directory of the same layout written by THIS repo's writer with the names this repo expects: it proves the test's own
    logic, not the importer's fidelity to TensorFlow."""
    import json
    import twvk_amd
    import test_reference_goldens as T
    rng = np.random.RandomState(5)
    if kind == "wavenet":
        from twvk_amd import weights as W
        dims = dict(dilations=[1, 2, 4, 8, 1, 2, 4, 8], residual_channels=32, dilation_channels=32, skip_channels=128, quantization_channels=256,
                    out_channels=30, scalar_input=True, initial_filter_width=32, gc_channels=32, gc_cardinality=2, lc_channels=80, upsample_factor=[5, 5, 12])
        specs = W.tensor_specs(8, S=128)
        variables = {n: rng.randn(*s).astype(np.float32) for n, s in specs}
        step = 7
    else:
        from twvk_amd.tacotron import tacotron_specs
        dims = dict(embedding_size=32, enc_prenet_sizes=[32, 16], enc_bank_size=4, enc_bank_channel_size=16, enc_proj_sizes=[16, 16],
                    enc_rnn_size=16, attention_size=32, attention_state_size=32, dec_rnn_size=32, dec_prenet_sizes=[32, 16],
                    post_bank_size=3, post_bank_channel_size=16, post_proj_sizes=[32, 80], post_rnn_size=16, num_freq=129, max_iters=4,
                    n_symbols=80, num_speakers=2)
        hp = twvk_amd.default_hparams()
        for k, v in dims.items():
            if hasattr(hp, k):
                setattr(hp, k, v)
        specs = tacotron_specs(hp, 2)
        variables = ck.tacotron_variables({n: rng.randn(*s).astype(np.float32) for n, s in specs})
        step = 3
    variables["global_step"] = np.array(step, np.int64)
    d = tmp_path / ("reference_ckpt_" + kind)
    d.mkdir()
    prefix = str(d / ("model.ckpt-%d" % step))
    ck.write_bundle(prefix, variables)
    ck.write_checkpoint_state(str(d), prefix)
    names = sorted(variables)
    np.savez_compressed(str(d / "values.npz"), **{"t%d" % i: variables[n] for i, n in enumerate(names)})
    json.dump({"prefix": os.path.basename(prefix), "names": names, "dims": dims}, open(str(d / "values.json"), "w"))
    monkeypatch.setattr(T, "GOLD", str(tmp_path))
    T.test_reference_saver_bundle_reads_back_bit_for_bit(kind)
