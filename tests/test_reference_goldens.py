"""Holds the CPU oracle to outputs of the REFERENCE ITSELF -- when they exist.

tests/golden/reference_*.npz are written by scripts/make_reference_goldens.py on a machine that has TensorFlow 1.x and a checkout
of the reference (neither exists in the build image: BASELINE.md section 2); every test here SKIPS while they are absent, so the
parity status stays "unpinned against TensorFlow" until someone runs that script once.  With the files present: integer class ids
bit-exact, floats within 1e-4 (north_star's bars), and the GPU path is held to the oracle bit for bit by the -m gpu suites."""
import json
import os

import numpy as np
import pytest

from helpers import make_case, first_mismatch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def _load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip("%s is absent: run scripts/make_reference_goldens.py where TensorFlow 1.x and the reference exist" % name)
    return np.load(path)


def test_reference_codec(oracle):
    f = _load("reference_codec.npz")
    assert np.array_equal(oracle.mu_law_encode(f["audio"], 256), f["q"])                         # ops.py:22-33, integers: bit-exact
    assert np.abs(oracle.mu_law_decode(np.arange(256, dtype=np.int32), 256) - f["dec"]).max() <= TOL
    assert np.abs(oracle.mu_law_expand(f["expand_in"], 256) - f["expand"]).max() <= TOL


def test_reference_wavenet_mol_raw_outputs(oracle):
    g = _load("reference_wavenet_mol_small.npz")
    dil = [int(v) for v in g["dilations"]]
    d, tensors, blob = make_case(oracle, dil, S=int(g["S"]), scale=float(g["scale"]), seed=int(g["weight_seed"]))
    assert oracle.receptive_field(d) == int(g["receptive_field"])                                # model.py:31-39
    U = oracle.upsample(d, blob, g["mel"])
    assert np.abs(U - g["upsampled"]).max() <= TOL                                               # model.py:102-111 ('same' alignment of the width-2 tap)
    forced = g["forced"]
    B, T = forced.shape
    st = oracle.State(d, B)
    raws = np.stack([oracle.step(d, blob, st, forced[:, t], U[:, t], g["gc_ids"]) for t in range(T)], axis=1)
    assert np.abs(raws - g["raw_incremental"]).max() <= TOL                                      # model.py:112-167 incremental (lc front slice included)
    rf = oracle.receptive_field(d)
    xin = np.concatenate([np.zeros((B, rf - 1), np.float32), forced], axis=1)
    full = oracle.forward_full(d, blob, xin, U[:, :xin.shape[1]], g["gc_ids"])
    assert np.abs(full - g["raw_full"]).max() <= TOL


def test_reference_wavenet_mulaw_class_ids(oracle):
    g = _load("reference_wavenet_mulaw_small.npz")
    dil = [int(v) for v in g["dilations"]]
    d, tensors, blob = make_case(oracle, dil, scalar_input=False, S=int(g["S"]), Q=int(g["Q"]), scale=float(g["scale"]), seed=int(g["weight_seed"]))
    for temp, key in ((1.0, "t10"), (0.8, "t08")):
        out = oracle.generate_mulaw(d, blob, oracle.State(d, 2), g["upsampled"], g["gc_ids"], g["first_input"], g["uniforms_" + key], temp)
        assert np.array_equal(out, g["samples_" + key]), (key, first_mismatch(out, g["samples_" + key]))   # integers: bit-exact


@pytest.mark.parametrize("name", ["reference_tacotron_small.npz", "reference_tacotron_small_single_speaker.npz"])
def test_reference_tacotron(oracle, name):
    t = _load(name)
    d = oracle.taco_dims(n_symbols=int(t["n_symbols"]), n_speakers=int(t["n_speakers"]), enc_bank=int(t["dims_enc_bank"]),
                         post_bank=int(t["dims_post_bank"]), max_iters=int(t["dims_max_iters"]), num_freq=int(t["dims_num_freq"]))
    blob = oracle.taco_blob(d, oracle.taco_random_tensors(d, seed=int(t["weight_seed"])))
    mel, lin, al = oracle.taco_infer(d, blob, t["tokens"], t["lengths"], t["speaker_ids"])
    assert np.abs(mel - t["mel"]).max() <= TOL                                                   # north_star: 1e-4 on float mel frames
    assert np.abs(lin - t["linear"]).max() <= TOL
    assert np.abs(al - t["alignments"]).max() <= TOL


def test_reference_variable_names():
    """what TensorFlow really named the variables vs the names weights.py / tacotron.py recall (checkpoint rows a20, f4)"""
    path = os.path.join(GOLD, "reference_variable_names.json")
    if not os.path.exists(path):
        pytest.skip("reference_variable_names.json is absent (scripts/make_reference_goldens.py)")
    rep = json.load(open(path))
    renamed = {k: v["renamed"] for k, v in rep.items() if isinstance(v, dict) and v.get("renamed")}
    assert not renamed, "TensorFlow names differ from the recalled ones: %r" % renamed


@pytest.mark.parametrize("kind", ["wavenet", "tacotron"])
def test_reference_saver_bundle_reads_back_bit_for_bit(kind):
    """a bundle written by tf.train.Saver ITSELF (scripts/make_reference_goldens.py, `ckpt`) through
    checkpoint.read_bundle (CRC-32C verified) == the values TensorFlow read back from its own variables, every tensor, every dtype;
    and generate.py:157-161 / synthesizer.py:69-70's restore -- the names weights.py / tacotron.py expect -- finds every tensor
    (by name or through remap_names) and returns the same bits."""
    import sys
    d = os.path.join(GOLD, "reference_ckpt_" + kind)
    if not os.path.exists(os.path.join(d, "values.json")):
        pytest.skip("tests/golden/reference_ckpt_%s is absent: run scripts/make_reference_goldens.py --only ckpt where TensorFlow 1.x and the reference exist" % kind)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import twvk_amd                                     # noqa: F401
    from twvk_amd import checkpoint as ck
    meta = json.load(open(os.path.join(d, "values.json")))
    vals = np.load(os.path.join(d, "values.npz"))
    truth = {n: vals["t%d" % i] for i, n in enumerate(meta["names"])}
    prefix = os.path.join(d, meta["prefix"])
    assert ck.latest_checkpoint(d) is not None and os.path.basename(ck.latest_checkpoint(d)) == meta["prefix"]     # the `checkpoint` state file
    got = ck.read_bundle(prefix, verify=True)
    assert sorted(got) == sorted(truth)
    for n, a in truth.items():
        assert got[n].dtype == a.dtype and got[n].shape == a.shape and got[n].tobytes() == np.ascontiguousarray(a).tobytes(), n
    shapes = ck.bundle_shapes(prefix)
    assert shapes == {n: tuple(a.shape) for n, a in truth.items() if a.dtype == np.float32 or n in shapes}
    dims = meta["dims"]
    if kind == "wavenet":
        from twvk_amd import weights as W
        wanted = W.tensor_specs(len(dims["dilations"]), R=dims["residual_channels"], D=dims["dilation_channels"], S=dims["skip_channels"],
                                Q=dims["quantization_channels"], out_channels=dims["out_channels"], scalar_input=dims["scalar_input"],
                                initial_filter_width=dims["initial_filter_width"], gc_channels=dims["gc_channels"],
                                gc_cardinality=dims["gc_cardinality"], lc_channels=dims["lc_channels"], upsample_factor=tuple(dims["upsample_factor"]))
    else:
        from twvk_amd.tacotron import tacotron_specs
        hp = twvk_amd.default_hparams()
        for k, v in dims.items():
            if hasattr(hp, k):
                setattr(hp, k, v)
        wanted = ck.tacotron_variable_specs(tacotron_specs(hp, dims["num_speakers"], n_symbols=dims["n_symbols"]))
    notes = []
    restored = ck.restore_variables(prefix, wanted, verify=True, log=notes.append)
    mapping = ck.remap_names([(n, tuple(s)) for n, s in wanted], shapes)
    assert sorted(restored) == sorted(n for n, _ in wanted)
    for n, _ in wanted:
        assert np.array_equal(restored[n], truth[mapping[n]]), (n, mapping[n])
    # nothing of the model is left over in the bundle (only the global step may be)
    left = sorted(set(truth) - set(mapping.values()) - {"global_step"})
    assert not left, "variables of the reference graph this repo does not know: %s" % left
