"""-m "not gpu": the oracle against known answers / independent float64 references / its committed restatement
fixtures, the host logic, and the C-ABI library's exported symbols (no compute calls without a GPU)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import first_mismatch, make_case, mol_uniforms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------- known answers derivable from the reference text
def test_receptive_fields(oracle):
    d50, d30 = [2 ** i for i in range(10)] * 5, [2 ** i for i in range(10)] * 3
    # generate.py:192 comment: 5117 for the 50-layer mu-law model; the rest by model.py:31-39
    for dil, scalar, want in ((d50, False, 5117), (d50, True, 5147), (d30, False, 3071), (d30, True, 3101)):
        assert oracle.receptive_field(oracle.make_dims(dil, scalar_input=scalar)) == want
    import twvk_amd
    from twvk_amd.wavenet import WaveNetModel
    assert WaveNetModel.calculate_receptive_field(2, d50, False, 32) == 5117
    assert WaveNetModel.calculate_receptive_field(2, d30, True, 32) == 3101


def test_mu_law_known_answers(oracle):
    q = oracle.mu_law_encode(np.array([0.0, 1.0, -1.0, 2.0, -3.0], np.float32), 256)
    assert list(q) == [128, 255, 0, 255, 0]          # ops.py:22-33; generate.py:190 silence = Q/2
    a = np.linspace(-1, 1, 4001).astype(np.float32)
    back = oracle.mu_law_decode(oracle.mu_law_encode(a, 256), 256)
    # round trip: at most half a quantisation step of the companded signal
    comp = np.sign(a) * np.log1p(255 * np.abs(a)) / np.log1p(255)
    comp_back = np.sign(back) * np.log1p(255 * np.abs(back)) / np.log1p(255)
    assert np.abs(comp - comp_back).max() <= 1.0 / 255 + 1e-5
    assert np.allclose(oracle.mu_law_expand(comp.astype(np.float32), 256), a, atol=2e-5)


def test_hparams_surface(tmp_path):
    import twvk_amd
    hp = twvk_amd.default_hparams()
    assert int(np.prod(hp.upsample_factor)) == hp.hop_size == 300        # hparams.py:79
    assert hp.num_freq == 1025 and hp.frame_shift_ms == 12.5 and hp.frame_length_ms == 50.0   # hparams.py:190-192
    assert (hp.residual_channels, hp.dilation_channels, hp.skip_channels, hp.out_channels) == (32, 32, 512, 30)
    assert len(hp.dilations) == 50 and hp.scalar_input and hp.input_type == "raw" and hp.gc_channels == 32
    # params.json round trip (utils/__init__.py:143-172): unknown keys skipped, known keys overridden
    hp.dilations = [1, 2, 4]
    twvk_amd.save_hparams(str(tmp_path), hp)
    data = json.load(open(tmp_path / "params.json"))
    data["not_a_hparam"] = 1
    data["sample_rate"] = 16000
    json.dump(data, open(tmp_path / "params.json", "w"))
    hp2 = twvk_amd.load_hparams(twvk_amd.default_hparams(), str(tmp_path))
    assert hp2.dilations == [1, 2, 4] and hp2.sample_rate == 16000 and not hasattr(hp2, "not_a_hparam")


# ---------------------------------------------------------------- elementary functions of the arithmetic contract
@pytest.mark.parametrize("name,ref,lo,hi,tol", [
    ("tanh", np.tanh, -12, 12, 3e-7), ("sigmoid", lambda v: 1 / (1 + np.exp(-v)), -25, 25, 2e-7)])
def test_rationals_accuracy(oracle, name, ref, lo, hi, tol):
    x = np.linspace(lo, hi, 50001).astype(np.float32)
    assert np.abs(oracle.elementwise(name, x) - ref(x.astype(np.float64))).max() < tol


def test_exp_log_accuracy(oracle):
    x = np.linspace(-87, 88, 50001).astype(np.float32)
    e = oracle.elementwise("exp", x).astype(np.float64)
    assert (np.abs(e - np.exp(x.astype(np.float64))) / np.exp(x.astype(np.float64))).max() < 2e-7
    xp = np.exp(np.linspace(-80, 80, 50001)).astype(np.float32)
    l = oracle.elementwise("log", xp).astype(np.float64)
    assert np.abs(l - np.log(xp.astype(np.float64))).max() < 5e-6
    xd = np.linspace(-700, 700, 20001)
    assert (np.abs(oracle.elementwise("exp64", xd) - np.exp(xd)) / np.exp(xd)).max() < 4e-16
    xl = np.exp(np.linspace(-700, 700, 20001))
    assert np.abs(oracle.elementwise("log64", xl) - np.log(xl)).max() < 2e-13


def test_cdot_is_the_contract(oracle):
    """AC-1 spelled out in numpy float32 (fma emulated in float64: exact for a product plus an addend)"""
    rng = np.random.RandomState(0)
    for K in (2, 16, 32, 80, 512):
        w = rng.randn(K).astype(np.float32); x = rng.randn(K).astype(np.float32)
        r = None
        for k0 in range(0, K, 32):
            s = [np.float32(0)] * 4
            for k in range(k0, min(k0 + 32, K)):
                s[(k - k0) & 3] = np.float32(np.float64(w[k]) * np.float64(x[k]) + np.float64(s[(k - k0) & 3]))
            a = np.float32(np.float32(s[0] + s[1]) + np.float32(s[2] + s[3]))
            r = a if r is None else np.float32(r + a)
        got = oracle.lib().twvo_cdot(w.ctypes.data_as(C.POINTER(C.c_float)), 1, x.ctypes.data_as(C.POINTER(C.c_float)), K)
        assert np.float32(got) == r, K


# ---------------------------------------------------------------- the network
@pytest.mark.parametrize("scalar", [True, False])
def test_incremental_equals_full_convolution(oracle, scalar):
    """the reference's design intent (model.py:236 comment; fast-wavenet): once the delay lines are warm, the
    incremental network reproduces the full 'valid' convolution -- here bit for bit"""
    d = oracle.make_dims([1, 2, 4, 1, 2, 4], S=64, Q=16, out_channels=30, scalar_input=scalar, ifw=8, G=32, gc_card=2, L=0)
    rf = oracle.receptive_field(d)
    T, B = rf + 40, 2
    blob = oracle.blob_from_tensors(d, oracle.random_tensors(d, seed=3, scale=0.2))
    rng = np.random.RandomState(1)
    inp = rng.uniform(-1, 1, (B, T)).astype(np.float32) if scalar else rng.randint(0, 16, (B, T)).astype(np.int32)
    gc = np.array([0, 1], np.int32)
    full = oracle.forward_full(d, blob, inp, None, gc)
    st = oracle.State(d, B)
    inc = np.stack([oracle.step(d, blob, st, inp[:, t], None, gc) for t in range(T)], 1)[:, rf - 1:]
    assert full.shape == (B, T - rf + 1, d.O)
    assert np.array_equal(inc, full)


def test_lc_uses_previous_frame(oracle):
    """model.py:79-80 slices the lc projection from the FRONT of the 2-deep queue: step t sees the frame pushed at t-1"""
    d, tensors, blob = make_case(oracle, [1, 2], S=64, G=0, scale=0.3)
    B = 1
    x = np.array([0.3], np.float32)
    f0 = np.full((B, 80), 0.5, np.float32); f1 = np.full((B, 80), -0.7, np.float32)
    st = oracle.State(d, B); a0 = oracle.step(d, blob, st, x, f0); a1 = oracle.step(d, blob, st, x, f1)
    st = oracle.State(d, B); b0 = oracle.step(d, blob, st, x, f1); b1 = oracle.step(d, blob, st, x, f1)
    assert np.array_equal(a0, b0)            # step 0 ignores the frame pushed at step 0 (queue front is still zeros)
    assert not np.array_equal(a1, b1)        # step 1 sees the frame pushed at step 0


def test_upsample_shape_and_linearity(oracle):
    d, tensors, blob = make_case(oracle, [1], S=64)
    mel = np.random.RandomState(2).uniform(-4, 4, (2, 5, 80)).astype(np.float32)
    up = oracle.upsample(d, blob, mel)
    assert up.shape == (2, 5 * 300, 80)                       # generate.py:152
    # every output row depends on exactly one input frame (kernel == stride along time)
    mel2 = mel.copy(); mel2[:, 3] += 1.0
    diff = np.abs(oracle.upsample(d, blob, mel2) - up).sum(axis=(0, 2))
    assert np.all(diff[:900] == 0) and np.all(diff[1200:] == 0) and np.all(diff[900:1200] > 0)


def test_mol_sampler_against_float64_formula(oracle):
    """mixture.py:84-114 evaluated independently in float64"""
    rng = np.random.RandomState(4)
    for _ in range(200):
        y = rng.randn(30).astype(np.float32) * 2
        u = rng.uniform(1e-5, 1 - 1e-5, 11).astype(np.float32)
        got = oracle.sample_mol(y, u)
        y64, u64 = y.astype(np.float64), u.astype(np.float64)
        g = y64[:10] - np.log(-np.log(u64[:10]))
        order = np.sort(g)
        if order[-1] - order[-2] < 1e-4:
            continue                                     # argmax too close to call in float32
        k = int(np.argmax(g))
        ls = max(y64[20 + k], np.log(1e-14))
        want = np.clip(y64[10 + k] + np.exp(ls) * (np.log(u64[10]) - np.log(1 - u64[10])), -1, 1)
        assert abs(got - want) < 1e-4 * max(1.0, np.exp(ls)), (got, want)


def test_categorical_sampler_matches_numpy_legacy_choice(oracle):
    """generate.py:219-231: float64 softmax -> float32, temperature rescale, np.random.choice == searchsorted(cumsum)"""
    rng = np.random.RandomState(5)
    agree = 0
    for trial in range(300):
        logits = (rng.randn(256) * 3).astype(np.float32)
        temp = [1.0, 0.7, 1.3][trial % 3]
        x = logits.astype(np.float64)
        p = np.exp(x - x.max()); p = (p / p.sum()).astype(np.float32)              # model.py:243
        with np.errstate(divide="ignore"):
            sp = np.log(p) / temp                                                      # generate.py:220
            sp = sp - np.logaddexp.reduce(sp, axis=-1, keepdims=True)                  # generate.py:221
            sp = np.exp(sp)                                                            # generate.py:222
        if temp == 1.0:
            assert np.allclose(p, sp, atol=1e-5)                                       # generate.py:227-228
        rs = np.random.RandomState(trial)
        u = np.random.RandomState(trial).random_sample()
        want = rs.choice(np.arange(256), p=sp)                                         # generate.py:231
        got, proba = oracle.sample_categorical(logits, temp, u)
        assert np.allclose(proba, sp, rtol=2e-5, atol=1e-9)
        agree += int(got == want)
    assert agree >= 298        # the two may differ only when u falls within float32 noise of a cdf boundary


def test_scan64_is_a_prefix_sum_in_the_documented_tree_order(oracle):
    """AC-5's one reduction primitive: exact prefix sums on integers, and the documented row_shr / row_bcast tree on floats"""
    v = np.arange(1, 65, dtype=np.float64)
    assert np.array_equal(oracle.scan64(v), np.cumsum(v))
    rng = np.random.RandomState(3)
    v = rng.uniform(0, 1, 64) * 10.0 ** rng.randint(-12, 3, 64)
    w = v.copy()
    for off in (1, 2, 4, 8):
        t = w.copy()
        for l in range(64):
            if (l & 15) >= off:
                w[l] = t[l] + t[l - off]
    w[16:32] += w[15]; w[48:64] += w[47]
    w[32:64] += w[31]
    got = oracle.scan64(v)
    assert np.array_equal(got, w)
    assert np.allclose(got, np.cumsum(v), rtol=1e-14)


def test_categorical_sampler_contract_against_numpy(oracle):
    """VERDICT r03 next-1(i).  generate.py:219-231 is pure numpy -- the one piece of the reference path that RUNS in this image --
    so the sampler contract (AC-5) is founded on it: the lines are executed literally (float32 `np.log(p) / T`, left-to-right
    float32 `np.logaddexp.reduce`, `np.exp`, then legacy `RandomState.choice` = float64 cumsum / last / searchsorted 'right')
    over 100 000 rows per temperature, and both forms of the oracle's sampler are held against it:
      * round 3's all-sequential form (`sequential=True`), which reproduced the ORDER of numpy's reduce with the contract's
        exp/log -- its scaled probabilities still differ from numpy's by a few 1e-6 (libm/SIMD vs Cephes), so the order pinned nothing;
      * the current form (max-shifted log-sum-exp, every sum in the lane-then-scan64 tree), which the kernels implement.
    The current form must agree with numpy on the drawn class at least as often as the sequential one."""
    Q, N = 256, 100_000
    rng = np.random.RandomState(11)
    lib = oracle.lib()
    import ctypes as C
    fp = C.POINTER(C.c_float)
    report = {}
    for temp in (1.0, 0.8, 1.3):
        logits = (rng.randn(N, Q) * rng.uniform(0.5, 4.0, (N, 1))).astype(np.float32)
        x = logits.astype(np.float64)
        p = np.exp(x - x.max(axis=1, keepdims=True)); p = (p / p.sum(axis=1, keepdims=True)).astype(np.float32)   # model.py:243
        with np.errstate(divide="ignore"):
            sp = np.log(p) / temp                                                      # generate.py:220
            sp = sp - np.logaddexp.reduce(sp, axis=-1, keepdims=True)                  # generate.py:221
            sp = np.exp(sp)                                                            # generate.py:222
        assert sp.dtype == np.float32
        if temp == 1.0:
            np.testing.assert_allclose(p, sp, atol=1e-5)                               # generate.py:227-228
        u = rng.random_sample(N)
        cdf = sp.astype(np.float64).cumsum(axis=1)                                     # RandomState.choice (legacy)
        cdf /= cdf[:, -1:]
        want = (cdf > u[:, None]).argmax(axis=1)                                       # searchsorted(u, side='right')
        # spot-check that formula against numpy's own choice() on a few rows
        for i in range(0, N, N // 50):
            rs = np.random.RandomState(i); ui = np.random.RandomState(i).random_sample()
            assert rs.choice(np.arange(Q), p=sp[i]) == np.searchsorted(cdf[i], ui, side="right")
        pr = np.empty(Q, np.float32)
        for name, fn in (("sequential", lib.twvo_sample_categorical_sequential), ("current", lib.twvo_sample_categorical)):
            agree, maxrel = 0, 0.0
            for i in range(N):
                k = fn(logits[i].ctypes.data_as(fp), Q, temp, float(u[i]), pr.ctypes.data_as(fp))
                agree += int(k == want[i])
                if i % 16 == 0:
                    nz = sp[i] > 1e-30
                    maxrel = max(maxrel, float(np.max(np.abs(pr[nz] - sp[i][nz]) / sp[i][nz])))
            report[(temp, name)] = (agree, maxrel)
    print("categorical sampler vs numpy (agreeing draws of %d, max relative p error):" % N)
    for key in sorted(report):
        print("   T=%.1f %-10s %6d  %.2e" % (key[0], key[1], report[key][0], report[key][1]))
    for temp in (1.0, 0.8, 1.3):
        a_new, e_new = report[(temp, "current")]
        a_old, e_old = report[(temp, "sequential")]
        assert a_new >= a_old, (temp, a_new, a_old)
        assert a_new >= N - 3, (temp, a_new)          # u within float32 noise of a cdf boundary: a few per million
        assert e_new < 2e-5 and e_old < 2e-5          # both differ from numpy by the exp/log implementations, not by the order


def test_restatement_fixtures(oracle):
    """the committed restatement_* fixtures pin the oracle against drift (they are NOT reference goldens)"""
    f = np.load(os.path.join(GOLD, "restatement_codec_math.npz"))
    assert np.array_equal(oracle.mu_law_encode(f["audio"], 256), f["q"])
    assert first_mismatch(oracle.mu_law_decode(np.arange(256, dtype=np.int32), 256), f["dec"]) is None
    for name in ("tanh", "sigmoid", "exp"):
        assert first_mismatch(oracle.elementwise(name, f["x"]), f[name]) is None, name
    assert first_mismatch(oracle.elementwise("log", f["xp"]), f["log"]) is None
    g = np.load(os.path.join(GOLD, "restatement_wavenet_mol_small.npz"))
    dil = [int(v) for v in g["dilations"]]
    d, tensors, blob = make_case(oracle, dil, S=int(g["S"]), scale=float(g["scale"]), seed=int(g["weight_seed"]))
    U = oracle.upsample(d, blob, g["mel"])
    assert first_mismatch(U[:, :8], g["upsampled_head"]) is None
    T = g["uniforms"].shape[1]
    out = oracle.generate_mol(d, blob, oracle.State(d, 2), U[:, :T], g["gc_ids"], g["first_input"], g["uniforms"])
    assert first_mismatch(out, g["samples"]) is None


def test_restatement_fixtures_mulaw_and_tacotron(oracle):
    """drift pins for the one-hot mu-law generation path and the Tacotron restatement (again: NOT reference goldens)"""
    g = np.load(os.path.join(GOLD, "restatement_wavenet_mulaw_small.npz"))
    dil = [int(v) for v in g["dilations"]]
    d, tensors, blob = make_case(oracle, dil, scalar_input=False, S=int(g["S"]), Q=int(g["Q"]), scale=float(g["scale"]), seed=int(g["weight_seed"]))
    for temp, key in ((1.0, "samples_t10"), (0.8, "samples_t08")):
        out = oracle.generate_mulaw(d, blob, oracle.State(d, 2), g["upsampled"], g["gc_ids"], g["first_input"], g["uniforms"], temp)
        assert np.array_equal(out, g[key]), key
    assert not np.array_equal(g["samples_t10"], g["samples_t08"])                # the temperature does something
    t = np.load(os.path.join(GOLD, "restatement_tacotron_small.npz"))
    d = oracle.taco_dims(enc_bank=int(t["dims_enc_bank"]), post_bank=int(t["dims_post_bank"]), max_iters=int(t["dims_max_iters"]),
                         num_freq=int(t["dims_num_freq"]))
    blob = oracle.taco_blob(d, oracle.taco_random_tensors(d, seed=int(t["weight_seed"])))
    mel, lin, al = oracle.taco_infer(d, blob, t["tokens"], t["lengths"], t["speaker_ids"])
    assert first_mismatch(mel, t["mel"]) is None and first_mismatch(lin, t["linear"]) is None and first_mismatch(al, t["alignments"]) is None
    assert t["mel"].shape == (3, 30, 80) and t["linear"].shape == (3, 30, 129) and t["alignments"].shape == (3, 17, 6)


@pytest.mark.parametrize("n_speakers", [2, 1, -2, "simple"])
def test_torch_tacotron_reference_agrees_with_the_c_restatement(oracle, n_speakers):
    """oracle/tacotron.c gets a second opinion -- tests/torch_tacotron_ref.py, a float64 torch restatement of
    tacotron.py:36-235 + modules.py + rnn_wrappers.py written from the reference source and TensorFlow's published semantics, not from
    the C file.  Default dims (hparams.py:126-165), 25 decoder steps, ragged lengths; multi-speaker (deepvoice) and single speaker.
    1e-5 absolute on mel / linear (values of order 1), 1e-6 on the alignments: float32 chain vs float64."""
    import torch_tacotron_ref as R
    simple = n_speakers == "simple"                                      # two speakers, hparams.model_type 'simple' (tacotron.py:85-90)
    n_speakers = 2 if simple else n_speakers
    tables = n_speakers < 0                                              # -2: two speakers, speaker_embedding_size == 1 (tacotron.py:69-75 get_embed tables)
    n_speakers = abs(n_speakers)
    d = oracle.taco_dims(max_iters=25, n_speakers=n_speakers, spk_emb=1 if tables else 16, model_simple=simple)
    w = oracle.taco_random_tensors(d, seed=3)
    if simple:
        assert w["speaker_embedding"].shape == (2, 16) and "dense_1/kernel" not in w and w["dense/kernel"].shape == (2 * d.post_rnn, d.num_freq)
        assert w["decoder/attention_wrapper/gru_cell/gates/kernel"].shape == (128 + 16 + 256 + 256, 512)
        assert w["decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/kernel"].shape == (256 + 256 + 16, 256)
    if tables:
        assert "speaker_embedding" not in w and w["before_highway"].shape == (2, 128) and w["decoder_rnn_init_states2"].shape == (2, 256)
        assert w["dense/kernel"].shape == (2 * d.post_rnn, d.num_freq) and "dense_1/kernel" not in w
    blob = oracle.taco_blob(d, w)
    rng = np.random.RandomState(4)
    N, T = 3, 19
    lengths = np.array([19, 12, 7], np.int32)
    tok = rng.randint(2, 80, (N, T)).astype(np.int32)
    for n, ln in enumerate(lengths):
        tok[n, ln - 1] = 1
        tok[n, ln:] = 0
    spk = np.array([0, 1, 0], np.int32) if n_speakers > 1 else None
    mel, lin, al = oracle.taco_infer(d, blob, tok, lengths, spk)
    m2, l2, a2 = R.infer(w, d, tok, lengths, spk)
    assert mel.shape == m2.shape == (N, 125, 80) and lin.shape == l2.shape and al.shape == a2.shape == (N, T, 25)
    assert np.abs(mel - m2).max() <= 1e-5, np.abs(mel - m2).max()
    assert np.abs(lin - l2).max() <= 1e-5, np.abs(lin - l2).max()
    assert np.abs(al - a2).max() <= 1e-6, np.abs(al - a2).max()
    assert np.abs(m2).max() > 0.5 and a2[1, 12:].max() == 0.0           # a real signal; nothing attends past input_lengths
    if n_speakers == 1:                                                  # tacotron.py:97-104: no speaker tensors at all
        assert "speaker_embedding" not in w and w["dense/kernel"].shape == (2 * d.post_rnn, d.num_freq) and "dense_1/kernel" not in w


# ---------------------------------------------------------------- host side of the product
def test_blob_layout_agrees_with_oracle(oracle):
    """the product's canonical blob (weights.py + the C-ABI's count) and the oracle's independent one"""
    import twvk_amd
    from twvk_amd import weights as W, _lib
    for kw in (dict(), dict(scalar_input=False), dict(use_bias=False, G=0), dict(L=0, S=128)):
        d = oracle.make_dims([1, 2, 4, 8], **kw)
        tensors = oracle.random_tensors(d, seed=1)
        specs = W.tensor_specs(4, 32, 32, d.S, d.Q, 30, bool(d.scalar_input), 32, bool(d.use_bias), d.G, d.gc_card, d.L, (5, 5, 12))
        assert [n for n, _ in specs] == [n for n, _ in oracle.tensor_specs(d)]
        assert np.array_equal(W.flatten(specs, tensors), oracle.blob_from_tensors(d, tensors))
        dims = _lib.Dims()
        dims.n_layers = 4
        for i, v in enumerate([1, 2, 4, 8]):
            dims.dilations[i] = v
        dims.residual_channels = dims.dilation_channels = 32
        dims.skip_channels, dims.quantization_channels, dims.out_channels = d.S, d.Q, 30
        dims.scalar_input, dims.initial_filter_width, dims.use_biases = d.scalar_input, 32, d.use_bias
        dims.gc_channels, dims.gc_cardinality, dims.lc_channels = d.G, d.gc_card, d.L
        dims.n_upsample = 3 if d.L else 0
        for i, v in enumerate((5, 5, 12)):
            dims.upsample_factor[i] = v
        h = C.c_void_p()
        L = _lib.lib()
        _lib.check(L.twv_wavenet_create(C.byref(dims), C.byref(h)))      # host-only: no device is touched
        assert L.twv_wavenet_blob_floats(h) == oracle.blob_floats(d)
        assert L.twv_wavenet_receptive_field(h) == oracle.receptive_field(d)
        assert L.twv_wavenet_hop_size(h) == (300 if d.L else 1)
        L.twv_wavenet_destroy(h)


def test_c_abi_exports_match_header():
    import twvk_amd
    from twvk_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "twv_amd.h")).read()
    declared = set(re.findall(r"\b(twv_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = _lib.lib()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(_lib.EXPORTS) <= declared
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    assert "wn_generate_kernel" in nm or "_Z18wn_generate_kernel" in nm, "gfx950 kernels must be in the library"


def test_create_rejects_unsupported_dims():
    import twvk_amd
    from twvk_amd.wavenet import WaveNetModel
    from twvk_amd._lib import TwvError
    with pytest.raises(TwvError):
        WaveNetModel(1, [1, 2], 2, 16, 16, 512, scalar_input=True, out_channels=30, device="cpu")     # R, D must be 32
    with pytest.raises(TwvError):
        WaveNetModel(1, [1, 2], 2, 32, 32, 500, scalar_input=True, out_channels=30, device="cpu")     # S % 64
    with pytest.raises(ValueError):
        WaveNetModel(1, [1, 2], 3, 32, 32, 512, device="cpu")                                          # filter_width


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "tacotron-wavenet-vocoder-korean_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("CPU checker", ""), os.path.join(dirpath, f)


# ---------------------------------------------------------------- multi-rank layout (gloo, world_size 2)
def test_shard_ranges():
    import twvk_amd
    from twvk_amd.shard import shard_range
    for n in (0, 1, 7, 8, 9, 64):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(n, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import twvk_amd
from twvk_amd.shard import shard_range, max_over_ranks, gather_on_rank0
from oracle import oracle as O
from helpers import make_case, mol_uniforms
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
# 5 utterances over 2 ranks (ragged 3 + 2): every rank generates ITS utterances; no collective on the data path
dil = [1, 2, 4]
d, tensors, blob = make_case(O, dil, S=64, L=0, scale=0.2)
B, T = 5, 20
u = mol_uniforms(B, T, 10); seed = np.linspace(-0.5, 0.5, B).astype(np.float32); gc = (np.arange(B) % 2).astype(np.int32)
a, b = shard_range(B, 2, rank)
mine = O.generate_mol(d, blob, O.State(d, b - a), None, gc[a:b], seed[a:b], u[a:b])
wall = max_over_ranks(1.0 + rank)
parts = gather_on_rank0(mine)
if rank == 0:
    whole = O.generate_mol(d, blob, O.State(d, B), None, gc, seed, u)
    assert wall == 2.0
    assert np.array_equal(np.concatenate(parts, 0), whole)      # sharding does not change any utterance
    print("OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_sharding_gloo(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]


_TRAIN_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import twvk_amd
from twvk_amd import weights as W
from twvk_amd.train import allreduce_sum_
import torch_train_ref as R
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
# data-parallel training step (SURVEY.md 8e): every rank differentiates the mean loss of ITS crops, the flat gradient buffer is
# summed over ranks and scaled by 1/world == the gradient of the mean loss over the global batch
dil = [1, 2]
specs = W.tensor_specs(len(dil), S=64)
tensors = W.random_tensors(specs, seed=0, scale=0.1)           # identical on both ranks
cfg = dict(dilations=dil, initial_filter_width=32, use_biases=True, upsample_factor=(5, 5, 12))
rng = np.random.RandomState(5)
audio = (rng.rand(4, 300) - 0.5).astype(np.float32); lc = rng.randn(4, 1, 80).astype(np.float32); gc = np.array([0, 1, 1, 0], np.int32)
a, b = 2 * rank, 2 * rank + 2
_, g = R.loss_and_grads(tensors, cfg, audio[a:b], lc[a:b], gc[a:b])
flat = torch.from_numpy(W.flatten(specs, g))
world = allreduce_sum_(flat)
flat *= 1.0 / world
assert world == 2
_, gw = R.loss_and_grads(tensors, cfg, audio, lc, gc)
whole = W.flatten(specs, gw)
err = np.abs(flat.numpy() - whole).max() / np.abs(whole).max()
assert err < 1e-5, err                                       # fp32 round-off of two half-batch means vs one full-batch mean
print("OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_gradient_allreduce_gloo(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "train_worker.py"
    script.write_text(_TRAIN_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0] and "OK" in outs[1]


def test_training_host_helpers():
    import twvk_amd
    from twvk_amd.train import exponential_decay, crop_length, allreduce_sum_
    assert crop_length(8000, 300) == 7800 and crop_length(15000, 300) == 15000       # datafeeder_wavenet.py:41-47
    assert exponential_decay(1e-3, 0, 300000, 0.5) == 1e-3
    assert abs(exponential_decay(1e-3, 300000, 300000, 0.5) - 5e-4) < 1e-18              # model.py:320
    import torch
    t = torch.ones(3)
    assert allreduce_sum_(t) == 1 and t.tolist() == [1.0, 1.0, 1.0]                       # no process group: identity


def test_torch_training_reference_agrees_with_the_c_restatement(oracle):
    """the two checkers (torch fp32 autograd model for C4, bit-exact C restatement) compute the same train-mode network"""
    import torch
    import torch_train_ref as R
    dil = [1, 2, 4, 1, 2]
    d, tensors, blob = make_case(oracle, dil, S=64)
    rng = np.random.RandomState(1)
    B, Tm = 2, 2
    audio = (rng.rand(B, Tm * 300).astype(np.float32) - 0.5)
    lc = (rng.randn(B, Tm, 80) * 0.5).astype(np.float32)
    gc = np.array([0, 1], np.int32)
    U = oracle.upsample(d, blob, lc)
    raw = oracle.forward_full(d, blob, audio[:, :-1], U, gc)
    P = {k: torch.tensor(v) for k, v in tensors.items()}
    Ut = R.upsample(torch.tensor(lc), [P["wavenet/upsample%d/kernel" % i] for i in range(3)], (5, 5, 12))
    assert np.abs(Ut.numpy() - U).max() < 1e-6
    cfg = dict(dilations=dil, initial_filter_width=32, use_biases=True, upsample_factor=(5, 5, 12))
    y = R.network(P, cfg, torch.tensor(audio[:, None, :-1]), Ut, torch.tensor(gc)).numpy()
    assert y.shape == raw.shape and np.abs(y - raw).max() < 1e-5     # tolerance: fp32 sums in a different order


def test_torch_training_reference_matmul_form_equals_conv_form():
    """tests/torch_train_ref.py: `network_mm` (what the full-size configs[3] GPU test runs in float64 on the device) is the same
    graph as the conv1d form -- loss and every gradient agree to float64 round-off, scalar-input and one-hot model"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_train_ref as R
    import twvk_amd  # noqa: F401
    from twvk_amd import weights as W
    for scalar in (True, False):
        dil = [1, 2, 4, 8, 1, 2]
        tensors = W.random_tensors(W.tensor_specs(len(dil), S=64, scalar_input=scalar, Q=256), seed=0, scale=0.1)
        rng = np.random.RandomState(1)
        B, Tm = 2, 2
        audio = ((rng.rand(B, Tm * 300) - 0.5) * 1.6).astype(np.float32)
        lc = (rng.randn(B, Tm, 80) * 0.5).astype(np.float32)
        gc = np.array([1, 0], np.int32)
        cfg = dict(dilations=dil, initial_filter_width=32, use_biases=True, upsample_factor=(5, 5, 12), scalar_input=scalar, Q=256)
        q = None if scalar else rng.randint(256, size=(B, Tm * 300))
        l1, g1 = R.loss_and_grads(tensors, cfg, audio, lc, gc, dtype=torch.float64, quantized=q)
        l2, g2 = R.loss_and_grads(tensors, cfg, audio, lc, gc, dtype=torch.float64, quantized=q, matmul_form=True)
        assert abs(l1 - l2) <= 1e-12 * abs(l1)
        for k in g1:
            assert np.abs(g1[k] - g2[k]).max() <= 1e-12 * max(np.abs(g1[k]).max(), 1e-30), k


def test_attention_trim_rule():
    """synthesizer.py:232-256: stop at the decoder step where attention has sat on the last token (<= 5 steps) or moves past it"""
    import twvk_amd
    from twvk_amd.e2e import attention_trim_frames, shard_utterances
    al = np.zeros((4, 10), np.float32)                 # (T_in, T_dec): argmax path 0,0,1,1,2,3,3,3,3,3
    for j, i in enumerate([0, 0, 1, 1, 2, 3, 3, 3, 3, 3]):
        al[i, j] = 1.0
    # end_idx = min(len-1, max argmax) = 3, it occurs 5 times -> max_counter 5 -> breaks at jdx = 9 (5th hit) ... the loop's
    # `len > jdx + 1` guard ends it at the last step; r = 5 -> 5*9 + 3
    assert attention_trim_frames(al, 4, 5) == 5 * 9 + 3
    al2 = np.zeros((4, 10), np.float32)
    for j, i in enumerate([0, 1, 2, 2, 2, 3, 3, 3, 3, 3]):
        al2[i, j] = 1.0
    # sequence_length 3 -> end_idx = 2; first hit at jdx 2, leaves token 2 for a later one after jdx 4 -> break at jdx 4
    assert attention_trim_frames(al2, 3, 5) == 5 * 4 + 3
    assert [shard_utterances(8, 8, r) for r in range(8)] == [(r, r + 1) for r in range(8)]


def test_audio_restatement_against_scipy():
    """oracle/audio_np.py (librosa's stft/istft restated): perfect reconstruction away from the edges, scipy's periodic Hann,
    scipy's lfilter for the inverse pre-emphasis, frame count 1 + len//hop"""
    from oracle import audio_np as A
    from scipy import signal
    rng = np.random.RandomState(0)
    y = rng.randn(300 * 20)
    D = A.stft(y, 2048, 300, 1200)
    assert D.shape == (1025, 21)
    yr = A.istft(D, 300, 1200)
    assert len(yr) == 300 * 20 and np.abs(yr[2048:-2048] - y[2048:len(yr) - 2048]).max() < 1e-12
    assert np.abs(signal.get_window("hann", 1200, fftbins=True) - A.hann_padded(1200, 1200)).max() < 1e-15
    assert A.hann_padded(1200, 2048)[:424].max() == 0 and A.hann_padded(1200, 2048)[1624:].max() == 0
    x = rng.randn(500)
    assert np.abs(signal.lfilter([1], [1, -0.97], x) - A.inv_preemphasis(x, 0.97)).max() < 1e-12
    assert A.denormalize(np.array([-9.0, -4.0, 0.0, 4.0, 9.0])).tolist() == [-100.0, -100.0, -50.0, 0.0, 0.0]     # utils/audio.py:225-227
    assert abs(A.db_to_amp(20.0) - 10.0) < 1e-12


# ---------------------------------------------------------------- Tacotron restatement (oracle/tacotron.c)
def test_tacotron_oracle_invariants(oracle):
    d = oracle.taco_dims(max_iters=10, enc_bank=3, post_bank=2, num_freq=33)
    tensors = oracle.taco_random_tensors(d, seed=2)
    blob = oracle.taco_blob(d, tensors)
    rng = np.random.RandomState(0)
    N, T = 2, 17
    tok = rng.randint(2, 80, (N, T)).astype(np.int32)
    tok[0, -1] = 1; tok[1, 9] = 1; tok[1, 10:] = 0
    ln = np.array([17, 10], np.int32)
    mel, lin, al = oracle.taco_infer(d, blob, tok, ln, np.array([0, 1], np.int32))
    assert mel.shape == (N, 10 * 5, 80) and lin.shape == (N, 50, 33) and al.shape == (N, T, 10)    # tacotron.py:204,219,223
    assert np.isfinite(mel).all() and np.isfinite(lin).all()
    assert al.min() >= 0 and np.all(al[1, 10:] == 0)                   # masked past input_lengths
    mass = al.sum(axis=1)
    assert np.all(mass <= 1 + 1e-5) and np.all(np.diff(mass, axis=1) <= 1e-6)     # monotonic attention only loses mass
    centre = (al * np.arange(T)[None, :, None]).sum(1) / np.maximum(mass, 1e-9)
    assert np.all(np.diff(centre[0]) >= -1e-4)                         # ... and only moves forward
    # utterances are independent: utterance 1 alone gives the same rows
    mel1, _, _ = oracle.taco_infer(d, blob, tok[1:], ln[1:], np.array([1], np.int32))
    assert np.array_equal(mel1[0], mel[1])
    # token 0 embeds to zeros (tacotron.py:56): changing the embedding row 0 changes nothing
    t2 = dict(tensors); e = t2["embedding"].copy(); e[0] += 1.0; t2["embedding"] = e
    mel2, _, _ = oracle.taco_infer(d, oracle.taco_blob(d, t2), tok, ln, np.array([0, 1], np.int32))
    assert np.array_equal(mel2, mel)


def test_tacotron_blob_layout_agrees(oracle):
    import twvk_amd
    from twvk_amd import tacotron as TP, _lib
    hp = twvk_amd.default_hparams()
    d = oracle.taco_dims()
    specs = TP.tacotron_specs(hp, 2)
    assert specs == oracle.taco_tensor_specs(d)
    tensors = oracle.taco_random_tensors(d, seed=1)
    assert np.array_equal(TP.flatten(specs, tensors), oracle.taco_blob(d, tensors))
    m = TP.Tacotron(hp, num_speakers=2, device="cpu")                   # host-only: create touches no device
    assert _lib.lib().twv_tacotron_blob_floats(m._h) == oracle.taco_blob(d, tensors).size


def test_oracle_streams_on_threads_give_the_same_bits(oracle):
    """the checker's generate loop runs one stream per host thread when asked (bench.py's all-cores baseline, the full-size GPU
    parity tests); a stream's arithmetic does not depend on the thread count"""
    from helpers import make_case, mol_uniforms
    dil = [1, 2, 4, 8, 16, 32]
    d, tensors, blob = make_case(oracle, dil)
    B, T = 5, 40
    rng = np.random.RandomState(0)
    U = rng.uniform(-1, 1, (B, T, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed = rng.uniform(-1, 1, B).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    cores = oracle.set_threads(1)
    assert cores >= 1
    a = oracle.generate_mol(d, blob, oracle.State(d, B), U, gc, seed, u)
    oracle.set_threads(4)
    try:
        b = oracle.generate_mol(d, blob, oracle.State(d, B), U, gc, seed, u)
    finally:
        oracle.set_threads(1)
    assert np.array_equal(a, b)


def test_library_is_stamped_with_the_source_hash():
    """_lib.build() compiles the hash of csrc/ + include/ + flags into twv_version(): a stale binary cannot pass for the tree
    (own process: this one may have loaded the library before a rebuild)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import twvk_amd; twvk_amd._lib.build(); L = twvk_amd._lib.lib(); "
            "assert L.twv_version().decode().endswith('src:' + twvk_amd._lib.source_hash()), L.twv_version()")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside a launcher becomes two ranks (torch.distributed.run on 127.0.0.1); --dry-run drives that
    path without a GPU: gloo rendezvous, barrier-bracketed region, MAX over ranks, n_gpus = ranks that reported, ONE JSON line"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["dry_run"] is True and res["n_gpus"] == 2 and res["world"] == 2
    assert res["max_seconds"] >= 0.1            # the slower rank (0.05 s x 2) bounds the timed region
    # --gpus 1 needs no launcher
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0 and json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_bench_line_schema_at_eight_ranks():
    """the line the driver parses at N = 8, without a GPU -- `bench.py --gpus 8 --dry-run` spawns eight gloo ranks and
    prints ONE line built by the same helpers as the measured run (headline_fields / checked_fields / train_collective): BASELINE.json's
    metric, whole-job value, n_gpus = ranks that produced samples, weak scaling, checked_ranks, and a training collective that names RCCL"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert base["metric"].startswith(res["metric"])                       # "...; mel frames/sec" is the secondary `tacotron` object's metric
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "checked_ranks", "train"):
        assert k in res, k
    assert res["n_gpus"] == 8 and res["world"] == 8 and res["checked_ranks"] == 8 and res["steps"] == 3 and res["warmup"] == 1
    assert res["scaling"] == "weak" and res["higher_is_better"] is True and res["vs_baseline"] is None and res["dtype"] == "f32" and res["data"] == "synthetic"
    assert set(res["config"]) >= {"workload", "batch_per_gpu", "samples_per_utterance", "sharding", "kernel"} and "model" not in res["config"]
    assert "no collective" in res["config"]["sharding"]
    assert set(res["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    # value = the samples ALL ranks produced / the slowest rank's time: 8 ranks x 8 utterances x 192 000 samples x 3 steps
    assert abs(res["value"] * res["max_seconds"] - 8 * 8 * 192000 * 3) < 1.0 and res["max_seconds"] >= 0.4
    assert "RCCL" in res["train"]["collective"] and "all-reduce" in res["train"]["collective"] and res["train"]["n_gpus"] == 8
    import bench
    assert bench.train_collective(1, 10) == "none (1 GPU)"


def test_crop_rule_of_the_wavenet_feeder():
    """datafeeder_wavenet.py:153-156: a random FRAME offset; audio and mel cut in step (hop multiples); global np.random"""
    import twvk_amd
    from twvk_amd.train_vocoder import crop_example, ensure_divisible
    hop, frames, max_frames = 300, 40, 26
    assert ensure_divisible(8000, hop, True) == 7800 and ensure_divisible(7800, hop, True) == 7800     # datafeeder_wavenet.py:41-47
    audio = np.arange(frames * hop, dtype=np.float32)
    mel = np.repeat(np.arange(frames, dtype=np.float32)[:, None], 80, axis=1)
    np.random.seed(7)
    want_s = np.random.randint(0, frames - max_frames + 1)
    np.random.seed(7)
    a, m = crop_example(audio.reshape(-1, 1), mel, max_frames, hop)
    assert a.shape == (max_frames * hop,) and m.shape == (max_frames, 80)
    assert a[0] == want_s * hop and m[0, 0] == want_s                      # same offset for both, in units of frames
    seen = set()
    for _ in range(200):
        a, m = crop_example(audio, mel, max_frames, hop)
        assert a[0] == m[0, 0] * hop and a[-1] == (m[-1, 0] + 1) * hop - 1
        seen.add(int(m[0, 0]))
    assert seen == set(range(frames - max_frames + 1))                      # every offset incl. both ends (randint's high is exclusive)
    with pytest.raises(AssertionError):
        crop_example(audio[:-1], mel, max_frames, hop)                      # assert_ready_for_upsampling


def test_wavenet_feeder_batches(tmp_path):
    """the feeder's batch rule (datafeeder_wavenet.py:104-118) on two speaker directories of npz examples"""
    import twvk_amd
    from twvk_amd.train_vocoder import DataFeederWavenet
    hp = twvk_amd.default_hparams()
    hp.sample_size = 1500
    dirs = []
    rng = np.random.RandomState(0)
    for spk in range(2):
        d = tmp_path / ("spk%d" % spk); d.mkdir(); dirs.append(str(d))
        for i in range(3):
            frames = 9 + i
            np.savez(str(d / ("ex%d.npz" % i)), audio=rng.uniform(-1, 1, frames * 300).astype(np.float32), mel=rng.randn(frames, 80).astype(np.float32))
    f = DataFeederWavenet(dirs, batch_size=4, receptive_field=1000, gc_enable=True, hp=hp)
    assert f.sample_size == 1500 and f.max_frames == 5
    ids = []
    for _ in range(8):
        a, m, g = f.next_batch()
        assert a.shape == (4, 1500) and m.shape == (4, 5, 80) and g.shape == (4,) and g.dtype == np.int32
        ids += g.tolist()
    assert set(ids) == {0, 1}
    # data-parallel ranks read disjoint examples of every speaker directory (the all-reduced gradient then covers world x batch crops)
    r0 = DataFeederWavenet(dirs, batch_size=4, receptive_field=1000, gc_enable=True, hp=hp, rank=0, world=2)
    r1 = DataFeederWavenet(dirs, batch_size=4, receptive_field=1000, gc_enable=True, hp=hp, rank=1, world=2)
    for d in dirs:
        assert r0.path_dict[d] and r1.path_dict[d] and not set(r0.path_dict[d]) & set(r1.path_dict[d])
        assert sorted(r0.path_dict[d] + r1.path_dict[d]) == sorted(f.path_dict[d])
    a0, _, _ = r0.next_batch(); a1, _, _ = r1.next_batch()
    assert a0.shape == a1.shape == (4, 1500) and not np.array_equal(a0, a1)


def test_train_vocoder_directory_rules(tmp_path):
    """utils/__init__.py:100-142 validate_directories as train_vocoder.py uses it"""
    import argparse
    import twvk_amd
    from twvk_amd.train_vocoder import validate_directories
    hp = twvk_amd.default_hparams()
    ns = argparse.Namespace(logdir=str(tmp_path / "a"), logdir_root=str(tmp_path), restore_from=None)
    with pytest.raises(ValueError):
        validate_directories(ns, hp)
    ns = argparse.Namespace(logdir=str(tmp_path / "a"), logdir_root=None, restore_from=str(tmp_path / "b"))
    with pytest.raises(ValueError):
        validate_directories(ns, hp)
    ns = argparse.Namespace(logdir=None, logdir_root=str(tmp_path / "root"), restore_from=str(tmp_path / "b"))
    d = validate_directories(ns, hp)
    assert d["restore_from"] == str(tmp_path / "b") and d["logdir"].startswith(str(tmp_path / "root")) and os.path.exists(os.path.join(d["logdir"], "params.json"))
    ns = argparse.Namespace(logdir=str(tmp_path / "c"), logdir_root=None, restore_from=None)
    d = validate_directories(ns, hp)
    assert d["logdir"] == d["restore_from"] == str(tmp_path / "c")


def test_bench_reports_counter_traffic_only_for_the_measured_sources(monkeypatch):
    """roofline.traffic comes from profiles/traffic.json, keyed by the hash of the generation kernels' sources: the committed entry
    belongs to THIS tree (a stale file would silently turn the field into null), any other code gets null"""
    import importlib
    import twvk_amd
    bench = importlib.import_module("bench")
    h = twvk_amd._lib.generation_hash()
    assert h != twvk_amd._lib.source_hash() and len(h) == 16
    got = bench.traffic_per_step("wn_xcd_generate_kernel", "B8_NL30")
    assert got is not None and 3000.0 < got < 10000.0, "profiles/traffic.json is not keyed to the generation sources of this tree (re-run scripts/profile_generation.sh)"
    assert bench.traffic_per_step("wn_xcd_generate_kernel", "B8_NL50") is None
    monkeypatch.setattr(twvk_amd._lib, "generation_hash", lambda: "0" * 16)
    assert bench.traffic_per_step("wn_xcd_generate_kernel", "B8_NL30") is None
    # the secondary rows the same way: a figure only for the code the counters ran on
    for fn, name in ((bench.tacotron_traffic, "tacotron_hash"), (bench.train_traffic, "train_hash")):
        got = fn()
        assert got is not None and 1e8 < got < 3e11, "profiles/traffic.json has no %s entry for this tree's sources (re-run scripts/r06_profile_all.sh)" % name
        monkeypatch.setattr(twvk_amd._lib, name, lambda: "0" * 16)
        assert fn() is None


def test_variant_builds_never_pass_for_the_plain_library(monkeypatch):
    """TWV_EXTRA_HIPCC_FLAGS (tuning builds) goes into the library's build stamp: a plain import after a variant build must rebuild"""
    from twvk_amd import _lib
    monkeypatch.delenv("TWV_EXTRA_HIPCC_FLAGS", raising=False)
    extra, plain = _lib.build_stamp()
    assert extra == [] and plain == _lib.source_hash()
    monkeypatch.setenv("TWV_EXTRA_HIPCC_FLAGS", "-DTWV_TRPROF -DX=1")
    extra, variant = _lib.build_stamp()
    assert extra == ["-DTWV_TRPROF", "-DX=1"] and variant != plain and variant.startswith(plain + "+")
