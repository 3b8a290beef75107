"""-m gpu: Tacotron text -> mel inference, HIP path (through the C-ABI) vs the CPU oracle, bit for bit."""
import copy

import os

import numpy as np
import pytest

from helpers import first_mismatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _hp(**kw):
    import twvk_amd
    hp = twvk_amd.default_hparams()
    for k, v in kw.items():
        setattr(hp, k, v)
    return hp


def _case(oracle, hp, N, T, lengths, seed=0, num_speakers=2):
    from twvk_amd.tacotron import Tacotron
    d = oracle.taco_dims(enc_bank=hp.enc_bank_size, post_bank=hp.post_bank_size, enc_hw_depth=hp.enc_highway_depth,
                         post_hw_depth=hp.post_highway_depth, dec_layers=hp.dec_layer_num, max_iters=hp.max_iters, num_freq=hp.num_freq,
                         r=hp.reduction_factor, n_speakers=num_speakers)
    tensors = oracle.taco_random_tensors(d, seed=seed)
    blob = oracle.taco_blob(d, tensors)
    rng = np.random.RandomState(seed + 1)
    tok = rng.randint(2, 80, (N, T)).astype(np.int32)
    for n, ln in enumerate(lengths):
        tok[n, ln - 1] = 1                      # EOS
        tok[n, ln:] = 0                         # pad
    spk = (np.arange(N) % 2).astype(np.int32) if num_speakers > 1 else None
    m = Tacotron(hp, num_speakers=num_speakers)
    assert [n for n, _ in m.specs] == [n for n, _ in oracle.taco_tensor_specs(d)]
    m.load_weights(tensors)
    return d, blob, tok, np.asarray(lengths, np.int32), spk, m


def test_tacotron_small(torch_cuda, oracle):
    hp = _hp(max_iters=6, enc_bank_size=4, post_bank_size=3, num_freq=129)
    d, blob, tok, ln, spk, m = _case(oracle, hp, 3, 19, [19, 12, 7])
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    mel, lin, al = m.infer(tok, ln, spk)
    assert mel.shape == (3, 30, 80) and lin.shape == (3, 30, 129) and al.shape == (3, 19, 6)
    assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
    assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
    assert first_mismatch(lin.cpu().numpy(), lin_o) is None, ("linear", first_mismatch(lin.cpu().numpy(), lin_o))
    assert np.all(al.cpu().numpy()[1, 12:] == 0)                       # nothing attends past input_lengths


@pytest.mark.parametrize("dims", ["small", "default"])
def test_tacotron_single_speaker(torch_cuda, oracle, dims):
    """tacotron.py:97-104 (num_speakers == 1, the reference CLI's default, synthesizer.py:375): no speaker embedding, no before_highway,
    zero initial states of the encoder biGRU, the attention cell and the decoder GRUs; the linear layer is the graph's first
    tf.layers.dense ("dense").  Bit for bit against the checker, ragged lengths."""
    hp = _hp(max_iters=6, enc_bank_size=4, post_bank_size=3, num_freq=129) if dims == "small" else _hp(max_iters=25)
    N, T, lengths = (3, 19, [19, 12, 7]) if dims == "small" else (5, 40, [40, 33, 21, 8, 1])
    d, blob, tok, ln, spk, m = _case(oracle, hp, N, T, lengths, seed=31, num_speakers=1)
    names = [n for n, _ in m.specs]
    assert "speaker_embedding" not in names and "dense/kernel" in names and "dense_1/kernel" not in names
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, None)
    mel, lin, al = m.infer(tok, ln, None)
    assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
    assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
    assert first_mismatch(lin.cpu().numpy(), lin_o) is None, ("linear", first_mismatch(lin.cpu().numpy(), lin_o))
    # a speaker id passed to a single-speaker model is ignored (synthesizer.py:150-151 feeds none)
    mel2, _, _ = m.infer(tok, ln, np.zeros(N, np.int32), want_linear=False, want_alignments=False)
    assert first_mismatch(mel2.cpu().numpy(), mel_o) is None


def test_committed_restatement_fixture_tacotron(torch_cuda, oracle):
    """HIP Tacotron against the committed fixture's mel / linear / alignments (the oracle is only asked for the seeded weights)"""
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "restatement_tacotron_small.npz"))
    from twvk_amd.tacotron import Tacotron
    hp = _hp(max_iters=int(t["dims_max_iters"]), enc_bank_size=int(t["dims_enc_bank"]), post_bank_size=int(t["dims_post_bank"]),
             num_freq=int(t["dims_num_freq"]))
    d = oracle.taco_dims(enc_bank=hp.enc_bank_size, post_bank=hp.post_bank_size, max_iters=hp.max_iters, num_freq=hp.num_freq)
    m = Tacotron(hp, num_speakers=2)
    m.load_weights(oracle.taco_random_tensors(d, seed=int(t["weight_seed"])))
    mel, lin, al = m.infer(t["tokens"], t["lengths"], t["speaker_ids"])
    assert first_mismatch(mel.cpu().numpy(), t["mel"]) is None
    assert first_mismatch(lin.cpu().numpy(), t["linear"]) is None
    assert first_mismatch(al.cpu().numpy(), t["alignments"]) is None


def test_tacotron_default_dims(torch_cuda, oracle):
    """default hparams (hparams.py:126-165) with a shortened decode: N=2, T_in=40, 25 decoder steps -> 125 mel frames"""
    hp = _hp(max_iters=25)
    d, blob, tok, ln, spk, m = _case(oracle, hp, 2, 40, [40, 31], seed=3)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    mel, lin, al = m.infer(tok, ln, spk)
    assert mel.shape == (2, 125, 80) and lin.shape == (2, 125, 1025) and al.shape == (2, 40, 25)   # tacotron.py:204,219,223
    assert first_mismatch(mel.cpu().numpy(), mel_o) is None
    assert first_mismatch(al.cpu().numpy(), al_o) is None
    assert first_mismatch(lin.cpu().numpy(), lin_o) is None
    # north_star tolerance for float mel frames is 1e-4; bit-exact is stricter
    assert np.abs(mel.cpu().numpy() - mel_o).max() <= 1e-4


def test_synthesizer_surface(torch_cuda, oracle):
    import twvk_amd
    from twvk_amd.tacotron import Synthesizer
    hp = _hp(max_iters=4, num_freq=65)
    d = oracle.taco_dims(max_iters=4, num_freq=65)
    tensors = oracle.taco_random_tensors(d, seed=5)
    syn = Synthesizer()
    syn.load(tensors, num_speakers=2, hparams=hp)
    out = syn.infer([[5, 9, 33, 12, 1], [7, 7, 1]], speaker_ids=[1, 0])
    assert out["input_lengths"] == [5, 3]                               # synthesizer.py:126 argmax(seq == 1) + 1
    tok = np.array([[5, 9, 33, 12, 1], [7, 7, 1, 0, 0]], np.int32)
    mel_o, _, _ = oracle.taco_infer(d, oracle.taco_blob(d, tensors), tok, np.array([5, 3], np.int32), np.array([1, 0], np.int32))
    assert first_mismatch(out["mel"].cpu().numpy(), mel_o) is None


@pytest.mark.parametrize("groups", [-1, 1, 2, 4, 8, 16, 32])
def test_decoder_launch_geometry_does_not_change_results(torch_cuda, oracle, groups):
    """decoder split over G workgroups per utterance (exchange through polled granules) == single-workgroup kernel == the XCD-local
    register-resident kernel (32) == oracle"""
    hp = _hp(max_iters=7, enc_bank_size=3, post_bank_size=2, num_freq=65)
    d, blob, tok, ln, spk, m = _case(oracle, hp, 3, 37, [37, 20, 5], seed=11)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    m.set_option("decoder_groups", groups)
    mel, lin, al = m.infer(tok, ln, spk)
    assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
    assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
    assert first_mismatch(lin.cpu().numpy(), lin_o) is None
    mel2, _, _ = m.infer(tok, ln, spk)                                  # a second pass reuses the exchange buffers
    assert first_mismatch(mel2.cpu().numpy(), mel_o) is None


_PLACEMENT_ORACLE = {}


@pytest.mark.parametrize("local,split_all", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("batch", [3, 32])
def test_decoder_placement_does_not_change_results(torch_cuda, oracle, local, split_all, batch):
    """the split decoder with an utterance's workgroups on one XCD (exchanges through its L2; the default) or spread over the XCDs,
    with the prenet and the query layer split over the workgroups or computed whole in each: the same bits as the oracle.
    batch 3: three XCDs host one utterance each; batch 32 with ragged lengths: all 256 workgroups, four utterances per XCD"""
    hp = _hp(max_iters=9) if batch == 32 else _hp(max_iters=7, enc_bank_size=3, post_bank_size=2, num_freq=65)
    T = 61
    rng = np.random.RandomState(5)
    lens = [T] + [int(x) for x in rng.randint(3, T + 1, batch - 1)]
    d, blob, tok, ln, spk, m = _case(oracle, hp, batch, T, lens, seed=17)
    if batch not in _PLACEMENT_ORACLE:                                  # one CPU evaluation per batch, shared by the four placements
        _PLACEMENT_ORACLE[batch] = oracle.taco_infer(d, blob, tok, ln, spk)
    mel_o, lin_o, al_o = _PLACEMENT_ORACLE[batch]
    m.set_option("decoder_groups", 8)
    m.set_option("decoder_local", local)
    m.set_option("decoder_split_all", split_all)
    for _ in range(2):                                                  # the second pass reuses the exchange buffers and tickets
        mel, lin, al = m.infer(tok, ln, spk)
        assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
        assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
        assert first_mismatch(lin.cpu().numpy(), lin_o) is None


def test_tacotron_minimal_and_long_inputs(torch_cuda, oracle):
    """edge cases: a single EOS token; one utterance; an input longer than 256 tokens (more than one key row per thread); more than 512
    (the attention recurrence's 64-step blocks take a second round of the eight waves)"""
    hp = _hp(max_iters=3, enc_bank_size=2, post_bank_size=2, num_freq=33)
    for N, T, lengths in ((1, 1, [1]), (2, 2, [2, 1]), (1, 300, [300]), (2, 600, [600, 531])):
        d, blob, tok, ln, spk, m = _case(oracle, hp, N, T, lengths, seed=7)
        mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
        mel, lin, al = m.infer(tok, ln, spk)
        assert first_mismatch(mel.cpu().numpy(), mel_o) is None, (N, T)
        assert first_mismatch(al.cpu().numpy(), al_o) is None, (N, T)
        assert first_mismatch(lin.cpu().numpy(), lin_o) is None, (N, T)


def test_tacotron_gemm_kernels_agree(torch_cuda, oracle):
    """the VALU GEMM (option gemm_valu) and the MFMA GEMM give the same bits (both equal the oracle)"""
    hp = _hp(max_iters=4, enc_bank_size=5, post_bank_size=3, num_freq=65)
    d, blob, tok, ln, spk, m = _case(oracle, hp, 2, 33, [33, 17], seed=13)
    mel_a, lin_a, _ = m.infer(tok, ln, spk)
    m.set_option("gemm_valu", 1)
    try:
        mel_b, lin_b, _ = m.infer(tok, ln, spk)
    finally:
        m.set_option("gemm_valu", 0)
    assert first_mismatch(mel_a.cpu().numpy(), mel_b.cpu().numpy()) is None
    assert first_mismatch(lin_a.cpu().numpy(), lin_b.cpu().numpy()) is None
    mel_o, lin_o, _ = oracle.taco_infer(d, blob, tok, ln, spk)
    assert first_mismatch(mel_a.cpu().numpy(), mel_o) is None and first_mismatch(lin_a.cpu().numpy(), lin_o) is None
    # round 4's launch structure (conv banks / GRU input halves / speaker layers as grouped grids, the highway layer fused into one
    # MFMA launch with interleaved H | T tiles) against one launch per GEMM + the separate highway kernel: the same bits
    m.set_option("gemm_group", 0)
    try:
        mel_c, lin_c, al_c = m.infer(tok, ln, spk)
    finally:
        m.set_option("gemm_group", 1)
    assert first_mismatch(mel_a.cpu().numpy(), mel_c.cpu().numpy()) is None
    assert first_mismatch(lin_a.cpu().numpy(), lin_c.cpu().numpy()) is None
    # the highway stack as one launch per layer (the form before round 6) gives the same bits as the fused stack
    m.set_option("highway_stack", 0)
    try:
        mel_d, lin_d, al_d = m.infer(tok, ln, spk)
    finally:
        m.set_option("highway_stack", 1)
    assert first_mismatch(mel_a.cpu().numpy(), mel_d.cpu().numpy()) is None
    assert first_mismatch(lin_a.cpu().numpy(), lin_d.cpu().numpy()) is None


@pytest.mark.parametrize("steps", [25, 200])
def test_tacotron_at_bench_geometry(torch_cuda, oracle, steps):
    """BASELINE configs[2] as bench.py launches it: default dims, B = 32 utterances of 101 tokens, decoder split over 8 workgroups
    per utterance (all 256 workgroups resident), 25 decoder steps and the full 200 (= 1000 mel frames); bit for bit"""
    hp = _hp(max_iters=steps)
    N, T = 32, 101
    d, blob, tok, ln, spk, m = _case(oracle, hp, N, T, [T] * N, seed=21)
    m.set_option("decoder_groups", 8)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    mel, lin, al = m.infer(tok, ln, spk)
    assert mel.shape == (N, steps * 5, 80)
    assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
    assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
    assert first_mismatch(lin.cpu().numpy(), lin_o) is None, ("linear", first_mismatch(lin.cpu().numpy(), lin_o))


def test_synthesize_writes_the_reference_outputs(torch_cuda, oracle, tmp_path):
    """Synthesizer.synthesize with the reference's signature (synthesizer.py:72-83): per utterance the attention-trimmed wav
    (Griffin-Lim on the GPU) and the mel .npy that generate.py --mel reads; without base_path the wav bytes come back"""
    import io
    from scipy.io import wavfile
    from twvk_amd.synthesizer import Synthesizer, plot_graph_and_save_audio, main as synth_main
    from twvk_amd.audio import inv_linear_spectrogram
    from twvk_amd.e2e import attention_trim_frames
    from twvk_amd.ops import wav_to_int16
    hp = _hp(max_iters=6, griffin_lim_iters=3)
    d = oracle.taco_dims(max_iters=6, num_freq=hp.num_freq)
    tensors = oracle.taco_random_tensors(d, seed=5)
    syn = Synthesizer()
    syn.load(tensors, num_speakers=2, hparams=hp)
    toks = [[5, 9, 33, 12, 1], [7, 7, 1]]
    res = syn.synthesize(tokens=toks, base_path=str(tmp_path), speaker_ids=[1, 0], attention_trim=True, seed=11)
    assert res == [True, True]
    wavs = sorted(str(p) for p in tmp_path.glob("*.wav")); mels = sorted(str(p) for p in tmp_path.glob("*.npy"))
    assert len(wavs) == 2 and [w.replace(".wav", ".npy") for w in wavs] == mels
    out = syn.infer(toks, speaker_ids=[1, 0])
    for i in range(2):
        al = out["alignments"][i].cpu().numpy()
        n = attention_trim_frames(al, len(out["sequences"][i]), hp.reduction_factor)       # synthesizer.py:232-256
        mel = np.load(mels[i])
        assert first_mismatch(mel, out["mel"][i, :n].cpu().numpy()) is None               # the vocoder's --mel input
        want = inv_linear_spectrogram(out["linear"][i:i + 1, :n], hp, seed=11)
        sr, data = wavfile.read(wavs[i])
        assert sr == hp.sample_rate and np.array_equal(data, wav_to_int16(want).cpu().numpy().reshape(-1))
    blob = syn.synthesize(tokens=toks[:1], speaker_ids=[1], seed=11)[0]                    # no path: wav bytes (synthesizer.py:283-287)
    sr, data = wavfile.read(io.BytesIO(blob))
    assert sr == hp.sample_rate and np.array_equal(data, wavfile.read(wavs[0])[1])
    with pytest.raises(ValueError):
        syn.synthesize(texts="text needs the frontend")
    with pytest.raises(ValueError):
        syn.synthesize(tokens=toks, manual_attention_mode=1)


def test_synthesizer_cli(torch_cuda, oracle, tmp_path):
    """synthesizer.py:371-388 flags; the checkpoint is a TF-V2 bundle directory (most recent step, or --checkpoint_step)"""
    import twvk_amd
    from twvk_amd import checkpoint as ckpt
    from twvk_amd.hparams import save_hparams
    from twvk_amd.synthesizer import main as synth_main, get_most_recent_checkpoint
    hp = _hp(max_iters=4, griffin_lim_iters=2)
    d = oracle.taco_dims(max_iters=4, num_freq=hp.num_freq)
    tensors = oracle.taco_random_tensors(d, seed=5)
    logdir = tmp_path / "logdir"; logdir.mkdir()
    save_hparams(str(logdir), hp)
    for step in (1000, 3000):
        ckpt.write_bundle(str(logdir / ("model.ckpt-%d" % step)), ckpt.tacotron_variables(tensors))
    assert get_most_recent_checkpoint(str(logdir)).endswith("model.ckpt-3000")
    assert get_most_recent_checkpoint(str(logdir), 1000).endswith("model.ckpt-1000")
    out = tmp_path / "samples"
    assert synth_main(["--load_path", str(logdir), "--sample_path", str(out), "--tokens", "5,9,33,12,1", "--num_speakers", "2",
                       "--speaker_id", "1", "--seed", "3"]) is True
    assert len(list(out.glob("*.wav"))) == 1 and len(list(out.glob("*.npy"))) == 1
    # the CLI's own default (--num_speakers 1, synthesizer.py:375): a single-speaker checkpoint
    d1 = oracle.taco_dims(max_iters=4, num_freq=hp.num_freq, n_speakers=1)
    logdir1 = tmp_path / "logdir1"; logdir1.mkdir()
    save_hparams(str(logdir1), hp)
    ckpt.write_bundle(str(logdir1 / "model.ckpt-500"), ckpt.tacotron_variables(oracle.taco_random_tensors(d1, seed=6)))
    out1 = tmp_path / "samples1"
    assert synth_main(["--load_path", str(logdir1), "--sample_path", str(out1), "--tokens", "5,9,33,12,1", "--seed", "3"]) is True
    assert len(list(out1.glob("*.wav"))) == 1


def test_speaker_embedding_size_one_uses_embedding_tables(torch_cuda, oracle):
    """tacotron.py:69-75: with hparams.speaker_embedding_size == 1 the deepvoice model has no speaker embedding and no deep_dense layers --
    before_highway, the encoder / attention / decoder initial states are embedding tables of their own (modules.py:10-12 get_embed),
    looked up by speaker id, and the linear-spectrogram layer is the graph's first tf.layers.dense.  Bit for bit against the checker,
    on the split and on the resident decoder."""
    hp = _hp(max_iters=6, enc_bank_size=4, post_bank_size=3, num_freq=129, speaker_embedding_size=1)
    from twvk_amd.tacotron import Tacotron
    N, T, lengths = 4, 23, [23, 15, 9, 2]
    d = oracle.taco_dims(enc_bank=4, post_bank=3, max_iters=6, num_freq=129, n_speakers=3, spk_emb=1)
    tensors = oracle.taco_random_tensors(d, seed=13)
    blob = oracle.taco_blob(d, tensors)
    rng = np.random.RandomState(14)
    tok = rng.randint(2, 80, (N, T)).astype(np.int32)
    for n, ln in enumerate(lengths):
        tok[n, ln - 1] = 1
        tok[n, ln:] = 0
    ln_ = np.asarray(lengths, np.int32)
    spk = np.array([2, 0, 1, 2], np.int32)
    m = Tacotron(hp, num_speakers=3)
    names = [n for n, _ in m.specs]
    assert names == [n for n, _ in oracle.taco_tensor_specs(d)]
    assert "speaker_embedding" not in names and "before_highway" in names and "decoder_rnn_init_states2" in names and "dense/kernel" in names and "dense_1/kernel" not in names
    m.load_weights(tensors)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln_, spk)
    for groups in (0, 8):
        m.set_option("decoder_groups", groups)
        mel, lin, al = m.infer(tok, ln_, spk)
        assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
        assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
        assert first_mismatch(lin.cpu().numpy(), lin_o) is None
    # a different speaker gives a different utterance (the tables are really read)
    mel2, _, _ = m.infer(tok, ln_, np.array([0, 0, 1, 2], np.int32), want_linear=False, want_alignments=False)
    assert first_mismatch(mel2.cpu().numpy()[0], mel_o[0]) is not None and first_mismatch(mel2.cpu().numpy()[1:], mel_o[1:]) is None


@pytest.mark.parametrize("N,groups", [(3, 0), (5, 4), (32, 8)])
def test_model_type_simple_concatenates_the_speaker_embedding_inside_the_decoder(torch_cuda, oracle, N, groups):
    """tacotron.py:85-90 (hparams.model_type 'simple'): no speaker-dependent initial states; the speaker embedding is concatenated to the
    decoder prenet's output (rnn_wrappers.py:425-432: attention-cell input [prenet | embed | context]) and to [output, attention] in
    front of the first projection (:455-463).  Bit for bit against the checker (which its float64 torch second opinion follows to 1e-5,
    tests/test_cpu.py), ragged lengths, several split-decoder geometries incl. the bench batch."""
    hp = _hp(max_iters=5, enc_bank_size=3, post_bank_size=2, num_freq=65, model_type="simple")
    from twvk_amd.tacotron import Tacotron
    T = 41
    rng = np.random.RandomState(50 + N)
    lengths = [T] + [int(x) for x in rng.randint(2, T + 1, N - 1)]
    d = oracle.taco_dims(enc_bank=3, post_bank=2, max_iters=5, num_freq=65, n_speakers=4, model_simple=True)
    tensors = oracle.taco_random_tensors(d, seed=51)
    blob = oracle.taco_blob(d, tensors)
    tok = rng.randint(2, 80, (N, T)).astype(np.int32)
    for n, ln in enumerate(lengths):
        tok[n, ln - 1] = 1
        tok[n, ln:] = 0
    ln_ = np.asarray(lengths, np.int32)
    spk = (np.arange(N) % 4).astype(np.int32)
    m = Tacotron(hp, num_speakers=4)
    assert [n for n, _ in m.specs] == [n for n, _ in oracle.taco_tensor_specs(d)]
    assert dict(m.specs)["decoder/attention_wrapper/gru_cell/gates/kernel"] == (128 + 16 + 256 + 256, 512)
    m.load_weights(tensors)
    m.set_option("decoder_groups", groups)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln_, spk)
    for _ in range(2):
        mel, lin, al = m.infer(tok, ln_, spk)
        assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
        assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
        assert first_mismatch(lin.cpu().numpy(), lin_o) is None
    if N == 3:
        from twvk_amd._lib import TwvError
        m.set_option("decoder_groups", 32)
        with pytest.raises(TwvError, match="split decoder"):
            m.infer(tok, ln_, spk)


@pytest.mark.parametrize("groups", [0, 8])
def test_an_utterance_does_not_depend_on_its_place_in_the_batch(torch_cuda, groups):
    """Size-independent property at BASELINE configs[2]'s full size (B = 32, 101 tokens, 200 decoder steps; no checker involved): an
    utterance's mel, linear and alignments are the same bits wherever it sits in the batch and whoever sits next to it.  In the
    XCD-resident decoder (groups = 0) the place decides the XCD and the lane group of the matrix-core tasks (utterance n -> XCD n / 4,
    slot n mod 4), in the split decoder (8) the XCD its workgroups share; the GEMM rows move between workgroups either way."""
    import twvk_amd
    from twvk_amd.tacotron import Tacotron
    hp = twvk_amd.default_hparams()
    m = Tacotron(hp, num_speakers=2)
    rng = np.random.RandomState(3)
    tt = {}
    for n_, shp in m.specs:
        if n_.endswith("batch_normalization"):
            c_ = shp[1]; tt[n_] = np.stack([np.ones(c_), np.zeros(c_), np.zeros(c_), np.ones(c_)]).astype(np.float32)
        elif n_.endswith("attention_g"): tt[n_] = np.array([np.sqrt(1.0 / hp.attention_size)], np.float32)
        elif n_.endswith("attention_score_bias"): tt[n_] = np.zeros(1, np.float32)
        else:
            fan = int(np.prod(shp[:-1])) if len(shp) > 1 else 1
            tt[n_] = (rng.randn(*shp) * (0.05 if len(shp) == 1 else min(0.5, 1.2 / np.sqrt(fan)))).astype(np.float32)
    m.load_weights(tt)
    m.set_option("decoder_groups", groups)
    N, T = 32, 101
    lengths = np.array([T - (i * 7) % 60 for i in range(N)], np.int32)
    tok = rng.randint(2, 80, (N, T)).astype(np.int32)
    for n in range(N):
        tok[n, lengths[n] - 1] = 1; tok[n, lengths[n]:] = 0
    spk = (np.arange(N) % 2).astype(np.int32)
    mel, lin, al = [x.cpu().numpy() for x in m.infer(tok, lengths, spk)]
    assert np.isfinite(mel).all() and mel.shape == (N, 1000, 80)
    perm = rng.permutation(N)
    mel_p, lin_p, al_p = [x.cpu().numpy() for x in m.infer(tok[perm], lengths[perm], spk[perm])]
    assert first_mismatch(mel_p, mel[perm]) is None and first_mismatch(lin_p, lin[perm]) is None and first_mismatch(al_p, al[perm]) is None
    # a batch of five of them, in another order: other neighbours, other XCD loads (one or two utterances per XCD: the row-broadcast task form)
    sub = np.array([17, 3, 30, 8, 21])
    mel_s, lin_s, al_s = [x.cpu().numpy() for x in m.infer(tok[sub], lengths[sub], spk[sub])]
    assert first_mismatch(mel_s, mel[sub]) is None and first_mismatch(lin_s, lin[sub]) is None and first_mismatch(al_s, al[sub]) is None


@pytest.mark.parametrize("enc_depth,post_depth,N,iters", [(1, 6, 3, 6), (8, 2, 5, 4), (4, 4, 17, 200)])
def test_highway_stack_depths_and_ragged_row_tiles(torch_cuda, oracle, enc_depth, post_depth, N, iters):
    """tc_highway_stack_kernel (a CBHG's highway layers in one launch, modules.py:40-41, 83-89) at other depths than the default four
    (1 .. 8 layers, the kernel's argument limit) and with row counts that end inside a row tile; the last case has 17 x 1000 = 17 000
    post-net rows: the 64-row instantiation with a ragged last tile (the small cases run the 32-row one).  Bit for bit against the
    checker, and against the one-launch-per-layer form."""
    hp = _hp(max_iters=iters, enc_bank_size=3, post_bank_size=2, num_freq=65, enc_highway_depth=enc_depth, post_highway_depth=post_depth)
    T = 29
    rng = np.random.RandomState(enc_depth * 10 + post_depth)
    lengths = [T] + [int(x) for x in rng.randint(3, T + 1, N - 1)]
    d, blob, tok, ln, spk, m = _case(oracle, hp, N, T, lengths, seed=90 + enc_depth)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    mel, lin, al = m.infer(tok, ln, spk)
    assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
    assert first_mismatch(lin.cpu().numpy(), lin_o) is None, ("linear", first_mismatch(lin.cpu().numpy(), lin_o))
    m.set_option("highway_stack", 0)
    mel_b, lin_b, _ = m.infer(tok, ln, spk)
    assert first_mismatch(mel.cpu().numpy(), mel_b.cpu().numpy()) is None and first_mismatch(lin.cpu().numpy(), lin_b.cpu().numpy()) is None


@pytest.mark.parametrize("N,T", [(26, 200), (9, 330), (32, 140), (32, 330)])
def test_resident_decoder_long_inputs(torch_cuda, oracle, N, T):
    """tc_decoder_x_kernel with inputs longer than 128 tokens: the attention block's score / context tasks no longer fit one pass of the
    workgroup's 512 threads (the branches behind `task0 > 0`), the recurrence walks more than two 64-step blocks (T = 330: the
    one-wave-per-utterance form), key / memory tables grow towards the LDS limit.  The library's own choice = the forced one where the
    kernel fits; where it does not (B = 32 x 330 tokens: the LDS carve), decoder_groups = 32 says so and the split kernel serves.  Bit for bit."""
    from twvk_amd._lib import TwvError
    hp = _hp(max_iters=3, enc_bank_size=3, post_bank_size=2, num_freq=65)
    rng = np.random.RandomState(N + T)
    lengths = [T] + [int(x) for x in rng.randint(T // 3, T + 1, N - 1)]
    d, blob, tok, ln, spk, m = _case(oracle, hp, N, T, lengths, seed=77)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    for groups in (0, 32):
        m.set_option("decoder_groups", groups)
        fits = not (N == 32 and T == 330)
        assert m.decoder_kernel_name(N, T) == ("tc_decoder_x_kernel" if fits else "tc_decoder_g_kernel")       # (what 32 would ask for and not get)
        try:
            mel, lin, al = m.infer(tok, ln, spk)
        except TwvError as e:
            assert groups == 32 and not fits and "XCD-local decoder" in str(e), (groups, str(e))
            continue
        assert fits or groups == 0
        assert first_mismatch(al.cpu().numpy(), al_o) is None, (groups, "alignments", first_mismatch(al.cpu().numpy(), al_o))
        assert first_mismatch(mel.cpu().numpy(), mel_o) is None, (groups, "mel", first_mismatch(mel.cpu().numpy(), mel_o))


@pytest.mark.parametrize("layers,r,N", [(2, 3, 5), (1, 2, 26), (2, 5, 17)])
def test_resident_decoder_other_sizes(torch_cuda, oracle, layers, r, N):
    """tc_decoder_x_kernel outside the folded hparams-default instantiation: other decoder depths / reduction factors (run-time sizes), one
    utterance per XCD with idle XCDs (N = 5), four and three utterances per XCD with ragged XCD loads (N = 26, 17: the matrix-core task
    form), ragged lengths; forced (decoder_groups = 32) and the library's own choice (the same kernel up to batch 32).  Bit for bit against the checker."""
    hp = _hp(max_iters=5, enc_bank_size=3, post_bank_size=2, num_freq=65, dec_layer_num=layers, reduction_factor=r)
    T = 45
    rng = np.random.RandomState(layers * 10 + r)
    lengths = [T] + [int(x) for x in rng.randint(2, T + 1, N - 1)]
    d, blob, tok, ln, spk, m = _case(oracle, hp, N, T, lengths, seed=41 + layers)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    for groups in (32, 0):
        m.set_option("decoder_groups", groups)
        for _ in range(2):                                              # the second pass reuses the exchange buffers and tickets
            mel, lin, al = m.infer(tok, ln, spk)
            assert first_mismatch(al.cpu().numpy(), al_o) is None, (groups, "alignments", first_mismatch(al.cpu().numpy(), al_o))
            assert first_mismatch(mel.cpu().numpy(), mel_o) is None, (groups, "mel", first_mismatch(mel.cpu().numpy(), mel_o))
            assert first_mismatch(lin.cpu().numpy(), lin_o) is None


def test_xcd_local_decoder_at_bench_geometry(torch_cuda, oracle):
    """decoder_groups = 32: every XCD's 32 workgroups hold the decoder in registers and serve 4 utterances; B = 32, 25 steps, ragged lengths"""
    hp = _hp(max_iters=25)
    N, T = 32, 101
    lengths = [T - (i % 7) * 9 for i in range(N)]
    d, blob, tok, ln, spk, m = _case(oracle, hp, N, T, lengths, seed=23)
    m.set_option("decoder_groups", 32)
    mel_o, lin_o, al_o = oracle.taco_infer(d, blob, tok, ln, spk)
    mel, lin, al = m.infer(tok, ln, spk)
    assert first_mismatch(al.cpu().numpy(), al_o) is None, ("alignments", first_mismatch(al.cpu().numpy(), al_o))
    assert first_mismatch(mel.cpu().numpy(), mel_o) is None, ("mel", first_mismatch(mel.cpu().numpy(), mel_o))
    assert first_mismatch(lin.cpu().numpy(), lin_o) is None
