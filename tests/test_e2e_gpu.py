"""C5: text -> mel -> wave on the GPU equals the CPU oracle's Tacotron restatement chained into its WaveNet restatement."""
import numpy as np
import pytest

from helpers import first_mismatch, make_case, make_model, mol_uniforms

pytestmark = pytest.mark.gpu


def test_text_to_wave_matches_chained_oracles(oracle):
    import twvk_amd
    from twvk_amd.tacotron import Synthesizer
    from twvk_amd.e2e import text_to_wave, attention_trim_frames
    hp = twvk_amd.default_hparams()
    hp.max_iters, hp.num_freq = 3, 65                                    # 15 mel frames -> 4500 samples at hop 300
    td = oracle.taco_dims(max_iters=3, num_freq=65)
    tt = oracle.taco_random_tensors(td, seed=5)
    syn = Synthesizer(); syn.load(tt, num_speakers=2, hparams=hp)
    dil = [1, 2, 4, 8]
    wd, wt, wblob = make_case(oracle, dil, S=64, scale=0.2)
    voc = make_model(2, dil, wt, S=64)
    tokens = [[5, 9, 33, 12, 1], [7, 7, 1]]
    spk = np.array([1, 0], np.int32)
    n_frames = 4
    T = n_frames * 300
    u = mol_uniforms(2, T, 10, seed=9)
    out = text_to_wave(syn, voc, tokens, spk, u, n_frames=n_frames)
    # oracle chain: Tacotron restatement -> (mel hand-off, synthesizer.py:279 / generate.py:151-155) -> WaveNet restatement
    tok = np.array([[5, 9, 33, 12, 1], [7, 7, 1, 0, 0]], np.int32)
    mel_o, _, al_o = oracle.taco_infer(td, oracle.taco_blob(td, tt), tok, np.array([5, 3], np.int32), spk)
    assert first_mismatch(out["mel"].cpu().numpy(), mel_o[:, :n_frames]) is None
    U = oracle.upsample(wd, wblob, mel_o[:, :n_frames])
    ref = oracle.generate_mol(wd, wblob, oracle.State(wd, 2), U, spk, np.zeros(2, np.float32), u)
    got = out["audio"].cpu().numpy()
    assert got.shape == (2, T)
    assert first_mismatch(got, ref) is None, first_mismatch(got, ref)
    # host-side trim rule (synthesizer.py:232-256) runs on the alignments without error and stays inside the decode
    k = attention_trim_frames(al_o[0][:5], 5, hp.reduction_factor)
    assert 3 <= k <= hp.reduction_factor * hp.max_iters + 3


def test_text_to_wave_at_default_dims_on_the_xcd_kernel(oracle):
    """BASELINE configs[4] at the geometry the product runs: default-dim Tacotron (hparams.py:126-165; 25 decoder steps = 125 mel
    frames) hands its mel -- in HBM -- to the 30-layer S = 512 MoL vocoder, which is served by the XCD-per-stream kernel with
    create_upsample + the lc projections inside the launch (fused conditioning asserted).  8 utterances in one batch (one per
    XCD), then utterance 0 alone as B = 1 (configs[4]'s one utterance per GPU: one XCD busy).  Mel and every sample bit for bit
    against the chained oracles (synthesizer.py:279-280 -> generate.py:151-155)."""
    import twvk_amd
    from twvk_amd.tacotron import Synthesizer
    from twvk_amd.e2e import text_to_wave
    hp = twvk_amd.default_hparams()
    hp.max_iters = 25
    td = oracle.taco_dims(max_iters=25)
    tt = oracle.taco_random_tensors(td, seed=31)
    syn = Synthesizer(); syn.load(tt, num_speakers=2, hparams=hp)
    dil = [2 ** i for i in range(10)] * 3
    wd, wt, wblob = make_case(oracle, dil, seed=7)
    B, n_frames = 8, 40                                                  # 40 of the 125 frames -> 12 000 samples per utterance
    T = n_frames * 300
    rng = np.random.RandomState(12)
    lengths = [40, 33, 40, 21, 37, 40, 9, 30]
    tokens = [list(rng.randint(2, 80, ln - 1)) + [1] for ln in lengths]
    spk = (np.arange(B) % 2).astype(np.int32)
    u = mol_uniforms(B, T, 10, seed=13)
    # ---- oracle chain
    tok = np.zeros((B, 40), np.int32)
    for i, t in enumerate(tokens):
        tok[i, :len(t)] = t
    mel_o, _, _ = oracle.taco_infer(td, oracle.taco_blob(td, tt), tok, np.asarray(lengths, np.int32), spk, want_linear=False)
    assert mel_o.shape == (B, 125, 80)
    U = oracle.upsample(wd, wblob, mel_o[:, :n_frames])
    oracle.set_threads(min(B, oracle.set_threads(1)))
    try:
        ref = oracle.generate_mol(wd, wblob, oracle.State(wd, B), U, spk, np.zeros(B, np.float32), u)
    finally:
        oracle.set_threads(1)
    # ---- 8 utterances, one launch
    voc = make_model(B, dil, wt)
    assert voc.fused_conditioning(), "the 30-layer S=512 MoL vocoder must be served by the XCD-per-stream kernel"
    out = text_to_wave(syn, voc, tokens, spk, u, n_frames=n_frames)
    assert out["input_lengths"] == lengths
    assert first_mismatch(out["mel"].cpu().numpy(), mel_o[:, :n_frames]) is None
    assert np.abs(out["mel"].cpu().numpy() - mel_o[:, :n_frames]).max() <= 1e-4        # north_star's float tolerance (bit-exact is stricter)
    got = out["audio"].cpu().numpy()
    assert got.shape == (B, T)
    assert first_mismatch(got, ref) is None, first_mismatch(got, ref)
    # ---- one utterance per GPU (B = 1): Tacotron and vocoder at batch 1, seven XCDs idle
    voc1 = make_model(1, dil, wt)
    assert voc1.fused_conditioning()
    out1 = text_to_wave(syn, voc1, tokens[:1], spk[:1], u[:1], n_frames=n_frames)
    assert first_mismatch(out1["mel"].cpu().numpy(), mel_o[:1, :n_frames]) is None
    assert first_mismatch(out1["audio"].cpu().numpy(), ref[:1]) is None


@pytest.mark.gpu
def test_driver_hooks_in_one_fresh_process():
    """build() loads the C-ABI library before anything has imported torch; smoke() must still see the GPU afterwards
    (PyTorch-ROCm carries its own HIP runtime: a second runtime in the process reports "no ROCm-capable device")."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout
