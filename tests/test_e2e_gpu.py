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
