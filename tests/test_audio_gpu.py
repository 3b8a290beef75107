"""-m gpu: spectrogram -> waveform (denormalise, Griffin-Lim with librosa's stft/istft conventions, inverse pre-emphasis) on the
device vs the float64 numpy restatement (oracle/audio_np.py).  Floating-point FFT work: tolerance parity, written at each assert."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hp(**kw):
    import twvk_amd
    hp = twvk_amd.default_hparams()
    for k, v in kw.items():
        setattr(hp, k, v)
    return hp


def _ref(lin, u, hp, iters):
    from oracle import audio_np as A
    return np.stack([A.inv_linear_spectrogram(lin[b].T, u[b].T, iters=iters, power=hp.power, ref_level_db=hp.ref_level_db, n_fft=hp.fft_size,
                                              hop=hp.hop_size, win_length=hp.win_size, preemphasis=hp.preemphasis) for b in range(lin.shape[0])])


@pytest.mark.parametrize("iters", [0, 3])
def test_inv_linear_spectrogram_matches_numpy(iters):
    from twvk_amd.audio import inv_linear_spectrogram
    hp = _hp(griffin_lim_iters=iters)
    rng = np.random.RandomState(iters)
    B, T = 2, 14
    lin = rng.uniform(-4.5, 4.5, (B, T, hp.num_freq)).astype(np.float32)       # beyond +-max_abs_value: exercises the clipping
    u = rng.rand(B, T, hp.num_freq).astype(np.float32)
    got = inv_linear_spectrogram(lin, hp, uniforms=u).cpu().numpy()
    want = _ref(lin, u, hp, iters)
    assert got.shape == want.shape == (B, hp.hop_size * (T - 1))
    # tolerance: float32 FFTs / transcendental functions vs float64; 2e-4 of the peak amplitude after <= 3 projections
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()


def test_griffin_lim_sixty_iterations_small_fft():
    """hparams.griffin_lim_iters = 60 on a smaller transform (keeps the float64 restatement fast); the projections do not amplify
    float32 round-off: still within 2e-3 of the peak"""
    from twvk_amd.audio import inv_linear_spectrogram
    hp = _hp(griffin_lim_iters=60, fft_size=256, win_size=200, hop_size=50)
    hp.num_freq = 129
    rng = np.random.RandomState(7)
    lin = rng.uniform(-4, 4, (1, 30, 129)).astype(np.float32)
    u = rng.rand(1, 30, 129).astype(np.float32)
    got = inv_linear_spectrogram(lin, hp, uniforms=u).cpu().numpy()
    want = _ref(lin, u, hp, 60)
    assert np.abs(got - want).max() <= 2e-3 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()


def test_synthesizer_linear_to_wave_and_save(tmp_path):
    """text -> linear spectrogram (Tacotron) -> wave (Griffin-Lim) -> .wav, the reference synthesizer's own output path"""
    import twvk_amd
    from oracle import oracle as O
    from twvk_amd.tacotron import Synthesizer
    from twvk_amd.audio import inv_linear_spectrogram, save_wav
    hp = _hp(max_iters=4, griffin_lim_iters=2)
    td = O.taco_dims(max_iters=4, num_freq=hp.num_freq)
    syn = Synthesizer(); syn.load(O.taco_random_tensors(td, seed=5), num_speakers=2, hparams=hp)
    out = syn.infer([[5, 9, 33, 12, 1]], speaker_ids=[1])
    wav = inv_linear_spectrogram(out["linear"], hp, seed=3)
    assert wav.shape == (1, hp.hop_size * (4 * hp.reduction_factor - 1)) and np.isfinite(wav.cpu().numpy()).all()
    path = str(tmp_path / "a.wav")
    save_wav(wav[0], path, hp.sample_rate)
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    ref = wav[0].cpu().numpy().copy()
    ref *= 32767 / max(0.01, np.max(np.abs(ref)))                      # utils/audio.py:15 (a quiet signal hits the 0.01 floor)
    assert sr == hp.sample_rate and data.dtype == np.int16 and np.array_equal(data, ref.astype(np.int16))
