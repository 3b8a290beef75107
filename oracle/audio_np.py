"""CPU restatement (numpy, float64) of the reference's spectrogram -> waveform path -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference utils/audio.py:
  :77-92   inv_linear_spectrogram (denormalize -> db_to_amp -> ** power -> Griffin-Lim -> inv_preemphasis)
  :127-137 _griffin_lim (random initial phase, griffin_lim_iters x {stft -> unit phase -> istft})
  :139-146 _stft / _istft = librosa.stft / librosa.istft(n_fft=fft_size, hop_length=hop_size, win_length=win_size)
  :27-30   inv_preemphasis = scipy.signal.lfilter([1], [1, -k], wav)
  :205-206 _db_to_amp, :222-234 _denormalize, :14-17 save_wav
librosa is a third-party dependency that is absent here (requirements.txt pins no version); its published algorithm is
restated [RECALLED-LIBROSA 0.6-0.9]: stft(center=True, pad_mode='reflect'), window = scipy.signal.get_window('hann', win_length,
fftbins=True) zero-padded symmetrically to n_fft, frames at multiples of hop; istft = windowed overlap-add of irfft frames divided
by the window sum-of-squares where it exceeds tiny(float32), then n_fft//2 trimmed from both ends.
Parity unpinned: neither librosa nor the reference can run here.  The random initial phase is injected (uniforms in [0,1))."""
import numpy as np


def hann_padded(win_length, n_fft):
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)        # periodic ("fftbins") Hann
    lpad = (n_fft - win_length) // 2
    return np.pad(w, (lpad, n_fft - win_length - lpad))


def stft(y, n_fft, hop, win_length):
    """librosa.stft: (1 + n_fft/2, 1 + len(y)//hop) complex"""
    w = hann_padded(win_length, n_fft)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(yp) - n_fft) // hop
    frames = np.stack([yp[i * hop:i * hop + n_fft] * w for i in range(n_frames)], axis=1)
    return np.fft.rfft(frames, axis=0)


def istft(D, hop, win_length):
    """librosa.istft: length hop * (n_frames - 1)"""
    n_fft = 2 * (D.shape[0] - 1)
    n_frames = D.shape[1]
    w = hann_padded(win_length, n_fft)
    y = np.zeros(n_fft + hop * (n_frames - 1))
    wss = np.zeros_like(y)
    for i in range(n_frames):
        y[i * hop:i * hop + n_fft] += w * np.fft.irfft(D[:, i], n_fft)
        wss[i * hop:i * hop + n_fft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:len(y) - n_fft // 2]


def denormalize(D, max_abs_value=4.0, min_level_db=-100.0):
    """utils/audio.py:222-227 (allow_clipping_in_normalization, symmetric_mels: the hparams defaults)"""
    return ((np.clip(D, -max_abs_value, max_abs_value) + max_abs_value) * -min_level_db / (2 * max_abs_value)) + min_level_db


def db_to_amp(x):
    return np.power(10.0, x * 0.05)


def griffin_lim(S, uniforms, iters, n_fft, hop, win_length):
    """utils/audio.py:127-137; S (num_freq, T) magnitudes ** power, uniforms (num_freq, T) replace np.random.rand"""
    angles = np.exp(2j * np.pi * uniforms)
    Sc = np.abs(S).astype(np.complex128)
    y = istft(Sc * angles, hop, win_length)
    for _ in range(iters):
        angles = np.exp(1j * np.angle(stft(y, n_fft, hop, win_length)))
        y = istft(Sc * angles, hop, win_length)
    return y


def inv_preemphasis(wav, k):
    """scipy.signal.lfilter([1], [1, -k], wav): y[n] = x[n] + k * y[n-1]"""
    y = np.empty_like(wav)
    acc = 0.0
    for i, x in enumerate(wav):
        acc = x + k * acc
        y[i] = acc
    return y


def inv_linear_spectrogram(lin, uniforms, iters=60, power=1.5, ref_level_db=20.0, n_fft=2048, hop=300, win_length=1200, preemphasis=0.97):
    """utils/audio.py:77-92 for lin (num_freq, T) (synthesizer.py:258 passes wav.T)"""
    S = db_to_amp(denormalize(np.asarray(lin, np.float64)) + ref_level_db)
    return inv_preemphasis(griffin_lim(S ** power, uniforms, iters, n_fft, hop, win_length), preemphasis)
