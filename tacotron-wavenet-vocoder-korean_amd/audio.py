"""Host-side mirror of the reference's spectrogram -> waveform helpers (utils/audio.py) over the HIP C-ABI.

    wav = inv_linear_spectrogram(linear, hparams, uniforms)      # utils/audio.py:77-92 (synthesizer.py:258), Griffin-Lim on the GPU
    save_wav(wav, path, hparams.sample_rate)                      # utils/audio.py:14-17 (peak normalisation on the GPU)

`linear` is the (B, T, num_freq) tensor Tacotron.infer returns (the reference transposes one utterance to (num_freq, T) first).
PyTorch is used for device memory and streams only."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .wavenet import _ptr, _stream


def inv_linear_spectrogram(linear, hparams, uniforms=None, seed=None, device="cuda:0"):
    """(B, T, num_freq) normalised linear spectrograms -> (B, hop_size*(T-1)) waveforms.
    uniforms (B, T, num_freq) in [0,1) replace utils/audio.py:131 np.random.rand; drawn from `seed` when absent."""
    if getattr(hparams, "use_lws", False):
        raise NotImplementedError("use_lws=True (hparams.py:15 default False) is not built")
    lin = torch.as_tensor(linear, dtype=torch.float32, device=device).contiguous()
    if lin.dim() == 2:
        lin = lin[None]
    B, T, F = lin.shape
    if F != hparams.fft_size // 2 + 1:
        raise ValueError("last dimension must be fft_size/2 + 1 = %d, got %d" % (hparams.fft_size // 2 + 1, F))
    if uniforms is None:
        uniforms = np.random.RandomState(seed).rand(B, T, F)
    u = torch.as_tensor(uniforms, dtype=torch.float32, device=device).contiguous()
    if tuple(u.shape) != (B, T, F):
        raise ValueError("uniforms must be %s" % ((B, T, F),))
    if not hparams.signal_normalization or not hparams.allow_clipping_in_normalization or not hparams.symmetric_mels:
        raise NotImplementedError("only the default normalisation (clipped, symmetric) is built")
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.twv_griffin_lim_create(hparams.fft_size, hparams.hop_size, hparams.win_size, T, B, C.byref(h)))
    try:
        with torch.cuda.device(lin.device):
            n = L.twv_griffin_lim_samples(h)
            ws = torch.empty(L.twv_griffin_lim_workspace_bytes(h) // 4 + 64, dtype=torch.float32, device=lin.device)
            out = torch.empty((B, n), dtype=torch.float32, device=lin.device)
            _lib.check(L.twv_inv_linear_spectrogram(h, _ptr(lin), _ptr(u), int(hparams.griffin_lim_iters), float(hparams.power),
                                                   float(hparams.ref_level_db), float(hparams.max_abs_value), float(hparams.min_level_db),
                                                   float(hparams.preemphasis) if hparams.preemphasize else 0.0, _ptr(ws), _ptr(out), _stream()))
            torch.cuda.current_stream().synchronize()
    finally:
        L.twv_griffin_lim_destroy(h)
    return out


def save_wav(wav, path, sr):
    """utils/audio.py:14-17"""
    from scipy.io import wavfile
    from .ops import wav_to_int16
    wavfile.write(path, sr, wav_to_int16(wav).cpu().numpy().reshape(-1))
