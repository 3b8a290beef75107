"""wavenet/ops.py:22-47 mu-law codec on the device (HIP C-ABI); torch tensors in, torch tensors out."""
import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def mu_law_encode(audio, quantization_channels, device="cuda:0"):
    a = torch.as_tensor(audio, dtype=torch.float32, device=device).contiguous()
    with torch.cuda.device(a.device):
        out = torch.empty(a.shape, dtype=torch.int32, device=a.device)
        _lib.check(_lib.lib().twv_mu_law_encode(_ptr(a), a.numel(), int(quantization_channels), _ptr(out), _stream()))
    return out


def mu_law_decode(output, quantization_channels, quantization=True, device="cuda:0"):
    L = _lib.lib()
    if quantization:
        q = torch.as_tensor(output, dtype=torch.int32, device=device).contiguous()
        with torch.cuda.device(q.device):
            out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
            _lib.check(L.twv_mu_law_decode(_ptr(q), q.numel(), int(quantization_channels), _ptr(out), _stream()))
        return out
    y = torch.as_tensor(output, dtype=torch.float32, device=device).contiguous()
    with torch.cuda.device(y.device):
        out = torch.empty(y.shape, dtype=torch.float32, device=y.device)
        _lib.check(L.twv_mu_law_expand(_ptr(y), y.numel(), int(quantization_channels), _ptr(out), _stream()))
    return out


def sample_categorical(logits, temperature, uniforms, want_proba=False, device="cuda:0", check=True):
    """generate.py:219-231 (+ model.py:243's float64 softmax) on rows of logits: (rows, Q) float32 logits, (rows,) float64
    uniforms in [0,1) -> (rows,) int32 class ids [, (rows, Q) scaled probabilities].  The draw np.random.choice makes from its
    RandomState is an INPUT here, as in the generation kernels.
    A row with a NaN / infinite logit has no distribution: np.random.choice (generate.py:231) raises ValueError there, and so does
    this function (check=True: one host sync); with check=False such a row's class id is -1."""
    y = torch.as_tensor(logits, dtype=torch.float32, device=device).contiguous()
    assert y.dim() == 2
    u = torch.as_tensor(uniforms, dtype=torch.float64, device=device).contiguous()
    assert u.numel() == y.shape[0]
    with torch.cuda.device(y.device):
        out = torch.empty(y.shape[0], dtype=torch.int32, device=y.device)
        proba = torch.empty_like(y) if want_proba else None
        _lib.check(_lib.lib().twv_sample_categorical(_ptr(y), y.shape[0], y.shape[1], float(temperature), _ptr(u), _ptr(out),
                                                     _ptr(proba) if want_proba else None, _stream()))
    if check and out.numel() and int(out.min().item()) < 0:
        raise ValueError("probabilities contain NaN")          # numpy's message (mtrand.RandomState.choice)
    return (out, proba) if want_proba else out


def eval_elementwise(name, x, device="cuda:0"):
    """contract functions evaluated on the device (parity tests)."""
    L = _lib.lib()
    if name in ("exp64", "log64", "exp64_nonpos"):
        fn = {"exp64": 0, "log64": 1, "exp64_nonpos": 2}[name]
        t = torch.as_tensor(x, dtype=torch.float64, device=device).contiguous()
        with torch.cuda.device(t.device):
            out = torch.empty_like(t)
            _lib.check(L.twv_eval_elementwise64(fn, _ptr(t), t.numel(), _ptr(out), _stream()))
        return out
    fn = {"tanh": 0, "sigmoid": 1, "exp": 2, "log": 3, "log1p": 4}[name]
    t = torch.as_tensor(x, dtype=torch.float32, device=device).contiguous()
    with torch.cuda.device(t.device):
        out = torch.empty_like(t)
        _lib.check(L.twv_eval_elementwise(fn, _ptr(t), t.numel(), _ptr(out), _stream()))
    return out


def wav_to_int16(wav, device="cuda:0"):
    """utils/audio.py:14-17 save_wav's peak normalisation + int16 conversion on the device; wav (n,) or (rows, n) -> int16 tensor"""
    a = torch.as_tensor(wav, dtype=torch.float32, device=device).contiguous()
    rows, n = (1, a.numel()) if a.dim() == 1 else (a.shape[0], a.shape[1])
    with torch.cuda.device(a.device):
        out = torch.empty(a.shape, dtype=torch.int16, device=a.device)
        scratch = torch.empty(rows * 64, dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().twv_wav_to_int16(_ptr(a), rows, n, _ptr(out), _ptr(scratch), _stream()))
    return out
