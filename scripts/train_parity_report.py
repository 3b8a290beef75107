"""Training-step parity in numbers: for every test geometry of tests/test_train_gpu.py the HIP gradients and the
float32 torch model's gradients against the float64 torch model, per tensor: e = max|g - g64| / max|g64|.  Prints, per case, the worst
e_hip, the worst e_t32, the worst ratio e_hip / max(e_t32, 1e-7) and the tensors they occur in -- the data behind the bars the tests
assert.  Run on the GPU box: python scripts/train_parity_report.py > gpurun_out/r05_train_parity_report.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch_train_ref as R                      # noqa: E402
from test_train_gpu import _case                 # noqa: E402

CASES = [
    ("small", dict(dil=[1, 2, 4, 1, 2], B=2, Tm=3), False),
    ("one-cycle", dict(dil=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512], B=3, Tm=6, S=128), False),
    ("mol-branches", dict(dil=[1, 2, 4, 1, 2], B=2, Tm=3, ls_bias=-4.0, clip_audio=True), False),
    ("no-bias", dict(dil=[1, 2, 4], B=1, Tm=2, use_bias=False), False),
    ("hop64", dict(dil=[1, 2, 4, 8], B=2, Tm=21, up=(4, 4, 4)), False),
    ("hop32", dict(dil=[1, 2, 4], B=2, Tm=40, up=(2, 4, 4)), False),
    ("no-bias-s128", dict(dil=[1, 2, 4], B=2, Tm=5, S=128, use_bias=False), False),
    ("s256", dict(dil=[1, 2], B=1, Tm=4, S=256), False),
    ("bench-geometry 16 x 3600", dict(dil=[2 ** i for i in range(10)] * 3, B=16, Tm=12, S=512), True),
    ("configs[3] 64 x 7800", dict(dil=[2 ** i for i in range(10)] * 3, B=64, Tm=26, S=512), True),
]
FLOOR = 1e-7          # below this a float32 tensor's own error says nothing (one ulp of its largest element)

print("%-26s %10s %10s %8s   %s" % ("case", "e_hip", "e_t32", "ratio", "worst tensor (by e_hip) | (by ratio)"))
for name, kw, big in CASES:
    tr, tensors, cfg, audio, lc, gc = _case(**kw)
    loss = float(tr.loss_and_gradients(audio, lc, gc).item())
    got = tr.gradients()
    extra = dict(device="cuda:0", matmul_form=True) if big else {}
    l64, g64 = R.loss_and_grads(tensors, cfg, audio, lc, gc, dtype=torch.float64, **extra)
    torch.cuda.empty_cache()
    l32, g32 = R.loss_and_grads(tensors, cfg, audio, lc, gc, dtype=torch.float32, **extra)
    torch.cuda.empty_cache()
    rows = []
    for k in g64:
        scale = max(float(np.abs(g64[k]).max()), 1e-30)
        e_hip = float(np.abs(got[k] - g64[k]).max()) / scale
        e_t32 = float(np.abs(g32[k] - g64[k]).max()) / scale
        rows.append((k, e_hip, e_t32, e_hip / max(e_t32, FLOOR), scale))
    wh = max(rows, key=lambda r: r[1]); wr = max(rows, key=lambda r: r[3]); wt = max(rows, key=lambda r: r[2])
    print("%-26s %10.3g %10.3g %8.2f   %s | %s (e_hip %.3g, e_t32 %.3g)" % (name, wh[1], wt[2], wr[3], wh[0].replace("wavenet/", ""), wr[0].replace("wavenet/", ""), wr[1], wr[2]))
    print("%-26s loss hip %.7f  f32 %.7f  f64 %.7f   |hip-f64| %.2e  |f32-f64| %.2e" % ("", loss, l32, l64, abs(loss - l64), abs(l32 - l64)))
    # distribution: how many tensors above a few thresholds
    for th in (1e-3, 3e-4, 1e-4, 3e-5):
        n = sum(1 for r in rows if r[1] > th)
        print("%-26s   tensors with e_hip > %.0e: %d of %d" % ("", th, n, len(rows)), ("(" + ", ".join(r[0].split("/")[-3] + "/" + r[0].split("/")[-2] for r in rows if r[1] > th)[:160] + ")") if 0 < n <= 6 else "")
    del tr
    torch.cuda.empty_cache()
