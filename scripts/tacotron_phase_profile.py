#!/usr/bin/env python
"""per-phase s_memtime breakdown of one Tacotron decoder step (utterance 0's workgroup), C3 config"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import twvk_amd
from twvk_amd.tacotron import Tacotron
exec(open(os.path.join(ROOT, "scripts", "tacotron_bench.py")).read().split("ap = argparse")[0].split("def random_tensors")[1].join(["def random_tensors", ""]) if False else "")
from importlib.machinery import SourceFileLoader
hp = twvk_amd.default_hparams()
m = Tacotron(hp, num_speakers=2)
src = open(os.path.join(ROOT, "scripts", "tacotron_bench.py")).read()
ns = {}
exec(src[src.index("def random_tensors"):src.index("ap = argparse")], {"np": np}, ns)
m.load_weights(ns["random_tensors"](m.specs))
rng = np.random.RandomState(1)
N, T = 32, 101
tok = rng.randint(2, 80, (N, T)).astype(np.int32); tok[:, -1] = 1
ln = np.full(N, T, np.int32); spk = (np.arange(N) % 2).astype(np.int32)
prof = torch.zeros(hp.max_iters * 16, dtype=torch.int64, device="cuda")
m._L.twv_tacotron_set_profile_buffer(m._h, C.c_void_p(prof.data_ptr()))
m.infer(tok, ln, spk); m.infer(tok, ln, spk); torch.cuda.synchronize()
p = prof.cpu().numpy().reshape(-1, 16)[20:180, [0, 1, 2, 3, 4, 5, 6, 7, 9]].astype(np.float64)
d = np.diff(p, axis=1).mean(0)
names = ["prenet", "attn GRU", "query", "score", "recurrence", "context", "proj", "res GRUs+output"]
tot = (p[:, 8] - p[:, 0]).mean()
for nme, v in zip(names, d):
    print("%-12s %8.0f ticks %5.1f%%" % (nme, v, 100 * v / tot))
q = prof.cpu().numpy().reshape(-1, 16)[20:180].astype(np.float64)
print("proj stage (wave 0 of WG 0): params %.0f | dots %.0f | prefetch issue %.0f | barrier %.0f | combine+publish %.0f | gather %.0f | barrier+post -> stamp7 %.0f" % tuple((q[:, b] - q[:, a]).mean() for a, b in ((6, 10), (10, 11), (11, 12), (12, 13), (13, 14), (14, 15), (15, 7))))
print("step total %8.0f ticks (s_memtime @100 MHz -> %.1f us)" % (tot, tot / 100.0))
