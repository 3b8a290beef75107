#!/usr/bin/env python
"""Per-stage anatomy of the XCD-local Tacotron decoder kernel (tuning aid): s_memtime stamps of slice 0 of XCD 0 in decoder step 3."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import twvk_amd
from twvk_amd.tacotron import Tacotron
from twvk_amd import _lib
sys.argv = [sys.argv[0]]
exec(open(os.path.join(ROOT, "scripts", "tacotron_bench.py")).read().split("ap = argparse")[0])
hp = twvk_amd.default_hparams()
m = Tacotron(hp, num_speakers=2); m.load_weights(random_tensors(m.specs)); m.set_option("decoder_groups", 32)
rng = np.random.RandomState(1); N, T = 32, 101
tok = rng.randint(2, 80, (N, T)).astype(np.int32); tok[:, -1] = 1
ln = np.full(N, T, np.int32); spk = (np.arange(N) % 2).astype(np.int32)
prof = torch.zeros(16 * 16 + 64, dtype=torch.int64, device="cuda")
m.infer(tok, ln, spk, want_linear=False)
_lib.check(m._L.twv_tacotron_set_profile_buffer(m._h, C.c_void_p(prof.data_ptr())))
m.infer(tok, ln, spk, want_linear=False); torch.cuda.synchronize()
p = prof.cpu().numpy().astype(np.int64)[:16 * 16].reshape(16, 16)
names = ["prenet1", "prenet2", "aGRU gates", "aGRU cand", "query+attention", "proj", "rGRU0 gates", "rGRU0 cand", "rGRU1 gates", "rGRU1 cand", "out"]
tot = 0; exch = 0
for st in range(11):
    r = p[st]; d = lambda a, b: (r[b] - r[a]) / 2.4e3
    print("%-16s dots %.2f | barrier %.2f | combine+publish %.2f | gather %.2f + update %.2f | barrier %.2f | stage %.2f us" % (names[st], d(0, 1), d(1, 2), d(2, 3), d(3, 12), d(12, 4), d(4, 5), d(0, 11)))
    if st == 4:
        print("   attention: score %.2f | barrier+sum+publish %.2f | gather p %.2f | recurrence %.2f | context+publish %.2f | gather ctx %.2f" % (d(5, 6), d(6, 7), d(7, 8), d(8, 9), d(9, 10), d(10, 11)))
    tot += d(0, 11); exch += d(3, 12) + (d(7, 8) + d(10, 11) if st == 4 else 0.0)
print("step total %.2f us = %.2f us in the 13 exchanges (publish -> every slice has polled the values in) + %.2f us of everything else "
      "(stamps included: seven to fourteen per stage, ~0.05 us each)" % (tot, exch, tot - exch))
