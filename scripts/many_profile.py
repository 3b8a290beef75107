#!/usr/bin/env python
"""Where the waves of the many-streams XCD kernel (wn_xcd_many_kernel, wait-accounting build) spend a generation step: per role and
wave of XCD 0 the s_memtime ticks per step and the share spent inside each kind of poll.  `python scripts/many_profile.py [B] [T]`."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_model, mol_uniforms
import twvk_amd
from twvk_amd import weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
dil = [2 ** i for i in range(10)] * 3
specs = W.tensor_specs(len(dil), 32, 32, 512, 256, 30, True, 32, True, 32, 2, 80, (5, 5, 12))
m = make_model(B, dil, W.random_tensors(specs, seed=0, scale=0.05), xcd_many=1)
rng = np.random.RandomState(1)
mel = torch.from_numpy(rng.uniform(-4, 4, (B, T // 300, 80)).astype(np.float32)).cuda()
gc = (np.arange(B) % 2).astype(np.int32); seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
u = torch.from_numpy(mol_uniforms(B, T, 10)).cuda()
U = m.create_upsample(mel)
m.generate(U, gc, seed_in, u); torch.cuda.synchronize()
prof = torch.zeros(64 * 64, dtype=torch.int64, device="cuda")
twvk_amd._lib.check(m._L.twv_wavenet_set_profile_buffer(m._h, C.c_void_p(prof.data_ptr()), 64))
m.queue_initializer()
import time
t0 = time.perf_counter(); m.generate(U, gc, seed_in, u); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("B %d, T %d: %.2f us/step (accounting build)" % (B, T, dt / T * 1e6))
p = prof.cpu().numpy().reshape(-1, 8, 8)
names = {0: ("chain", ["PG/LG", "input", "sampler", "-"]), 4: ("service", ["layer-in", "lc-ring", "-", "-"]), 8: ("skip", ["nap", "z", "total-in", "-"]),
         16: ("conv1", ["nap", "h1", "count", "-"]), 24: ("lc", ["throttle", "row+stages", "dots", "publish"])}
for role in range(28):
    base = max(k for k in names if k <= role)
    nm, cats = names[base]
    rows = []
    for w in range(8):
        tot = float(p[role, w, 0])
        if tot <= 0:
            continue
        waits = [float(p[role, w, 1 + i]) for i in range(4)]
        busy = tot - sum(waits[i] for i in range(4) if cats[i] != "-" and not (nm == "chain" and i == 2))
        rows.append("w%d %5.0f t/step busy %4.1f%% [%s]" % (w, tot / T, 100 * busy / tot, " ".join("%s %4.1f%%" % (cats[i], 100 * waits[i] / tot) for i in range(4) if cats[i] != "-")))
    if rows:
        print("%-8s %2d: " % (nm, role - base) + " | ".join(rows[:8]))

tr = prof.cpu().numpy()[2048:2048 + 512].astype(np.int64)
t0 = tr[0]
if t0 > 0:
    print("timeline of step 2000, stream 0 of XCD 0 (us after the head pushed the causal layer's output; 10 ns clock):")
    print("  z stored, layers 0..29:", " ".join("%.2f" % ((tr[64 + l] - t0) / 100.0) for l in range(30)))
    print("  next step's push (= step period): %.2f" % ((tr[1] - t0) / 100.0))
    print("  conv1 h1 seen (g x wave):", " ".join("%.2f" % ((tr[400 + i] - t0) / 100.0) for i in range(64) if tr[400 + i] > 0))
    print("  conv1d_2 partials stored (g x summer):", " ".join("%.2f" % ((tr[480 + i] - t0) / 100.0) for i in range(16) if tr[480 + i] > 0))
    for r in range(8):
        rows = []
        for m in range(8):
            e = tr[128 + (r * 8 + m) * 4: 128 + (r * 8 + m) * 4 + 4]
            if e[1] > 0:
                rows.append("w%d %.2f/%.2f/%.2f" % (m, (e[0] - t0) / 100.0, (e[1] - t0) / 100.0, (e[3] - t0) / 100.0))
        print("  skip %d (group %d half %d) wake/z-seen/total-out: " % (r, r >> 1, r & 1) + " | ".join(rows))
