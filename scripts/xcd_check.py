"""XCD-per-stream kernel vs the generic kernel vs the CPU checker at the bench geometry (B=8, 30 layers), plus step time."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_case, make_model, mol_uniforms, first_mismatch
from oracle import oracle as O

B = int(os.environ.get("XB", 8)); T = int(os.environ.get("XT", 3000)); TL = int(os.environ.get("XTL", 24000))
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3
d, tensors, blob = make_case(O, dil)
rng = np.random.RandomState(1)
Tm = (max(T, TL) + 299) // 300
mel = rng.uniform(-4, 4, (B, Tm, 80)).astype(np.float32)
gc = (np.arange(B) % 2).astype(np.int32)
seed = rng.uniform(-1, 1, B).astype(np.float32)
u = mol_uniforms(B, max(T, TL), 10)

m = make_model(B, dil, tensors)
print("fused conditioning (xcd path):", m.fused_conditioning(), flush=True)
U = m.create_upsample(mel)
t0 = time.time(); got = m.generate(U, gc, seed, u[:, :T]).cpu().numpy(); print("xcd   T=%d: %.3f s" % (T, time.time() - t0), flush=True)
m0 = make_model(B, dil, tensors, xcd=0)
U0 = m0.create_upsample(mel)
old = m0.generate(U0[:, :T].contiguous(), gc, seed, u[:, :T]).cpu().numpy()
print("xcd vs generic kernel:", first_mismatch(got, old), flush=True)
if os.environ.get("XORACLE", "1") == "1":
    t0 = time.time()
    want = O.generate_mol(d, blob, O.State(d, B), O.upsample(d, blob, mel)[:, :T], gc, seed, u[:, :T])
    print("oracle %.1f s; xcd vs oracle:" % (time.time() - t0), first_mismatch(got, want), " generic vs oracle:", first_mismatch(old, want), flush=True)
# timing
for name, mm, UU in (("xcd", m, U), ("generic", m0, U0[:, :TL].contiguous())):
    mm.queue_initializer()
    mm.generate(UU, gc, seed, u[:, :TL])
    torch.cuda.synchronize(); t0 = time.time()
    out = mm.generate(UU, gc, seed, u[:, :TL]); torch.cuda.synchronize(); dt = time.time() - t0
    print("%-8s B=%d T=%d: %.1f ms -> %.2f us/step, %.0f samples/s" % (name, B, TL, dt * 1e3, dt / TL * 1e6, B * TL / dt), flush=True)
