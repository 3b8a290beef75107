"""XCD kernel with more than 30 layers (second chain workgroup, LDS-resident early tiles) vs the CPU checker + step time."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_case, make_model, mol_uniforms, first_mismatch
from oracle import oracle as O

B = int(os.environ.get("XB", 2)); T = int(os.environ.get("XT", 700)); NL = int(os.environ.get("XNL", 50)); TL = int(os.environ.get("XTL", 12000))
dil = ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 6)[:NL]
d, tensors, blob = make_case(O, dil)
rng = np.random.RandomState(1)
Tm = (max(T, TL) + 299) // 300
mel = rng.uniform(-4, 4, (B, Tm, 80)).astype(np.float32)
gc = (np.arange(B) % 2).astype(np.int32)
seed = rng.uniform(-1, 1, B).astype(np.float32)
u = mol_uniforms(B, max(T, TL), 10)
m = make_model(B, dil, tensors)
print("NL", NL, "B", B, "fused conditioning (xcd path):", m.fused_conditioning(), flush=True)
got = m.generate(m.create_upsample(mel), gc, seed, u[:, :T]).cpu().numpy()
O.set_threads(min(B, O.set_threads(1)))
want = O.generate_mol(d, blob, O.State(d, B), O.upsample(d, blob, mel)[:, :T], gc, seed, u[:, :T])
print("xcd vs oracle:", first_mismatch(got, want), flush=True)
for name, mm in (("xcd", m), ("generic", make_model(B, dil, tensors, xcd=0))):
    U = mm.create_upsample(mel)
    if name == "generic": U = U[:, :TL].contiguous()
    mm.queue_initializer(); mm.generate(U, gc, seed, u[:, :TL]); torch.cuda.synchronize(); t0 = time.time()
    mm.generate(U, gc, seed, u[:, :TL]); torch.cuda.synchronize(); dt = time.time() - t0
    print("%-8s NL=%d B=%d T=%d: %.1f ms -> %.2f us/step, %.0f samples/s" % (name, NL, B, TL, dt * 1e3, dt / TL * 1e6, B * TL / dt), flush=True)
