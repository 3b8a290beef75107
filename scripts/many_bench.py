#!/usr/bin/env python
"""One line for the many-streams XCD kernel: B streams (default 64) x `--seconds` of 24 kHz audio, BASELINE configs[1]'s model, fused
conditioning; HIP-event time of the generation launch.  Used by scripts/profile_generation.sh (rocprofv3 stats / PMC passes)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_model, mol_uniforms
import twvk_amd
from twvk_amd import weights as W
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--seconds", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=2); ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--many", type=int, default=-1, help="1 / 0: force the many-streams kernel on / off at batch <= 32 (option xcd_many); -1: the library's choice")
args = ap.parse_args()
B = args.batch
T = int(args.seconds * 24000) // 300 * 300
dil = [2 ** i for i in range(10)] * 3
specs = W.tensor_specs(len(dil), 32, 32, 512, 256, 30, True, 32, True, 32, 2, 80, (5, 5, 12))
m = make_model(B, dil, W.random_tensors(specs, seed=0, scale=0.05))
if args.many >= 0: m.set_option("xcd_many", args.many)
assert m.fused_conditioning()
rng = np.random.RandomState(1)
mel = torch.from_numpy(rng.uniform(-4, 4, (B, T // 300, 80)).astype(np.float32)).cuda()
gc = (np.arange(B) % 2).astype(np.int32); seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
u = torch.from_numpy(mol_uniforms(B, T, 10)).cuda()
ms = []
for i in range(args.warmup + args.steps):
    m.queue_initializer()
    U = m.create_upsample(mel)
    cond = m._condition(U, gc, T)
    fi = torch.as_tensor(seed_in, device="cuda"); out = torch.empty((B, T), dtype=torch.float32, device="cuda")
    import ctypes as C
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    twvk_amd._lib.check(m._L.twv_wavenet_generate(m._h, C.c_void_p(m._packed.data_ptr()), C.c_void_p(m._state.data_ptr()), C.c_void_p(cond.data_ptr()),
                                                 C.c_void_p(fi.data_ptr()), C.c_void_p(u.data_ptr()), 1.0, B, T, C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(m._status.data_ptr()), None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    e1.record(); torch.cuda.synchronize()
    twvk_amd._lib.check(m._L.twv_wavenet_status(C.c_void_p(m._status.data_ptr()), None))
    if i >= args.warmup:
        ms.append(e0.elapsed_time(e1))
k = float(np.mean(ms))
print(json.dumps({"kernel": "wn_xcd_many_kernel" if (B > 32 or args.many == 1) else "wn_xcd_generate_kernel", "streams": B, "steps_per_launch": T, "kernel_ms": k,
                  "us_per_generation_step": k * 1e3 / T, "samples_per_s": B * T / (k * 1e-3), "finite": bool(torch.isfinite(out).all())}))
