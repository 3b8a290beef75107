#!/usr/bin/env python
"""Times the mu-law-256 variant of BASELINE configs[1] (one-hot input, 256-way softmax output) at batch B.
usage: python scripts/mulaw_bench.py [--batch 8] [--steps 6000] [--xcd -1|0|1] [--layers 30] [--check 600]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import twvk_amd
from twvk_amd import weights as W
from twvk_amd.wavenet import WaveNetModel

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=6000)
ap.add_argument("--xcd", type=int, default=-1)
ap.add_argument("--layers", type=int, default=30)
ap.add_argument("--check", type=int, default=600, help="steps compared with the CPU checker (0 = none)")
ap.add_argument("--temperature", type=float, default=1.0)
args = ap.parse_args()
hp = twvk_amd.default_hparams()
dev = "cuda:0"
B = args.batch
dil = ([2 ** i for i in range(10)] * 5)[:args.layers]
T = args.steps // hp.hop_size * hp.hop_size
m = WaveNetModel(B, dil, hp.filter_width, hp.residual_channels, hp.dilation_channels, hp.skip_channels, quantization_channels=256,
                 out_channels=hp.out_channels, use_biases=hp.use_biases, scalar_input=False, initial_filter_width=hp.initial_filter_width,
                 global_condition_channels=hp.gc_channels, global_condition_cardinality=2, local_condition_channels=hp.num_mels,
                 upsample_factor=hp.upsample_factor, train_mode=False, device=dev)
if args.xcd >= 0:
    m.set_option("xcd", args.xcd)
tensors = W.random_tensors(m.specs, seed=0, scale=0.05)
m.load_weights(tensors)
rq = np.random.RandomState(91)
mel = rq.uniform(-4, 4, (B, T // hp.hop_size, hp.num_mels)).astype(np.float32)
u = rq.random_sample((B, T))
fq = rq.randint(256, size=B).astype(np.int32)
gc = (np.arange(B) % 2).astype(np.int32)
melq = torch.from_numpy(mel).to(dev); uq = torch.from_numpy(u).to(dev)
Uq = m.create_upsample(melq)
m.generate(Uq[:, :600].contiguous(), gc, fq, uq[:, :600].contiguous(), temperature=args.temperature)
m.queue_initializer()
torch.cuda.synchronize()
q0 = time.perf_counter()
oq = m.generate(Uq, gc, fq, uq, temperature=args.temperature)
torch.cuda.synchronize()
qdt = time.perf_counter() - q0
res = {"streams": B, "layers": len(dil), "steps": T, "samples_per_s": B * T / qdt, "us_per_generation_step": qdt / T * 1e6,
       "realtime_factor_per_stream": T / qdt / hp.sample_rate, "fused_conditioning": bool(m.fused_conditioning()),
       "classes_drawn": int(torch.unique(oq).numel())}
if args.check:
    from oracle import oracle as O
    n = min(args.check, T)
    d = O.make_dims(dil, scalar_input=False, Q=256)
    blob = O.blob_from_tensors(d, tensors)
    Uo = O.upsample(d, blob, mel[:, :(n + hp.hop_size - 1) // hp.hop_size])[:, :n]
    O.set_threads(min(B, O.set_threads(1)))
    want = O.generate_mulaw(d, blob, O.State(d, B), Uo, gc, fq, u[:, :n], args.temperature)
    O.set_threads(1)
    got = oq[:, :n].cpu().numpy()
    res["checked_steps"] = n
    res["bit_exact"] = bool(np.array_equal(got, want))
    if not res["bit_exact"]:
        bad = np.argwhere(got != want)
        res["first_mismatch"] = [int(v) for v in bad[0]]
print(json.dumps(res))
