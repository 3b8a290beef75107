#!/usr/bin/env python
"""In-kernel phase timing of the generation kernel (tuning aid): prints per-phase microseconds of stream 0's chain wave."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import twvk_amd
from twvk_amd.wavenet import WaveNetModel
from twvk_amd import weights as W, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--workers", type=int, default=0)
ap.add_argument("--layers", type=int, default=30)
ap.add_argument("--groups", type=int, default=-1)
ap.add_argument("--no-stamps", action="store_true")
ap.add_argument("--onehot", action="store_true", help="mu-law-256 one-hot input, 256-way softmax output")
args = ap.parse_args()
hp = twvk_amd.default_hparams()
dil = ([2 ** i for i in range(10)] * 5)[:args.layers]
B, T = args.batch, args.steps
m = WaveNetModel(B, dil, 2, 32, 32, 512, out_channels=256 if args.onehot else 30, quantization_channels=256, use_biases=True,
                 scalar_input=not args.onehot, initial_filter_width=32,
                 global_condition_channels=32, global_condition_cardinality=2, local_condition_channels=80,
                 upsample_factor=[5, 5, 12], train_mode=False)
if args.workers: m.set_option("workers", args.workers)
if args.groups >= 0: m.set_option("groups", args.groups)
m.load_weights(W.random_tensors(m.specs, 0, 0.05))
rng = np.random.RandomState(0)
U = torch.from_numpy(rng.uniform(-1, 1, (B, T, 80)).astype(np.float32)).cuda()
u = torch.from_numpy(rng.uniform(1e-5, 1 - 1e-5, (B, T, 11)).astype(np.float32)).cuda()
if args.onehot: u = torch.from_numpy(rng.uniform(0, 1, (B, T))).cuda()
NP = min(T, 2000)
prof = torch.zeros((NP, 80), dtype=torch.int64, device="cuda")
_lib.check(m._L.twv_wavenet_set_profile_buffer(m._h, C.c_void_p(prof.data_ptr()), 0 if args.no_stamps else NP))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
m.generate(U, np.zeros(B, np.int32), np.zeros(B, np.int32 if args.onehot else np.float32), u, check=True)   # warm
m.queue_initializer()
e0.record(); m.generate(U, np.zeros(B, np.int32), np.zeros(B, np.int32 if args.onehot else np.float32), u, check=False); e1.record(); torch.cuda.synchronize()
if args.no_stamps:
    print("no stamps: %.2f us/step (events)" % (e0.elapsed_time(e1) * 1e3 / T)); sys.exit(0)
p = prof.cpu().numpy().astype(np.int64)
s = p[100:NP]   # skip the cold start
ticks = s[-1, 0] - s[0, 0]; wall = (s[-1, 7] - s[0, 7]) / 100e6
mhz = ticks / wall / 1e6
def us(a): return float(np.mean(a)) / mhz
print("total kernel+cond %.2f ms for %d steps -> %.2f us/step (events); s_memtime clock %.0f MHz" % (e0.elapsed_time(e1), T, e0.elapsed_time(e1) * 1e3 / T, mhz))
print("step period        %.2f us" % us(np.diff(s[:, 0])))
print("causal layer       %.2f us" % us(s[:, 1] - s[:, 0]))
print("residual stack     %.2f us  (%.3f us/layer)" % (us(s[:, 2] - s[:, 1]), us(s[:, 2] - s[:, 1]) / len(dil)))
print("post (skip wait, conv1d_1, conv1d_2, sampler) %.2f us" % us(s[1:, 0] - s[:-1, 2]))
lay = np.diff(np.concatenate([s[:, 1:2], s[:, 8:8 + len(dil)]], axis=1), axis=1)
print("per-layer us:", " ".join("%.2f" % (v / mhz) for v in lay.mean(axis=0)))
f = s[:, 72:77]
print("layer 5 fine: wait_ready %.3f | tiles+conv %.3f | bias/act/z publish %.3f | dense+cdone %.3f us" % (
    us(f[:, 1] - f[:, 0]), us(f[:, 2] - f[:, 1]), us(f[:, 3] - f[:, 2]), us(f[:, 4] - f[:, 3])))

ld = s[:, 40:44]
print("loader0, layer-6 item: issue %.3f | landed after %.3f | acc0+publish %.3f us ; staged %.2f us before the chain reaches the layer" % (
    us(ld[:, 1] - ld[:, 0]), us(ld[:, 2] - ld[:, 1]), us(ld[:, 3] - ld[:, 2]), us(s[:, 8 + 5] - ld[:, 3])))
wk = s[:, 44:52]
print("worker0: skip pass %.2f (ends %.2f us after chain stack end) | gather h1 %.2f | conv1d_1 %.2f | gather h2 %.2f | conv1d_2 %.2f | wait partials %.2f | sampler %.2f us" % (
    us(wk[:, 1] - wk[:, 0]), us(wk[:, 1] - s[:, 2]), us(wk[:, 2] - wk[:, 1]), us(wk[:, 3] - wk[:, 2]), us(wk[:, 4] - wk[:, 3]),
    us(wk[:, 5] - wk[:, 4]), us(wk[:, 6] - wk[:, 5]), us(wk[:, 7] - wk[:, 6])))
hh = s[:, 52:57]
if hh[:, 0].any():
    print("helper(0,0) wave 0: h1 published -> seen %.2f | two chunk dots %.2f | wait all partials %.2f | ordered sum %.2f | conv1d_2 partials + publish %.2f | -> table complete in the sampler wave %.2f us" % (
        us(hh[:, 0] - wk[:, 1]), us(hh[:, 1] - hh[:, 0]), us(hh[:, 2] - hh[:, 1]), us(hh[:, 3] - hh[:, 2]), us(hh[:, 4] - hh[:, 3]), us(wk[:, 4] - hh[:, 4])))
if args.onehot:
    oh = s[:, 58:64]
    print("one-hot sampler: logits ready -> max+exp64 %.2f | f64 sum %.2f | log p / T %.2f | logaddexp chain + rescale %.2f | f64 cumsum %.2f | search+publish %.2f us" % (
        us(oh[:, 1] - oh[:, 0]), us(oh[:, 2] - oh[:, 1]), us(oh[:, 3] - oh[:, 2]), us(oh[:, 4] - oh[:, 3]), us(oh[:, 5] - oh[:, 4]), us(wk[:, 7] - oh[:, 5])))
