#!/usr/bin/env python
"""In-kernel phase timing of the XCD-per-stream generation kernel (tuning aid): s_memtime stamps of stream 0's workgroups
(all on one XCD, one clock), averaged over the steps; prints the anatomy of a generation step in microseconds."""
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
ap.add_argument("--layers", type=int, default=30)
ap.add_argument("--onehot", action="store_true", help="the one-hot mu-law-256 model (scalar_input False) instead of the MoL model")
ap.add_argument("--ghz", type=float, default=0.0, help="s_memtime ticks per ns (0 = calibrate from the event time of the launch)")
args = ap.parse_args()
dil = ([2 ** i for i in range(10)] * 5)[:args.layers]
B, T = args.batch, args.steps
m = WaveNetModel(B, dil, 2, 32, 32, 512, out_channels=30, use_biases=True, scalar_input=not args.onehot, initial_filter_width=32,
                 global_condition_channels=32, global_condition_cardinality=2, local_condition_channels=80,
                 upsample_factor=[5, 5, 12], train_mode=False)
m.load_weights(W.random_tensors(m.specs, 0, 0.05))
assert m.fused_conditioning(), "the XCD-per-stream kernel does not serve this configuration"
rng = np.random.RandomState(0)
mel = rng.uniform(-4, 4, (B, (T + 299) // 300, 80)).astype(np.float32)
U = m.create_upsample(mel)
u = torch.from_numpy(rng.random_sample((B, T))).cuda() if args.onehot else torch.from_numpy(rng.uniform(1e-5, 1 - 1e-5, (B, T, 11)).astype(np.float32)).cuda()
NP = T
prof = torch.zeros((NP, 64), dtype=torch.int64, device="cuda")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
gc, fi = np.zeros(B, np.int32), (np.full(B, 128, np.int32) if args.onehot else np.zeros(B, np.float32))
m.generate(U, gc, fi, u)   # warm
m.queue_initializer()
e0.record(); m.generate(U, gc, fi, u, check=False); e1.record(); torch.cuda.synchronize()
plain = e0.elapsed_time(e1) * 1e3 / T
_lib.check(m._L.twv_wavenet_set_profile_buffer(m._h, C.c_void_p(prof.data_ptr()), NP))
m.queue_initializer()
e0.record(); m.generate(U, gc, fi, u, check=True); e1.record(); torch.cuda.synchronize()
inst = e0.elapsed_time(e1) * 1e3 / T
p = prof.cpu().numpy().astype(np.int64)[200:]            # steady state
step = np.diff(p[:, 0]).mean()
tick = args.ghz if args.ghz else step / (inst * 1e3)      # ticks per ns
us = lambda a: a.mean() / tick / 1e3
print("production build %.2f us/step, instrumented build %.2f us/step (events); %.3f ticks/ns" % (plain, inst, tick))
print("step period (wave 7 head to head)     %.2f us" % us(np.diff(p[:, 0])))
print("causal layer (queue shift, dot, box)  %.2f" % us(p[:, 1] - p[:, 0]))
nw = 8
recv = [p[:, 2 + w] for w in range(nw)]; done = [p[:, 10 + w] for w in range(nw)]; pa = [p[:, 26 + w] for w in range(nw)]
print("hand-off causal -> wave 0             %.2f" % us(recv[0] - p[:, 1]))
for w in range(nw):
    if done[w].any():
        print("wave %d: layers %.2f us%s" % (w, us(done[w] - recv[w]), ("   hand-off to wave %d %.2f" % (w + 1, us(recv[w + 1] - done[w]))) if w + 1 < nw and done[w + 1].any() else ""))
last = max(w for w in range(nw) if done[w].any())
print("residual stack total (wave 0 in -> last layer out)  %.2f" % us(done[last] - recv[0]))
# round 6 (KF = 1): the skip workgroups serve layers 0 .. NL-2 and publish the RAW running sum; the conv1 workgroups poll the last layer's z
# themselves, have its skip value ready when the sum arrives, and finish the sum + relu (stamp 22)
print("post: last layer out -> the skip workgroup's summing wave has seen ITS last z (layer NL-2) %.2f" % us(p[:, 20] - done[last]))
print("      its dot + ordered sum -> running sum of layers 0..NL-2 published  %.2f" % us(p[:, 21] - p[:, 20]))
print("      running sum -> conv1 workgroup has it, last skip value added, relu %.2f" % us(p[:, 22] - p[:, 21]))
print("      two chunk dots -> partials in LDS             %.2f" % us(p[:, 23] - p[:, 22]))
print("      wait for the other waves                      %.2f" % us(p[:, 24] - p[:, 23]))
if args.onehot:
    print("      ordered sum, relu -> h2 block published       %.2f" % us(p[:, 25] - p[:, 24]))
    print("      h2 -> wave 1 of conv1 workgroup 0 sees block 1 %.2f" % us(p[:, 34] - p[:, 25]))
    print("      conv1d_2 chunk dot -> partials in LDS         %.2f" % us(p[:, 35] - p[:, 34]))
    print("      wait for the other waves                      %.2f" % us(p[:, 36] - p[:, 35]))
    print("      ordered sum, bias -> 32 logits published      %.2f" % us(p[:, 37] - p[:, 36]))
    print("      logits -> sampler has all 256                 %.2f" % us(p[:, 18] - p[:, 37]))
    print("      sampler (f64 softmax, rescale, cdf, search)   %.2f" % us(p[:, 19] - p[:, 18]))
    print("        max + 4 float64 exps %.2f | denominator scan %.2f | e/sum, log p / T, max %.2f | log-sum-exp %.2f | exp + 4 cdf scans %.2f | search %.2f"
          % (us(p[:, 38] - p[:, 18]), us(p[:, 39] - p[:, 38]), us(p[:, 40] - p[:, 39]), us(p[:, 41] - p[:, 40]), us(p[:, 42] - p[:, 41]), us(p[:, 43] - p[:, 42])))
else:
    print("      ordered sum, relu, conv1d_2 chunk, publish    %.2f" % us(p[:, 25] - p[:, 24]))
    print("      partial table -> sampler has all of it        %.2f" % us(p[:, 18] - p[:, 25]))
    print("      sampler                                       %.2f" % us(p[:, 19] - p[:, 18]))
print("      sample -> next step's head                    %.2f" % us(p[1:, 0] - p[:-1, 19]))
print("post total (last layer out -> next head)            %.2f" % us(p[1:, 0] - done[last][:-1]))
for w in range(nw):
    if done[w].any():
        print("wave %d: conditioning granules ready %.2f us before its input arrives" % (w, us(recv[w] - pa[w])))
