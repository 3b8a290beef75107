#!/usr/bin/env python
"""Anatomy of one Tacotron decoder step (tc_decoder_g_kernel, workgroup 0 of utterance 0) at BASELINE configs[2] (B = 32, 101 tokens,
200 steps): s_memtime stamps of the instrumented build, calibrated against the launch's HIP-event time; prints per-stage
microseconds, what every exchange costs over the measured one-way hop, and the latency floor that follows from the pieces
(the `tacotron.roofline.decoder` object of the bench line quotes this file: profiles/r04_tacotron_decoder_phase_profile.txt)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import twvk_amd
from twvk_amd.tacotron import Tacotron
hp = twvk_amd.default_hparams()
m = Tacotron(hp, num_speakers=2)
src = open(os.path.join(ROOT, "scripts", "tacotron_bench.py")).read()
ns = {}
exec(src[src.index("def random_tensors"):src.index("ap = argparse")], {"np": np}, ns)
m.load_weights(ns["random_tensors"](m.specs))
LOCAL = int(sys.argv[1]) if len(sys.argv) > 1 else -1       # optional argument: decoder_local 0 / 1
if LOCAL >= 0: m.set_option("decoder_local", LOCAL)
m.set_option("decoder_groups", 8)                            # the split kernel (round 6: the library's own choice at this batch is tc_decoder_x_kernel)
rng = np.random.RandomState(1)
N, T = 32, 101
tok = rng.randint(2, 80, (N, T)).astype(np.int32); tok[:, -1] = 1
ln = np.full(N, T, np.int32); spk = (np.arange(N) % 2).astype(np.int32)
ITERS = hp.max_iters
# production build: decoder time = pass with 200 steps minus pass with ... simpler: time the whole pass, then the instrumented one
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
m.infer(tok, ln, spk, want_linear=False); torch.cuda.synchronize()
e0.record(); m.infer(tok, ln, spk, want_linear=False); e1.record(); torch.cuda.synchronize()
plain_ms = e0.elapsed_time(e1)
prof = torch.zeros(ITERS * 64, dtype=torch.int64, device="cuda")
m._L.twv_tacotron_set_profile_buffer(m._h, C.c_void_p(prof.data_ptr()))
m.infer(tok, ln, spk, want_linear=False); torch.cuda.synchronize()
e0.record(); m.infer(tok, ln, spk, want_linear=False); e1.record(); torch.cuda.synchronize()
inst_ms = e0.elapsed_time(e1)
p = prof.cpu().numpy().reshape(ITERS, 64).astype(np.float64)
steps = np.diff(p[:, 0])                                     # stage 0 start to stage 0 start
span_ticks = p[-1, 53] - p[0, 0]
# the decoder kernel is most of the mel-only pass; calibrate the tick on the in-kernel span vs (instrumented pass - non-decoder part).
# The non-decoder part (embedding, prenet, encoder CBHG, keys) is measured by a 1-step decode.
hp1 = twvk_amd.default_hparams(); hp1.max_iters = 1
m1 = Tacotron(hp1, num_speakers=2); m1.load_weights(ns["random_tensors"](m1.specs))
if LOCAL >= 0: m1.set_option("decoder_local", LOCAL)
m1.set_option("decoder_groups", 8)
m1.infer(tok, ln, spk, want_linear=False); torch.cuda.synchronize()
e0.record(); m1.infer(tok, ln, spk, want_linear=False); e1.record(); torch.cuda.synchronize()
front_ms = e0.elapsed_time(e1)
dec_plain_us = (plain_ms - front_ms) * 1e3 / (ITERS - 1)
dec_inst_us = (inst_ms - front_ms) * 1e3 / (ITERS - 1)
tick = steps[20:180].mean() / dec_inst_us                    # ticks per microsecond
us = lambda a: float(np.mean(a)) / tick
P = p[20:180]
SPLIT_ALL = LOCAL != 0            # the library's default: prenet and query layer are split when the exchanges are L2-local
tag = "" if SPLIT_ALL else " (redundant)"
names = ["prenet dense_1" + tag, "prenet dense_2" + tag, "attention GRU gates", "attention GRU candidate", "query layer" + tag,
         "concat projection", "res GRU 1 gates", "res GRU 1 candidate", "res GRU 2 gates", "res GRU 2 candidate", "output projection"]
split = [int(SPLIT_ALL), int(SPLIT_ALL), 1, 1, int(SPLIT_ALL), 1, 1, 1, 1, 1, 1]
# one-way granule exchange measured in isolation: 0.27 us inside one XCD's L2 (plain store, sc1 load: the hand-offs of twv_wavenet_xcd.hip),
# 0.39 us between workgroups on different XCDs (scripts/ubench/tile_latency.hip: ~930 cycles)
HOP = 0.39 if LOCAL == 0 else 0.27
print("decoder step: %.2f us (production build, HIP events, (200-step pass - 1-step pass) / 199); instrumented build %.2f us; %.1f ticks/us" % (dec_plain_us, dec_inst_us, tick))
print("%-30s %7s %7s %9s %9s %7s" % ("stage", "dots", "combine", "exchange", "post", "total"))
tot_dots = tot_comb = tot_exch = tot_post = 0.0
n_exch = 0
for st, nme in enumerate(names):
    s0, s1, s2, s3 = P[:, 4 * st], P[:, 4 * st + 1], P[:, 4 * st + 2], P[:, 4 * st + 3]
    nxt = P[:, 4 * (st + 1)] if st + 1 < len(names) else P[:, 53]
    if st == 4:
        nxt = s3                                             # the attention block follows the query layer; listed separately
    dots, comb, exch, post = us(s1 - s0), us(s2 - s1), us(s3 - s2), us(nxt - s3)
    print("%-30s %7.2f %7.2f %9.2f %9.2f %7.2f" % (nme, dots, comb, exch, post, dots + comb + exch + post))
    tot_dots += dots; tot_comb += comb; tot_post += post
    if split[st]:
        tot_exch += exch; n_exch += 1
    else:
        tot_post += exch                                     # barrier only
q3 = P[:, 4 * 4 + 3]
sc, pg, rec, cpart, cg, cend = us(P[:, 48] - q3), us(P[:, 49] - P[:, 48]), us(P[:, 50] - P[:, 49]), us(P[:, 51] - P[:, 50]), us(P[:, 52] - P[:, 51]), us(P[:, 20] - P[:, 52])
print("attention: score chunk dots %.2f | p exchange %.2f | monotonic recurrence (wave 0) %.2f | context partial dots %.2f | context exchange %.2f | concat %.2f" % (sc, pg, rec, cpart, cg, cend))
tot_exch += pg + cg; n_exch += 2
att = sc + rec + cpart + cend
step_sum = tot_dots + tot_comb + tot_exch + tot_post + att
floor = tot_dots + tot_comb + tot_post + att + n_exch * HOP
print("sum of the pieces %.2f us (stamped step %.2f): tile dots %.2f + chunk sums/activation/publish %.2f + barriers/cell updates %.2f + attention compute %.2f + %d exchanges %.2f"
      % (step_sum, us(steps[20:180]), tot_dots, tot_comb, tot_post, att, n_exch, tot_exch))
print("exchanges: %.2f us each on average against a %.2f us one-way hop measured in isolation -- the rest is the next stage's weight tiles queued in front of the polls in the CU's memory pipeline, and skew between the 8 workgroups" % (tot_exch / n_exch, HOP))
print("latency floor of this decomposition = everything but the exchanges + %d x %.2f = %.2f us  (instrumented step %.2f: frac_of_floor %.3f)" % (n_exch, HOP, floor, dec_inst_us, floor / dec_inst_us))
print("JSON {\"us_per_step\": %.3f, \"us_per_step_instrumented\": %.3f, \"latency_floor_us\": %.3f, \"exchanges\": %d, \"exchange_us_mean\": %.3f, \"hop_us\": %.2f}" % (dec_plain_us, dec_inst_us, floor, n_exch, tot_exch / n_exch, HOP))
if P[:, 54].any():
    print("score phase detail (thread 0): query stage end -> indices %.2f | eight tanh terms + fma chain %.2f | lane exchanges + store %.2f | barrier %.2f" % (
        us(P[:, 54] - P[:, 4 * 4 + 3]), us(P[:, 55] - P[:, 54]), us(P[:, 56] - P[:, 55]), us(P[:, 48] - P[:, 56])))
