#!/usr/bin/env python
"""Tacotron text->mel throughput at BASELINE configs[2]: B=32, 100 tokens + EOS, 200 decoder steps -> 1000 mel frames/utterance."""
import argparse, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import twvk_amd
from twvk_amd.tacotron import Tacotron, tacotron_specs

def random_tensors(specs, seed=0):
    rng = np.random.RandomState(seed); t = {}
    for n, shp in specs:
        if n.endswith("batch_normalization"):
            c = shp[1]; t[n] = np.stack([np.ones(c), np.zeros(c), np.zeros(c), np.ones(c)]).astype(np.float32)   # BN stats (0,1)
        elif n.endswith("gates/bias"): t[n] = np.ones(shp, np.float32)
        elif n.endswith("T/bias"): t[n] = -np.ones(shp, np.float32)
        elif n.endswith("attention_g"): t[n] = np.array([np.sqrt(1.0 / 256)], np.float32)
        elif n.endswith("attention_score_bias"): t[n] = np.zeros(1, np.float32)
        else:
            fan = int(np.prod(shp[:-1])) if len(shp) > 1 else 1
            t[n] = (rng.randn(*shp) * (0.05 if len(shp) == 1 else min(0.5, 1.2 / np.sqrt(fan)))).astype(np.float32)
    return t

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--no-post", action="store_true")
ap.add_argument("--no-linear", action="store_true"); ap.add_argument("--decoder-groups", type=int, default=0)
ap.add_argument("--decoder-local", type=int, default=-1, help="1: an utterance's decoder workgroups on one XCD, 0: spread over the XCDs")
ap.add_argument("--split-all", type=int, default=-1)
ap.add_argument("--nap-idle", type=int, default=-1); ap.add_argument("--nap-owner", type=int, default=-1)
ap.add_argument("--nap-round", type=int, default=-1); ap.add_argument("--nap-w0", type=int, default=-1)
ap.add_argument("--gemm-group", type=int, default=-1, help="0: one launch per GEMM and separate highway kernels (A/B against the grouped launches)"); args = ap.parse_args()
hp = twvk_amd.default_hparams()
m = Tacotron(hp, num_speakers=2)
m.load_weights(random_tensors(m.specs))
if args.decoder_groups: m.set_option("decoder_groups", args.decoder_groups)
if args.decoder_local >= 0: m.set_option("decoder_local", args.decoder_local)
if args.split_all >= 0: m.set_option("decoder_split_all", args.split_all)
if args.nap_idle >= 0: m.set_option("xdec_nap_idle", args.nap_idle)
if args.nap_owner >= 0: m.set_option("xdec_nap_owner", args.nap_owner)
if args.nap_round >= 0: m.set_option("xdec_nap_round", args.nap_round)
if args.nap_w0 >= 0: m.set_option("xdec_nap_w0", args.nap_w0)
if args.gemm_group >= 0: m.set_option("gemm_group", args.gemm_group)
rng = np.random.RandomState(1)
N, T = args.batch, 101
tok = rng.randint(2, 80, (N, T)).astype(np.int32); tok[:, -1] = 1
ln = np.full(N, T, np.int32); spk = (np.arange(N) % 2).astype(np.int32)
m.infer(tok, ln, spk, want_linear=not args.no_linear); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps): mel, lin, al = m.infer(tok, ln, spk, want_linear=not args.no_linear)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
frames = N * hp.max_iters * hp.reduction_factor
print(json.dumps({"metric": "Tacotron mel frames/sec", "value": frames / dt, "unit": "mel frames/s", "ms_per_pass": dt * 1e3,
                  "config": {"workload": "configs[2]: Tacotron text->mel, B=%d, T_in=%d, %d decoder steps, post-CBHG+linear %s" % (N, T, hp.max_iters, "off" if args.no_linear else "on")},
                  "finite": bool(torch.isfinite(mel).all().item())}))
