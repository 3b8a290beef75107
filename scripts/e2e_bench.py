#!/usr/bin/env python
"""BASELINE configs[4]: end-to-end text -> mel -> wave, 8 utterances sharded over the GPUs (one process per GPU, no collective
on the data path).  Default hparams: Tacotron 200 decoder steps -> 1000 mel frames -> 300 000 samples (12.5 s at 24 kHz) each.
    python scripts/e2e_bench.py [--frames 1000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/e2e_bench.py"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import twvk_amd
from twvk_amd.wavenet import WaveNetModel
from twvk_amd.tacotron import Synthesizer, tacotron_specs, Tacotron
from twvk_amd import weights as W
from twvk_amd.e2e import text_to_wave, shard_utterances
from twvk_amd.shard import max_over_ranks

ap = argparse.ArgumentParser()
ap.add_argument("--utterances", type=int, default=8); ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--tokens", type=int, default=100)
args = ap.parse_args()
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local); dev = "cuda:%d" % local
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device(dev))
hp = twvk_amd.default_hparams()
a, b = shard_utterances(args.utterances, world, rank)
B = b - a
rng = np.random.RandomState(3)
tokens = [list(rng.randint(2, 80, args.tokens)) + [1] for _ in range(args.utterances)][a:b]
spk = (np.arange(args.utterances) % 2).astype(np.int32)[a:b]
tm = Tacotron(hp, num_speakers=2, device=dev)
trng = np.random.RandomState(7); tt = {}
for n_, shp in tm.specs:
    if n_.endswith("batch_normalization"):
        c_ = shp[1]; tt[n_] = np.stack([np.ones(c_), np.zeros(c_), np.zeros(c_), np.ones(c_)]).astype(np.float32)
    elif n_.endswith("gates/bias"): tt[n_] = np.ones(shp, np.float32)
    elif n_.endswith("T/bias"): tt[n_] = -np.ones(shp, np.float32)
    elif n_.endswith("attention_g"): tt[n_] = np.array([np.sqrt(1.0 / hp.attention_size)], np.float32)
    elif n_.endswith("attention_score_bias"): tt[n_] = np.zeros(1, np.float32)
    else:
        fan = int(np.prod(shp[:-1])) if len(shp) > 1 else 1
        tt[n_] = (trng.randn(*shp) * (0.05 if len(shp) == 1 else min(0.5, 1.2 / np.sqrt(fan)))).astype(np.float32)
syn = Synthesizer(); syn.load(tt, num_speakers=2, hparams=hp, device=dev)
dil = [2 ** i for i in range(10)] * 3
voc = WaveNetModel(B, dil, hp.filter_width, hp.residual_channels, hp.dilation_channels, hp.skip_channels,
                   quantization_channels=hp.quantization_channels, out_channels=hp.out_channels, use_biases=hp.use_biases, scalar_input=True,
                   initial_filter_width=hp.initial_filter_width, global_condition_channels=hp.gc_channels, global_condition_cardinality=2,
                   local_condition_channels=hp.num_mels, upsample_factor=hp.upsample_factor, train_mode=False, device=dev)
voc.load_weights(W.random_tensors(voc.specs, seed=0, scale=0.05))
T = args.frames * voc.hop_size
lo, hi = np.float32(1e-5), np.float32(1 - 1e-5)
u = torch.from_numpy(np.random.RandomState(11 + rank).random_sample((B, T, 11)).astype(np.float32) * (hi - lo) + lo).to(dev)
text_to_wave(syn, voc, tokens, spk, u[:, :300 * 4], n_frames=4); torch.cuda.synchronize()      # warm-up
if world > 1: dist.barrier()
t0 = time.perf_counter()
out = text_to_wave(syn, voc, tokens, spk, u, n_frames=args.frames)
torch.cuda.synchronize()
if world > 1: dist.barrier()
dt = max_over_ranks(time.perf_counter() - t0, device=dev)
ok = bool(torch.isfinite(out["audio"]).all().item())
if rank == 0:
    print(json.dumps({"metric": "end-to-end text->mel->wave audio samples/sec", "value": args.utterances * T / dt, "unit": "samples/s",
                      "seconds": dt, "n_gpus": world, "realtime_factor_aggregate": args.utterances * T / dt / hp.sample_rate, "finite": ok,
                      "config": {"workload": "configs[4]: %d utterances x (%d tokens -> %d mel frames -> %d samples), %d per GPU, "
                                             "random-init weights" % (args.utterances, args.tokens + 1, args.frames, T, B)}}))
if world > 1: dist.destroy_process_group()
