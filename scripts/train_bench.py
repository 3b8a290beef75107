#!/usr/bin/env python
"""WaveNet teacher-forced training step at BASELINE configs[3]: MoL output, 30 layers, per-GPU batch 64 x 8000 samples
(cropped to 7800, datafeeder_wavenet.py:41-47); one process per GPU, gradient all-reduce over RCCL when WORLD_SIZE > 1:
    python scripts/train_bench.py --steps 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_bench.py"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import twvk_amd
from twvk_amd.wavenet import WaveNetModel
from twvk_amd.train import WaveNetTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--samples", type=int, default=8000)
ap.add_argument("--layers", type=int, default=30, help="30 as in configs[1]; hparams.py default is 50")
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
args = ap.parse_args()
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
hp = twvk_amd.default_hparams()
hp.dilations = ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 8)[:args.layers]
net = WaveNetModel(batch_size=args.batch, dilations=hp.dilations, filter_width=hp.filter_width, residual_channels=hp.residual_channels,
                   dilation_channels=hp.dilation_channels, skip_channels=hp.skip_channels, quantization_channels=hp.quantization_channels,
                   out_channels=hp.out_channels, use_biases=hp.use_biases, scalar_input=hp.scalar_input,
                   initial_filter_width=hp.initial_filter_width, global_condition_channels=hp.gc_channels,
                   global_condition_cardinality=2, local_condition_channels=hp.num_mels, upsample_factor=hp.upsample_factor,
                   train_mode=True, device="cuda:%d" % local)
tr = WaveNetTrainer(net, hp, sample_size=args.samples)
tr.init_weights(seed=0)                                     # identical on every rank
rng = np.random.RandomState(100 + rank)
T = tr.sample_size
audio = torch.from_numpy(((rng.rand(args.batch, T) - 0.5)).astype(np.float32)).cuda()
lc = torch.from_numpy((rng.randn(args.batch, T // net.hop_size, hp.num_mels) * 0.5).astype(np.float32)).cuda()
gc = torch.from_numpy(rng.randint(0, 2, args.batch).astype(np.int32)).cuda()
losses = []
for _ in range(args.warmup):
    losses.append(float(tr.step(audio, lc, gc).item()))
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = tr.step(audio, lc, gc)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = time.perf_counter() - t0
losses.append(float(loss.item()))
from twvk_amd.shard import max_over_ranks
dt = max_over_ranks(dt, device="cuda") / args.steps
if rank == 0:
    print(json.dumps({"metric": "WaveNet training samples/sec (teacher-forced step, MoL)", "value": world * args.batch * T / dt,
                      "unit": "audio samples/s", "steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "n_gpus": world, "scaling": "weak",
                      "dtype": "f32", "config": {"workload": "configs[3]: train_vocoder.py step, %d layers, per-GPU batch %d x %d samples, "
                                                 "out_w %d, %d parameters" % (len(hp.dilations), args.batch, T, tr.output_width, tr.n_params)},
                      "losses": losses}))
if world > 1:
    dist.destroy_process_group()
