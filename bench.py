#!/usr/bin/env python
"""bench.py -- WaveNet autoregressive synthesis throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input: B=8 utterances x `--seconds` s of 24 kHz
audio (default 8 s = 192 000 samples each; BASELINE.json configs[1]): mel -> conditioning -> ONE persistent generation
launch (create_upsample, lc projections and the 192 000-step sample loop inside it) -> samples; inputs resident in HBM.
Prints ONE JSON line (rank 0).

`python bench.py --gpus N` starts its own N ranks (torch.distributed.run, 127.0.0.1) when it was not launched under one;
every rank drives one GPU with its own batch of utterances (weak scaling, no collective on the data path).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=8.0, help="audio seconds per utterance (8.0 = the BASELINE config)")
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU (8 = the BASELINE config; 1 = configs[4]'s one utterance per GPU)")
    ap.add_argument("--xcd", type=int, default=-1, help="0 = force the generic generation kernel (default: XCD-per-stream kernel where it qualifies)")
    ap.add_argument("--groups", type=int, default=-1, help="generic kernel: workgroups per stream (-1 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the secondary streams sweep (B = 8, 16, 32 on one GPU)")
    ap.add_argument("--no-tacotron", action="store_true", help="skip the secondary Tacotron mel-frames/s measurement")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement (configs[3], RCCL all-reduce at N>1)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="host time per CPU-baseline leg")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise the launch / rendezvous / max-over-ranks path (gloo)")
    ap.add_argument("--e2e", action="store_true", help="configs[4] instead: end-to-end text -> mel -> wave, 8 utterances sharded over the GPUs (scripts/e2e_bench.py)")
    ap.add_argument("--e2e-frames", type=int, default=1000, help="mel frames per utterance handed to the vocoder with --e2e (1000 = the full 200-step decode)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` outside a launcher: become N ranks (one per GPU) and hand back their exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def traffic_per_step(kernel, key):
    """HBM bytes per generation step from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    scripts/pmc_to_traffic.py), valid only for the generation-kernel sources it was measured on (_lib.generation_hash())."""
    try:
        import twvk_amd
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        e = t.get(twvk_amd._lib.generation_hash(), {}).get(kernel, {}).get(key)
        return None if e is None else float(e["fetch_bytes_per_step"]) + float(e["write_bytes_per_step"])
    except Exception:
        return None


def tacotron_traffic():
    """HBM bytes of the matrix-core kernels of one configs[2] pass, from the committed PMC passes of THIS build of the Tacotron kernels
    (profiles/traffic.json, key "tacotron:" + _lib.tacotron_hash(); scripts/pmc_to_tacotron_traffic.py); None for any other code"""
    try:
        import twvk_amd
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        e = t.get("tacotron:" + twvk_amd._lib.tacotron_hash(), {}).get("B32_T101")
        return None if e is None else float(e["gemm_fetch_bytes_per_pass"]) + float(e["gemm_write_bytes_per_pass"])
    except Exception:
        return None


def tacotron_decoder_traffic(key="B32_T101"):
    """fabric bytes of the decoder kernel of one pass (the same committed PMC passes; `key`: B32_T101 = the default (XCD-resident) decoder at
    the bench batch, B32_T101_split = the split decoder forced there (decoder_groups = 8), B16_T101 = batch 16)"""
    try:
        import twvk_amd
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        e = t.get("tacotron:" + twvk_amd._lib.tacotron_hash(), {}).get(key)
        if e is None or "decoder_fetch_bytes_per_pass" not in e:
            return None
        return {"kernel": e.get("decoder_kernel"), "bytes_per_pass": float(e["decoder_fetch_bytes_per_pass"]) + float(e["decoder_write_bytes_per_pass"])}
    except Exception:
        return None


def train_traffic():
    """HBM bytes of one configs[3] training step (all kernels), from the committed FETCH_SIZE / WRITE_SIZE passes of THIS build of the
    training kernels (profiles/traffic.json, key "train:" + _lib.train_hash(); scripts/train_traffic.sh); None for any other code"""
    try:
        import twvk_amd
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        e = t.get("train:" + twvk_amd._lib.train_hash(), {}).get("B64_T7800")
        return None if e is None else float(e["step_fetch_bytes"]) + float(e["step_write_bytes"])
    except Exception:
        return None


# ---- pieces of the result line that the measured run and `--dry-run` (the N-rank skeleton on CPU, tests/test_cpu.py) share, so that the
# schema the driver parses at N = 2, 4, 8 is exercised without a GPU
def headline_fields(value, n_ok, steps, warmup, dt, workload, B, T, kernel_label):
    return {"metric": "WaveNet autoregressive audio samples/sec at 24 kHz, batch=8",
            "value": value, "unit": "samples/s", "n_gpus": n_ok, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": B, "samples_per_utterance": T,
                       "sharding": "utterances, one batch of %d per GPU, no collective" % B, "kernel": kernel_label}}


def train_collective(world, n_params):
    """what the training step exchanges: one all-reduce per step; backend "nccl" of torch.distributed IS RCCL on ROCm"""
    return "all-reduce(sum) of one flat f32 gradient buffer, %d elements, RCCL" % n_params if world > 1 else "none (1 GPU)"


def checked_fields(ncheck, B, n_matched, world):
    return {"checked_against_oracle": "first %d samples of all %d streams of the last timed pass: bit-identical on %d of %d ranks" % (ncheck, B, n_matched, world),
            "checked_ranks": n_matched}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    assert world == args.gpus, "launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch

    if args.e2e:
        # BASELINE configs[4] (one line of its own): the same launcher path, scripts/e2e_bench.py as the rank program
        import runpy
        sys.argv = [os.path.join(ROOT, "scripts", "e2e_bench.py"), "--frames", str(args.e2e_frames)]
        runpy.run_path(sys.argv[0], run_name="__main__")
        return

    if args.dry_run:
        # the multi-rank skeleton without a GPU: rendezvous on 127.0.0.1, barrier-bracketed timed region, MAX over ranks, rank-0 report
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.05 * (rank + 1))
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        produced = torch.tensor([1.0, dt], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(produced[:1], op=dist.ReduceOp.SUM)
            dist.all_reduce(produced[1:], op=dist.ReduceOp.MAX)
        if rank == 0:
            # the line of the measured run, same keys through the same helpers, with the skeleton's numbers: every rank "generated" its
            # batch and "matched" the checker; the timed region is the slowest rank's
            n_ok, dt = int(produced[0].item()), float(produced[1].item())
            B, T = args.batch, int(round(args.seconds * 24000))
            res = headline_fields(n_ok * B * T * args.steps / dt, n_ok, args.steps, args.warmup, dt, "dry run (no GPU work)", B, T, "none (dry run)")
            res["roofline"] = {"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None}
            res["cpu_baseline"] = None                      # (rank 0 at N = 1 only, and not in a dry run)
            res.update(checked_fields(0, B, n_ok, world))
            res["train"] = {"n_gpus": world, "scaling": "weak", "collective": train_collective(world, 1157578)}
            res.update({"dry_run": True, "world": world, "max_seconds": dt})
            print(json.dumps(res))
        if dist is not None:
            dist.destroy_process_group()
        return

    import ctypes as C
    import twvk_amd
    from twvk_amd import _lib
    from twvk_amd.wavenet import WaveNetModel
    from twvk_amd import weights as W

    assert torch.cuda.device_count() > local_rank, "rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(dev))

    hp = twvk_amd.default_hparams()
    dil = [2 ** i for i in range(10)] * 3            # BASELINE configs[1]: 30 dilated residual layers
    Tm = int(round(args.seconds * hp.sample_rate / hp.hop_size))
    T = Tm * hp.hop_size
    tensors = None

    def make_vocoder(B, dil=dil, own_weights=False):
        nonlocal tensors
        m = WaveNetModel(B, dil, hp.filter_width, hp.residual_channels, hp.dilation_channels, hp.skip_channels,
                         quantization_channels=hp.quantization_channels, out_channels=hp.out_channels, use_biases=hp.use_biases,
                         scalar_input=True, initial_filter_width=hp.initial_filter_width,
                         global_condition_channels=hp.gc_channels, global_condition_cardinality=2,
                         local_condition_channels=hp.num_mels, upsample_factor=hp.upsample_factor, train_mode=False, device=dev)
        if args.xcd >= 0:
            m.set_option("xcd", args.xcd)
        if args.groups >= 0:
            m.set_option("groups", args.groups)
        if own_weights:
            m.load_weights(W.random_tensors(m.specs, seed=0, scale=0.05))
            return m
        if tensors is None:
            tensors = W.random_tensors(m.specs, seed=0, scale=0.05)
        m.load_weights(tensors)
        return m

    def make_inputs(B, T_, seed):
        rng = np.random.RandomState(seed)
        mel = torch.from_numpy(rng.uniform(-4, 4, (B, (T_ + hp.hop_size - 1) // hp.hop_size, hp.num_mels)).astype(np.float32)).to(dev)
        gc = (np.arange(B) % 2).astype(np.int32)
        seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
        lo, hi = np.float32(1e-5), np.float32(1 - 1e-5)
        u = torch.from_numpy((rng.random_sample((B, T_, 11)).astype(np.float32) * (hi - lo) + lo)).to(dev)
        return mel, gc, seed_in, u

    def one_pass(m, B, T_, mel, gc, seed_in, u):
        """the whole hot path for one batch; the events bracket the generation kernel on ITS stream"""
        m.queue_initializer()
        U = m.create_upsample(mel)                   # fused path: only wraps the mel; generic path: the stand-alone upsampling kernel
        cond = m._condition(U, gc, T_)
        fi = torch.as_tensor(seed_in, device=dev)
        out = torch.empty((B, T_), dtype=torch.float32, device=dev)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(m._L.twv_wavenet_generate(m._h, C.c_void_p(m._packed.data_ptr()), C.c_void_p(m._state.data_ptr()),
                                             C.c_void_p(cond.data_ptr()), C.c_void_p(fi.data_ptr()), C.c_void_p(u.data_ptr()), 1.0,
                                             B, T_, C.c_void_p(out.data_ptr()), C.c_void_p(m._status.data_ptr()), None, 0,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        e1.record()
        return out, (e0, e1), cond

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    B = args.batch
    m = make_vocoder(B)
    fused = m.fused_conditioning()
    mel, gc, seed_in, u = make_inputs(B, T, 1 + rank)
    for _ in range(args.warmup):
        out, _ev, _c = one_pass(m, B, T, mel, gc, seed_in, u)
    sync_all()
    t0 = time.perf_counter()
    evs = []
    for _ in range(args.steps):
        out, ev, _c = one_pass(m, B, T, mel, gc, seed_in, u)
        evs.append(ev)
    sync_all()
    dt = time.perf_counter() - t0
    _lib.check(m._L.twv_wavenet_status(C.c_void_p(m._status.data_ptr()), None))
    samples = out.cpu().numpy()
    ok = bool(np.isfinite(samples).all() and np.abs(samples).max() <= 1.0)
    # the checker on THIS rank's timed output (after the timed region): the first 600 samples of every stream, bit for bit
    ncheck = min(T, 600)
    matched = 0.0
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        d_chk = O.make_dims(dil)
        blob_chk = O.blob_from_tensors(d_chk, tensors)
        Uo = O.upsample(d_chk, blob_chk, mel.cpu().numpy()[:, :(ncheck + hp.hop_size - 1) // hp.hop_size])[:, :ncheck]
        O.set_threads(min(B, O.set_threads(1)))
        want = O.generate_mol(d_chk, blob_chk, O.State(d_chk, B), Uo, gc, seed_in, u[:, :ncheck].cpu().numpy())
        O.set_threads(1)
        matched = 1.0 if np.array_equal(samples[:, :ncheck], want) else 0.0
    produced = torch.tensor([1.0 if ok else 0.0, dt, matched], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(produced[:1], op=dist.ReduceOp.SUM)       # ranks that produced samples: what n_gpus reports
        dist.all_reduce(produced[1:2], op=dist.ReduceOp.MAX)
        dist.all_reduce(produced[2:], op=dist.ReduceOp.SUM)       # ranks whose output matched the checker
    n_ok, dt, n_matched = int(produced[0].item()), float(produced[1].item()), int(produced[2].item())
    assert ok, "rank %d: samples are not finite / not in [-1, 1]" % rank
    assert args.no_cpu_baseline or matched == 1.0, "rank %d: timed output differs from the CPU checker in the first %d samples" % (rank, ncheck)
    gen_ms = [e0.elapsed_time(e1) for e0, e1 in evs]

    # secondary: configs[3] teacher-forced training step, per-GPU batch 64 x 8000 (cropped to 7800) samples, data-parallel with a
    # gradient all-reduce over RCCL when N > 1 -- every rank takes part, so it runs before the rank-0 report is assembled
    train_res = None
    if not args.no_train:
        # set-up first, then ONE agreement (MAX of a failure flag over the ranks) before anything enters the gradient all-reduce: a
        # rank that could not build its trainer must not leave the others waiting inside trn.step's collective
        trn, train_err = None, None
        try:
            from twvk_amd.train import WaveNetTrainer
            tnet = WaveNetModel(64, dil, hp.filter_width, hp.residual_channels, hp.dilation_channels, hp.skip_channels,
                                quantization_channels=hp.quantization_channels, out_channels=hp.out_channels, use_biases=hp.use_biases,
                                scalar_input=True, initial_filter_width=hp.initial_filter_width, global_condition_channels=hp.gc_channels,
                                global_condition_cardinality=2, local_condition_channels=hp.num_mels, upsample_factor=hp.upsample_factor,
                                train_mode=True, device=dev)
            trn = WaveNetTrainer(tnet, hp, sample_size=8000)
            trn.init_weights(seed=0)                                   # identical replicas
            TT_ = trn.sample_size
            trng = np.random.RandomState(100 + rank)
            taudio = torch.from_numpy((trng.rand(64, TT_) - 0.5).astype(np.float32)).to(dev)
            tlc = torch.from_numpy((trng.randn(64, TT_ // tnet.hop_size, hp.num_mels) * 0.5).astype(np.float32)).to(dev)
            tgc = torch.from_numpy(trng.randint(0, 2, 64).astype(np.int32)).to(dev)
        except Exception as e:
            train_err = e
        if dist is not None:
            tf_ = torch.tensor([0.0 if train_err is None else 1.0], dtype=torch.float32, device=dev)
            dist.all_reduce(tf_, op=dist.ReduceOp.MAX)
            if float(tf_.item()) > 0.0 and train_err is None:
                train_err = RuntimeError("another rank could not set the training step up")
        try:
            if train_err is not None:
                raise train_err
            l0 = float(trn.step(taudio, tlc, tgc).item())
            sync_all()
            q0 = time.perf_counter()
            for _ in range(5):
                tl = trn.step(taudio, tlc, tgc)
            sync_all()
            qdt = (time.perf_counter() - q0) / 5
            if dist is not None:
                tq = torch.tensor([qdt], dtype=torch.float64, device=dev)
                dist.all_reduce(tq, op=dist.ReduceOp.MAX)
                qdt = float(tq.item())
            # executed matrix-core work of one step (forward + two backward contractions per GEMM; the lc projection is NOT in it: it
            # runs at frame rate as 8 VALU fmas per row): per layer (2 taps x 32x64 + dense 32x32) MACs on the rows the
            # layer kernels walk and the skip 1x1 (32x512) on the B*out_w output rows; conv1d_1 512x512 and conv1d_2 512x30 on the output rows
            # Rows per layer: the layer kernels walk a batch entry from the 32-row tile that holds the layer's receptive offset on (forward:
            # the output's offset off[l+1], the two backward kernels: the input's offset off[l]); the rows in front of it do not exist in the
            # reference's 'valid' convolutions and are not computed.
            rows_ow = 64 * trn.output_width
            off_, mac_layers = hp.initial_filter_width - 1, 0.0
            for d_ in dil:
                rows_f = 64 * ((TT_ - 1) - ((off_ + d_) // 32) * 32); rows_b = 64 * ((TT_ - 1) - (off_ // 32) * 32)
                mac_layers += (rows_f + 2.0 * rows_b) * (2 * 32 * 64 + 32 * 32)
                off_ += d_
            mac_post = len(dil) * rows_ow * 32 * 512 + rows_ow * (512 * 512 + 512 * 30)
            flop_step = 2.0 * (mac_layers + 3.0 * mac_post)
            ttraf_ = train_traffic() if (TT_ == 7800 and len(dil) == 30) else None
            train_res = {"metric": "WaveNet training audio samples/sec (teacher-forced step: MoL loss, backward, all-reduce, Adam, EMA)",
                         "roofline": {"bound": "mfma", "kernel": "whole step: tr_layer_{fwdc,bwd1,bwd2c}_kernel + the wide f32 GEMMs (rocBLAS MI16x16x4)",
                                      "achieved": flop_step / qdt / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flop_step / qdt / 1e12 / 157.3,
                                      "flop_per_step": flop_step, "traffic": ttraf_,
                                      "hbm": None if ttraf_ is None else {"bytes_per_step": ttraf_, "achieved": ttraf_ / qdt / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                                           "frac": ttraf_ / qdt / 1e9 / 8000.0,
                                                                           "note": "the step's second ruler: the fused layer kernels (37 % of it) run at 3.8-4.1 TB/s, the elementwise "
                                                                                   "passes at 5-6 TB/s; profiles/r06_train_traffic.txt"},
                                      "note": "executed f32 matrix-core FLOPs only (2 x 3 x forward MACs of the dense contractions; per-kernel durations and "
                                              "SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_VALU_MFMA_MOPS_F32 counters: profiles/r06_rocprofv3_mfma_summary_tacotron_train.txt, r06_rocprofv3_kernel_stats_train_c4.csv)"},
                         "value": world * 64 * TT_ / qdt, "unit": "audio samples/s", "steps_per_s": 1.0 / qdt, "ms_per_step": qdt * 1e3,
                         "n_gpus": world, "scaling": "weak", "dtype": "f32",
                         "collective": train_collective(world, trn.n_params),
                         "config": {"workload": "configs[3]: train_vocoder.py step, 30 layers, per-GPU batch 64 x %d samples, random-init weights" % TT_},
                         "loss_first": l0, "loss_last": float(tl.item())}
            del trn, tnet, taudio, tlc
        except Exception as e:       # the headline metric must still be reported
            train_res = {"error": repr(e)}

    if rank == 0:
        total_samples = n_ok * B * T * args.steps
        value = total_samples / dt
        # roofline: algorithmic HBM bytes per generation step for all B streams, weights streamed every step
        NL = len(dil)
        per_layer = 2 * (2 * 32 * 32 + 32) + 2 * 80 * 32 + (32 * 32 + 32) + (32 * 512 + 512)   # gc hoisted
        wfloats = NL * per_layer + (512 * 512 + 512) + (512 * 30 + 30) + 32 * 32
        bytes_per_step = wfloats * 4 + B * (80 + 1 + 1) * 4
        k_ms = float(np.mean(gen_ms))
        achieved = bytes_per_step * T / (k_ms * 1e-3) / 1e9
        kernel = m.kernel_name()                 # the library's own choice for this (model, batch, options): twv_wavenet_kernel_name
        us_step = k_ms * 1e3 / T
        # what binds this kernel is the sample-to-sample dependency chain, not HBM: the floor of that chain from the micro-benchmarks
        # (scripts/ubench/chain_contract_ubench.hip -> profiles/r05_chain_contract_ubench.txt: contract C7 (the product's since round 5), shape R =
        # 417 core clocks = 0.174 us per layer; hand-offs / post phase: scripts/xcd_phase_profile.py -> profiles/r06_xcd_phase_profile.txt).
        # Round 6 (KF = 1): the last layer's z goes straight to the conv1 workgroups, so the three-hop post path starts at the SECOND-TO-LAST
        # layer and runs next to the last layer: one layer time comes off the post phase's floor
        LAYER_US, HANDOFF_US, CAUSAL_US = 0.174, 0.075, 0.15
        POST_US = 3 * 0.26 + 1.08 - LAYER_US
        floor_us = NL * LAYER_US + 8 * HANDOFF_US + CAUSAL_US + POST_US
        # BASELINE.json north_star: >= 100x real time at batch 8 = 2.4 M samples/s = one generation step every 3.33 us.  Round 5 priced the
        # alternatives (profiles/r05_chain_contract_ubench.txt) and ADOPTED the cheapest one (C7 = AC-1b / AC-2 of DESIGN.md section 2: 562 ->
        # 504 clocks per layer in the product's shape, together with the stores taken off the sample path 9.48 -> 8.6 us per step): the
        # chain wave is ISSUE-bound (a layer is ~100 instructions at ~4.5 core clocks each on a lone wave; 48 of them are the fmas of the two
        # dot products, which no arithmetic contract removes), and 30 layers x 0.174 us (the arithmetic alone, registers only) is 5.2 us.
        target_us = 1e6 / (100.0 * hp.sample_rate / B)
        floor_r04_contract_us = NL * 0.192 + 8 * HANDOFF_US + CAUSAL_US + POST_US + LAYER_US
        macs_stream = NL * (2 * 32 * 64 + 32 * 32 + 32 * 512 + 80 * 64) + 32 * 32 + 512 * 512 + 512 * 30   # executed per stream and step (gc hoisted)
        flop_step = 2.0 * macs_stream * B
        tps = traffic_per_step(kernel, "B%d_NL%d" % (B, NL))
        res = headline_fields(value, n_ok, args.steps, args.warmup, dt,
                              "configs[1]: WaveNet autoregressive synth, 30 dilated residual layers (3x[1..512]), R=D=32, S=512, "
                              "MoL-30 output, gc+lc conditioning, 24 kHz, batch=%d x %.2f s (%d samples each) per GPU, random-init weights, "
                              "injected uniforms" % (B, T / hp.sample_rate, T), B, T,
                              kernel + (" (stream b on XCD b % 8, weights register-resident, create_upsample + lc projections fused into the launch)" if fused else ""))
        res.update({
            "target_100x_at_batch_8": {"reachable": False, "target_us_per_step": target_us, "us_per_step": us_step,
                                       "floor_us": floor_us, "floor_us_under_the_contract_of_rounds_1_to_4": floor_r04_contract_us,
                                       "contract": "AC-1b / AC-2 (round 5): the cheapest bit-reproducible contract of the seven priced (C7), adopted in oracle, "
                                                   "fixtures and all three generation kernels",
                                       "evidence": "profiles/r05_chain_contract_ubench.txt (contracts C0-C7 x shapes R/P/D/G/N/M, each bit-checked against its "
                                                   "canonical fmaf form); profiles/r05_ab_*.txt (interleaved A/B of every step); DESIGN.md sections 3, 11, 12",
                                       "note": "a 30-layer step is a dependent chain issued by ONE wave per layer: 30 x (32 + 16 fmas + activation) cannot be "
                                               "issued in 3.33 us under any contract priced; the per-GPU figure comes with more streams (streams_sweep: 100x real "
                                               "time is passed at batch 32, 270x at batch 64)"},
            "realtime_factor_aggregate": value / hp.sample_rate,
            "realtime_factor_per_stream": value / hp.sample_rate / (n_ok * B),
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0,
                         # HBM bytes per launch from rocprofv3 PMC (separate --pmc passes; FETCH_SIZE / WRITE_SIZE with the guide's gfx950
                         # corrections), per generated step, for THESE generation-kernel sources (profiles/traffic.json is keyed by _lib.generation_hash())
                         "traffic": None if tps is None else tps * T, "algorithmic_bytes_per_launch": bytes_per_step * T,
                         # SURVEY 8d's second convention, reported beside the streamed-weights one: with every weight resident on chip the
                         # algorithmic HBM bytes of a step are its I/O only (per stream 80 lc floats in, one sample in, one out)
                         "weights_on_chip": {"algorithmic_bytes_per_step": B * (80 + 1 + 1) * 4,
                                             "measured_bytes_per_step": tps,
                                             "measured_over_algorithmic": None if tps is None else tps / (B * (80 + 1 + 1) * 4),
                                             "note": "counter traffic / I-O-only bytes: what the register-resident kernel really moves "
                                                     "(uniforms in, progress / exchange words that spill from L2 included)"},
                         "kernel_ms": k_ms, "us_per_generation_step": us_step,
                         "latency_floor_us": floor_us, "frac_of_floor": floor_us / us_step,
                         "latency_floor_formula": "%d layers x 0.174 us (a layer's arithmetic alone on a lone wave, registers only: 32+16 fmas started from the "
                                                  "addends, Estrin rational with a software reciprocal; 417 core clocks) + 8 wave hand-offs x 0.075 + causal layer 0.15 + "
                                                  "post phase (3 L2 hops x 0.26 + skip 0.22 + chunk dots 0.28 + ordered sum/conv1d_2 0.27 + sampler 0.31 - one layer time: since round 6 "
                                                  "the last layer's skip 1x1 runs in the conv1 workgroups and the three-hop path starts behind the second-to-last layer); "
                                                  "measured pieces, profiles/r05_chain_contract_ubench.txt and profiles/r06_xcd_phase_profile.txt" % NL,
                         "fp32_flop_per_step": flop_step, "fp32_tflops": flop_step / (us_step * 1e-6) / 1e12,
                         "fp32_frac": flop_step / (us_step * 1e-6) / 1e12 / 157.3,
                         "note": "weights are register-/L2-resident: the sample loop is a dependent chain (latency), not a bandwidth stream; "
                                 "algorithmic bytes assume the weights were re-read from HBM every step (SURVEY.md 8d)"},
        })
        if not args.no_sweep and world == 1:
            # what more streams on the same GPU are worth: 1 s of audio per stream at B = 8 (XCD-per-stream kernel), 16 and 32 (generic kernel)
            sweep = []
            T1 = hp.sample_rate // hp.hop_size * hp.hop_size
            for Bs in (8, 16, 32, 48, 64, 72, 96):
                try:
                    ms = m if Bs == B else make_vocoder(Bs)
                    inp = make_inputs(Bs, T1, 50 + Bs)
                    one_pass(ms, Bs, T1, *inp)
                    torch.cuda.synchronize()
                    _o, (s0, s1), _c2 = one_pass(ms, Bs, T1, *inp)
                    torch.cuda.synchronize()
                    kms = s0.elapsed_time(s1)
                    _lib.check(ms._L.twv_wavenet_status(C.c_void_p(ms._status.data_ptr()), None))
                    sweep.append({"streams": Bs, "samples_per_s": Bs * T1 / (kms * 1e-3), "us_per_generation_step": kms * 1e3 / T1,
                                  "realtime_factor_aggregate": Bs * T1 / (kms * 1e-3) / hp.sample_rate,
                                  "realtime_factor_per_stream": T1 / (kms * 1e-3) / hp.sample_rate,
                                  "fp32_frac": 2.0 * macs_stream * Bs / (kms * 1e-3 / T1) / 1e12 / 157.3,
                                  "kernel": ms.kernel_name()})
                    if ms is not m:
                        del ms
                except Exception as e:
                    sweep.append({"streams": Bs, "error": repr(e)})
            res["streams_sweep"] = sweep
            # hparams.py's own default stack is 5 x [1..512] = 50 layers (BASELINE configs[1] quotes 30): the same pass at B = 8 x 1 s
            try:
                m50 = make_vocoder(B, dil=[2 ** i for i in range(10)] * 5, own_weights=True)
                inp = make_inputs(B, T1, 77)
                one_pass(m50, B, T1, *inp)
                torch.cuda.synchronize()
                _o, (s0, s1), _c2 = one_pass(m50, B, T1, *inp)
                torch.cuda.synchronize()
                kms = s0.elapsed_time(s1)
                _lib.check(m50._L.twv_wavenet_status(C.c_void_p(m50._status.data_ptr()), None))
                res["hparams_default_50_layers"] = {"streams": B, "samples_per_s": B * T1 / (kms * 1e-3), "us_per_generation_step": kms * 1e3 / T1,
                                                    "realtime_factor_per_stream": T1 / (kms * 1e-3) / hp.sample_rate,
                                                    "kernel": m50.kernel_name()}
                del m50
            except Exception as e:
                res["hparams_default_50_layers"] = {"error": repr(e)}
        if not args.no_sweep and world == 1:
            # the mu-law-256 variant of configs[1] (one-hot input, 256-way softmax output) -- the model north_star's integer
            # parity bar is stated on.  Round 4: on the XCD-per-stream kernel (wn_xcd_generate_kernel<.., ONEHOT>): causal layer as two
            # kernel rows, conv1d_2 split by output over the conv1 workgroups, generate.py:219-231's sampler lane-parallel on the head wave
            # (AC-5, founded on numpy itself: tests/test_cpu.py::test_categorical_sampler_contract_against_numpy).
            try:
                Tq = hp.sample_rate // 2 // hp.hop_size * hp.hop_size
                mq = WaveNetModel(B, dil, hp.filter_width, hp.residual_channels, hp.dilation_channels, hp.skip_channels,
                                  quantization_channels=256, out_channels=hp.out_channels, use_biases=hp.use_biases, scalar_input=False,
                                  initial_filter_width=hp.initial_filter_width, global_condition_channels=hp.gc_channels,
                                  global_condition_cardinality=2, local_condition_channels=hp.num_mels, upsample_factor=hp.upsample_factor,
                                  train_mode=False, device=dev)
                if args.xcd >= 0:
                    mq.set_option("xcd", args.xcd)
                tq = W.random_tensors(mq.specs, seed=0, scale=0.05)
                mq.load_weights(tq)
                rq = np.random.RandomState(91)
                melq_h = rq.uniform(-4, 4, (B, Tq // hp.hop_size, hp.num_mels)).astype(np.float32)
                uq_h = rq.random_sample((B, Tq))
                melq = torch.from_numpy(melq_h).to(dev)
                uq = torch.from_numpy(uq_h).to(dev)
                fq = rq.randint(256, size=B).astype(np.int32)
                fqd = torch.as_tensor(fq, device=dev)
                qfused = mq.fused_conditioning()

                def q_pass():
                    mq.queue_initializer()
                    Uq = mq.create_upsample(melq)
                    condq = mq._condition(Uq, gc, Tq)
                    oq_ = torch.empty((B, Tq), dtype=torch.int32, device=dev)
                    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
                    s0.record()
                    _lib.check(mq._L.twv_wavenet_generate(mq._h, C.c_void_p(mq._packed.data_ptr()), C.c_void_p(mq._state.data_ptr()),
                                                          C.c_void_p(condq.data_ptr()), C.c_void_p(fqd.data_ptr()), C.c_void_p(uq.data_ptr()), 1.0,
                                                          B, Tq, C.c_void_p(oq_.data_ptr()), C.c_void_p(mq._status.data_ptr()), None, 0,
                                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                    s1.record()
                    torch.cuda.synchronize()
                    _lib.check(mq._L.twv_wavenet_status(C.c_void_p(mq._status.data_ptr()), None))
                    return oq_, s0.elapsed_time(s1), condq
                q_pass()
                oq, qms, _cq = q_pass()
                q_us = qms * 1e3 / Tq
                # algorithmic bytes per step (SURVEY 8d's convention: every weight re-read per step): the layers as above, the one-hot
                # causal kernel (2 x 256 x 32), conv1d_1, the 512 x 256 conv1d_2; I/O per stream: 80 lc floats, class in, class out, one f64 draw
                q_wfloats = NL * per_layer + (512 * 512 + 512) + (512 * 256 + 256) + 2 * 256 * 32
                q_bytes = q_wfloats * 4 + B * (80 * 4 + 4 + 4 + 8)
                q_floor = NL * 0.174 + 8 * 0.075 + 0.18 + (4 * 0.26 + 0.22 + 0.28 + 0.12 + 0.17 + 0.12 + 1.55) - 0.174   # (- one layer time: KF = 1, see the headline's floor)
                q_tps = traffic_per_step("wn_xcd_generate_kernel_onehot", "B%d_NL%d" % (B, NL)) if qfused else None
                q_match = None
                if not args.no_cpu_baseline:
                    nq = 400
                    dq = O.make_dims(dil, scalar_input=False, Q=256)
                    blobq = O.blob_from_tensors(dq, tq)
                    Uo = O.upsample(dq, blobq, melq_h[:, :(nq + hp.hop_size - 1) // hp.hop_size])[:, :nq]
                    O.set_threads(min(B, O.set_threads(1)))
                    wantq = O.generate_mulaw(dq, blobq, O.State(dq, B), Uo, gc, fq, uq_h[:, :nq], 1.0)
                    O.set_threads(1)
                    q_match = bool(np.array_equal(oq[:, :nq].cpu().numpy(), wantq))
                    assert q_match, "mu-law-256 variant: timed output differs from the CPU checker"
                res["mulaw_256"] = {"streams": B, "samples_per_s": B * Tq / (qms * 1e-3), "us_per_generation_step": q_us,
                                    "realtime_factor_per_stream": Tq / (qms * 1e-3) / hp.sample_rate,
                                    "kernel": "wn_xcd_generate_kernel (one-hot instantiation)" if qfused else "wn_generate_kernel",
                                    "classes_drawn": int(torch.unique(oq).numel()), "dtype": "f32 network, f64 softmax/cdf, int32 class ids",
                                    "checked_against_oracle": None if q_match is None else "first 400 class ids of all %d streams: identical" % B,
                                    "roofline": {"bound": "hbm", "achieved": q_bytes / (q_us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                 "frac": q_bytes / (q_us * 1e-6) / 1e9 / 8000.0, "algorithmic_bytes_per_step": q_bytes,
                                                 "traffic": None if q_tps is None else q_tps * Tq, "measured_bytes_per_step": q_tps,
                                                 "kernel_ms": qms, "latency_floor_us": q_floor, "frac_of_floor": q_floor / q_us,
                                                 "latency_floor_formula": "%d layers x 0.174 us + 8 wave hand-offs x 0.075 + causal row load 0.18 + post phase: 4 L2 hops x 0.26 "
                                                                          "(z, h1, h2, logits) + skip 0.22 + conv1d_1 chunk dots 0.28 + ordered sum 0.12 + conv1d_2 chunk dot 0.17 + "
                                                                          "ordered sum 0.12 + sampler 1.55 (its ~850 VALU instructions at 4 cycles each, one wave, 2.2 GHz; measured 3.18) - one layer time (round 6: the last layer's skip 1x1 in the conv1 workgroups); "
                                                                          "pieces: profiles/r05_chain_contract_ubench.txt, profiles/r06_xcd_onehot_phase_profile.txt" % NL,
                                                 "note": "as for the headline kernel: weights are register-resident, the step is a dependent chain (latency), "
                                                         "algorithmic bytes assume every weight re-read per step (SURVEY.md 8d)"},
                                    "config": {"workload": "configs[1]'s stack with one-hot mu-law-256 input and a 256-way softmax output (model.py:223-227,243; "
                                                           "generate.py:219-231 with the uniform draw injected), batch=%d x %d steps" % (B, Tq)}}
                del mq
                # more streams on the same GPU (two and four per XCD), 0.25 s of audio per stream
                qs = []
                for Bs in (16, 32):
                    try:
                        mqs = WaveNetModel(Bs, dil, hp.filter_width, hp.residual_channels, hp.dilation_channels, hp.skip_channels,
                                           quantization_channels=256, out_channels=hp.out_channels, use_biases=hp.use_biases, scalar_input=False,
                                           initial_filter_width=hp.initial_filter_width, global_condition_channels=hp.gc_channels,
                                           global_condition_cardinality=2, local_condition_channels=hp.num_mels, upsample_factor=hp.upsample_factor,
                                           train_mode=False, device=dev)
                        if args.xcd >= 0:
                            mqs.set_option("xcd", args.xcd)
                        mqs.load_weights(tq)
                        Ts = hp.sample_rate // 4 // hp.hop_size * hp.hop_size
                        rs_ = np.random.RandomState(92 + Bs)
                        mels = torch.from_numpy(rs_.uniform(-4, 4, (Bs, Ts // hp.hop_size, hp.num_mels)).astype(np.float32)).to(dev)
                        us_ = torch.from_numpy(rs_.random_sample((Bs, Ts))).to(dev)
                        fs_ = rs_.randint(256, size=Bs).astype(np.int32)
                        gcs_ = (np.arange(Bs) % 2).astype(np.int32)
                        Us_ = mqs.create_upsample(mels)
                        mqs.generate(Us_, gcs_, fs_, us_)
                        mqs.queue_initializer()
                        torch.cuda.synchronize()
                        c0 = time.perf_counter()
                        mqs.generate(Us_, gcs_, fs_, us_)
                        torch.cuda.synchronize()
                        cdt = time.perf_counter() - c0
                        qs.append({"streams": Bs, "samples_per_s": Bs * Ts / cdt, "us_per_generation_step": cdt / Ts * 1e6,
                                   "realtime_factor_per_stream": Ts / cdt / hp.sample_rate})
                        del mqs
                    except Exception as e:
                        qs.append({"streams": Bs, "error": repr(e)})
                res["mulaw_256"]["streams_sweep"] = qs
            except Exception as e:
                res["mulaw_256"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            res.update(checked_fields(ncheck, B, n_matched, world))
        if not args.no_cpu_baseline and world == 1:
            d, blob = d_chk, blob_chk

            # the same workload on the host (rank 0, N = 1 only): 1 thread, then one stream per core on every core
            def cpu_leg(nstreams, threads):
                rng = np.random.RandomState(77)
                probe = 100
                gcs = (np.arange(nstreams) % 2).astype(np.int32)
                sd = rng.uniform(-1, 1, nstreams).astype(np.float32)
                uu = rng.uniform(1e-5, 1 - 1e-5, (nstreams, probe, 11)).astype(np.float32)
                Up = rng.uniform(-1, 1, (nstreams, probe, 80)).astype(np.float32)
                O.set_threads(threads)
                st = O.State(d, nstreams)
                c0 = time.perf_counter()
                O.generate_mol(d, blob, st, Up, gcs, sd, uu)
                per = (time.perf_counter() - c0) / probe
                n = int(max(probe, min(T, args.cpu_seconds / max(per, 1e-9))))
                uu = rng.uniform(1e-5, 1 - 1e-5, (nstreams, n, 11)).astype(np.float32)
                Up = rng.uniform(-1, 1, (nstreams, n, 80)).astype(np.float32)
                st = O.State(d, nstreams)
                c0 = time.perf_counter()
                O.generate_mol(d, blob, st, Up, gcs, sd, uu)
                cdt = time.perf_counter() - c0
                O.set_threads(1)
                return nstreams * n / cdt, n, cdt
            cores = O.set_threads(1)
            v1, n1, t1 = cpu_leg(B, 1)
            res["cpu_baseline"] = {"value": v1, "unit": "samples/s", "cores": 1, "kind": "port",
                                   "sample": "build's CPU restatement (oracle/, plain C, 1 thread) of the same model on the GPU box's host: "
                                             "B=%d x %d generation steps = %.1f s of CPU work; NOT the reference generate.py "
                                             "(TensorFlow is absent; parity unpinned)" % (B, n1, t1)}
            if cores > 1:
                vc, nc, tc = cpu_leg(cores, cores)
                quota, eff = "", cores
                try:
                    with open("/sys/fs/cgroup/cpu.max") as fh:
                        q = fh.read().split()
                    quota = "; cgroup cpu.max = " + " ".join(q)
                    if q[0] != "max":
                        eff = max(1, min(cores, int(float(q[0]) / float(q[1]))))      # the cores the cgroup actually grants
                except Exception:
                    pass
                res["cpu_baseline"]["all_cores"] = {"value": vc, "unit": "samples/s", "cores": eff, "threads": cores, "kind": "port",
                                                    "sample": "the same restatement, one stream per thread (OpenMP, %d threads on %d granted cores), %d streams x %d steps = %.1f s%s"
                                                              % (cores, eff, cores, nc, tc, quota)}
        if not args.no_tacotron and world == 1:
            # secondary half of BASELINE.json's metric: Tacotron mel frames/sec at configs[2] (B=32, 100 tokens + EOS, 200 decoder steps)
            try:
                from twvk_amd.tacotron import Tacotron
                tm = Tacotron(hp, num_speakers=2, device=dev)
                trng = np.random.RandomState(7)
                tt = {}
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
                tm.load_weights(tt)
                TN, TT = 32, 101
                tok = trng.randint(2, 80, (TN, TT)).astype(np.int32); tok[:, -1] = 1
                tln = np.full(TN, TT, np.int32); tsp = (np.arange(TN) % 2).astype(np.int32)
                tm.infer(tok, tln, tsp); torch.cuda.synchronize()
                q0 = time.perf_counter()
                for _ in range(3): tmel, _, _ = tm.infer(tok, tln, tsp)
                torch.cuda.synchronize()
                qdt = (time.perf_counter() - q0) / 3
                # the matrix-core kernels of the pass, timed live: HIP events around every dense contraction of one more pass
                tm.set_option("gemm_timing", 1)
                tm.infer(tok, tln, tsp); torch.cuda.synchronize()
                gflop, gms, gn = tm.gemm_stats()
                tm.set_option("gemm_timing", 0)
                # the decoder (tc_decoder_x_kernel: one persistent launch, 200 steps): per-step time = (200-step pass - 1-step pass) / 199,
                # both without the post-net; its ruler is a latency floor, from the stamped anatomy of the same kernel
                # (scripts/tacotron_xdec_profile.py -> profiles/r06_tacotron_xdec_phase_profile.txt)
                import copy
                hp1 = copy.copy(hp); hp1.max_iters = 1
                tm1 = Tacotron(hp1, num_speakers=2, device=dev)
                tm1.load_weights(tt)
                def _mel_only(model, reps=3):
                    model.infer(tok, tln, tsp, want_linear=False); torch.cuda.synchronize()
                    c0 = time.perf_counter()
                    for _ in range(reps): model.infer(tok, tln, tsp, want_linear=False)
                    torch.cuda.synchronize()
                    return (time.perf_counter() - c0) / reps
                dec_us = (_mel_only(tm) - _mel_only(tm1)) * 1e6 / (hp.max_iters - 1)
                # pieces of one step (us, workgroup 0 of utterance 0, 160 steps averaged): everything that is not an exchange, and the hop
                DEC_NONEXCH_US, DEC_EXCHANGES, DEC_HOP_US = 18.48, 13, 0.27
                dec_floor = DEC_NONEXCH_US + DEC_EXCHANGES * DEC_HOP_US
                dec_alg_bytes = hp.max_iters * 1.63e6 * 4 + TN * (2 * TT * 256 * 4 + hp.max_iters * hp.reduction_factor * hp.num_mels * 4)
                del tm1
                # other batch sizes on the same GPU (not the metric's configuration: the decoder is a latency chain, so more utterances
                # per pass cost little; at 64 the 256 CUs hold 4 workgroups per utterance instead of 8)
                bsweep = []
                for TB in (8, 16, 64):
                    tokb = trng.randint(2, 80, (TB, TT)).astype(np.int32); tokb[:, -1] = 1
                    tlb = np.full(TB, TT, np.int32); tsb = (np.arange(TB) % 2).astype(np.int32)
                    tm.infer(tokb, tlb, tsb); torch.cuda.synchronize()
                    c0 = time.perf_counter()
                    for _ in range(2): tm.infer(tokb, tlb, tsb)
                    torch.cuda.synchronize()
                    bdt = (time.perf_counter() - c0) / 2
                    bsweep.append({"batch": TB, "ms_per_pass": bdt * 1e3, "mel_frames_per_s": TB * hp.max_iters * hp.reduction_factor / bdt,
                                   "decoder_kernel": tm.decoder_kernel_name(TB, TT)})
                res["tacotron"] = {"metric": "Tacotron mel frames/sec", "value": TN * hp.max_iters * hp.reduction_factor / qdt,
                                   "unit": "mel frames/s", "ms_per_pass": qdt * 1e3, "dtype": "f32", "batch_sweep": bsweep,
                                   # the DOMINANT kernel's ruler first: the decoder is 64 % of the pass and a latency chain
                                   "roofline": {"bound": "hbm", "kernel": tm.decoder_kernel_name(TN, TT) + " (XCD-resident: 32 workgroups per XCD hold 16 columns of every decoder matrix in registers "
                                                                         "for the whole launch and serve that XCD's four utterances; 11 matvec stages on v_mfma_f32_4x4x1 + attention, "
                                                                         "13 all-gathers per step through that XCD's L2)",
                                                # HBM convention as for the headline: algorithmic bytes = the decoder's weights once per pass per XCD-resident copy would be
                                                # 8 x 6.5 MB; kept as in round 5 (weights once per step + keys / memory in + mel out) so the figure stays comparable;
                                                # `traffic` = what the counters saw
                                                "achieved": dec_alg_bytes / (dec_us * hp.max_iters * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                "frac": dec_alg_bytes / (dec_us * hp.max_iters * 1e-6) / 1e9 / 8000.0,
                                                "traffic": (tacotron_decoder_traffic("B32_T101") or {}).get("bytes_per_pass"),
                                                "algorithmic_bytes_per_pass": dec_alg_bytes,
                                                "us_per_step": dec_us, "ms_per_pass": dec_us * hp.max_iters * 1e-3, "latency_floor_us": dec_floor, "frac_of_floor": dec_floor / dec_us,
                                                "formula": "per step (slice 0 of XCD 0, stamped): %.2f us outside the exchanges (tasks on the matrix core ~0.5 per stage, chunk sums / bias / "
                                                           "activation / publish ~0.3, cell updates 0.2-0.55, barriers; attention: scores 0.8, sum 0.5, monotonic recurrence 1.2, context 0.9) "
                                                           "+ %d exchanges x %.2f us (one-way granule hop inside an XCD's L2 measured in isolation; in the kernel an exchange averages "
                                                           "0.57 us from the publish to the last poll); profiles/r06_tacotron_xdec_phase_profile.txt" % (DEC_NONEXCH_US, DEC_EXCHANGES, DEC_HOP_US),
                                                "note": "what binds this kernel is its latency chain (latency_floor_us / frac_of_floor), not bandwidth: traffic = fabric bytes of the decoder kernel per pass (counters) "
                                                        "-- the weights cross the fabric once per launch.  The split kernel it replaced as the default at this batch (`split_decoder`: 8 workgroups per "
                                                        "utterance, the 6.4 MB of decoder weights re-streamed through every XCD's L2 once per step) is still what runs above batch 32 and for "
                                                        "model_type 'simple'",
                                                "split_decoder": {"kernel": "tc_decoder_g_kernel", "selected_by": "decoder_groups = 1 / 2 / 4 / 8 / 16, batch > 32, t_in > 512, non-default widths not divisible by 4, model_type 'simple'",
                                                                  "traffic_at_batch_32": (tacotron_decoder_traffic("B32_T101_split") or {}).get("bytes_per_pass"),
                                                                  "evidence": "profiles/r06_rocprofv3_tacotron_traffic.txt, profiles/r06_tacotron_decoder_ab.txt (pass times of both kernels at batch 8 / 16 / 24 / 32), "
                                                                              "profiles/r06_tacotron_decoder_phase_profile.txt"},
                                                "gemm": {"bound": "mfma", "kernel": "tc_gemm_mfma_{,group_,highway_,ck_}kernel (%d launches per pass: CBHG conv banks as one grouped launch each, projections, "
                                                                                    "fused highway layers, grouped GRU input halves, attention keys, linear)" % gn,
                                                         "achieved": gflop / (gms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": gflop / (gms * 1e-3) / 1e12 / 157.3,
                                                         "flop_per_pass": gflop, "kernel_ms_per_pass": gms, "traffic": tacotron_traffic(),
                                                         "note": "useful FLOPs (2*rows*K*N, unpadded) of the dense contractions / their summed HIP-event time; 22 % of the pass; "
                                                                 "counters: profiles/r06_rocprofv3_mfma_summary_tacotron_train.txt, r06_rocprofv3_kernel_stats_tacotron_c3.csv; HBM bytes (`traffic`): r06_rocprofv3_tacotron_traffic.txt"}},
                                   "config": {"workload": "configs[2]: Tacotron text->mel (CBHG encoder, monotonic Bahdanau attention decoder, post-CBHG, "
                                                          "linear), batch=32, 101 tokens, 200 decoder steps = 1000 mel frames/utterance, random-init weights"},
                                   "finite": bool(torch.isfinite(tmel).all().item())}
            except Exception as e:   # the headline metric must still be reported
                res["tacotron"] = {"error": repr(e)}
        if train_res is not None:
            res["train"] = train_res
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
