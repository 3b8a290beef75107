#!/usr/bin/env python
"""bench.py -- WaveNet autoregressive synthesis throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input: B=8 utterances x `--seconds` s of 24 kHz
audio (default 8 s = 192 000 samples each; BASELINE.json configs[1]), mel -> upsample -> hoisted conditioning ->
persistent generation kernel -> samples, inputs resident in HBM.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=8.0, help="audio seconds per utterance (8.0 = the BASELINE config)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--groups", type=int, default=-1, help="workgroups per stream (-1 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tacotron", action="store_true", help="skip the secondary Tacotron mel-frames/s measurement")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement (configs[3], RCCL all-reduce at N>1)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="oracle sample size in generation steps (0 = auto, about 15 s)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import twvk_amd
    from twvk_amd.wavenet import WaveNetModel
    from twvk_amd import weights as W

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(dev))

    hp = twvk_amd.default_hparams()
    dil = [2 ** i for i in range(10)] * 3            # BASELINE configs[1]: 30 dilated residual layers
    B = args.batch
    Tm = int(round(args.seconds * hp.sample_rate / hp.hop_size))
    T = Tm * hp.hop_size
    m = WaveNetModel(B, dil, hp.filter_width, hp.residual_channels, hp.dilation_channels, hp.skip_channels,
                     quantization_channels=hp.quantization_channels, out_channels=hp.out_channels, use_biases=hp.use_biases,
                     scalar_input=True, initial_filter_width=hp.initial_filter_width,
                     global_condition_channels=hp.gc_channels, global_condition_cardinality=2,
                     local_condition_channels=hp.num_mels, upsample_factor=hp.upsample_factor, train_mode=False, device=dev)
    if args.workers:
        m.set_option("workers", args.workers)
    if args.groups >= 0:
        m.set_option("groups", args.groups)
    tensors = W.random_tensors(m.specs, seed=0, scale=0.05)
    m.load_weights(tensors)
    rng = np.random.RandomState(1 + rank)
    mel = torch.from_numpy(rng.uniform(-4, 4, (B, Tm, hp.num_mels)).astype(np.float32)).to(dev)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    lo, hi = np.float32(1e-5), np.float32(1 - 1e-5)
    u = torch.from_numpy((rng.random_sample((B, T, 11)).astype(np.float32) * (hi - lo) + lo)).to(dev)

    gen_ms = []

    def one_pass(timed):
        m.queue_initializer()
        U = m.create_upsample(mel)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        # generate() = condition (hoisted projections) + the persistent kernel; events bracket the kernel on ITS stream
        B_, T_ = B, T
        cond = m._condition(U, gc, T_)
        import ctypes as C
        from twvk_amd import _lib
        fi = torch.as_tensor(seed_in, device=dev)
        out = torch.empty((B_, T_), dtype=torch.float32, device=dev)
        e0.record()
        _lib.check(m._L.twv_wavenet_generate(m._h, C.c_void_p(m._packed.data_ptr()), C.c_void_p(m._state.data_ptr()),
                                             C.c_void_p(cond.data_ptr()), C.c_void_p(fi.data_ptr()), C.c_void_p(u.data_ptr()), 1.0,
                                             B_, T_, C.c_void_p(out.data_ptr()), C.c_void_p(m._status.data_ptr()), None, 0,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        e1.record()
        return out, (e0, e1)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out, _ev = one_pass(False)
    sync_all()
    t0 = time.perf_counter()
    evs = []
    for _ in range(args.steps):
        out, ev = one_pass(True)
        evs.append(ev)
    sync_all()
    dt = time.perf_counter() - t0
    _lib_status = m._L.twv_wavenet_status
    from twvk_amd import _lib
    import ctypes as C
    _lib.check(_lib_status(C.c_void_p(m._status.data_ptr()), None))
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    gen_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    samples = out.cpu().numpy()
    assert np.isfinite(samples).all() and np.abs(samples).max() <= 1.0

    # secondary: configs[3] teacher-forced training step, per-GPU batch 64 x 8000 (cropped to 7800) samples, data-parallel with a
    # gradient all-reduce over RCCL when N > 1 -- every rank takes part, so it runs before the rank-0 report is assembled
    train_res = None
    if not args.no_train:
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
            train_res = {"metric": "WaveNet training audio samples/sec (teacher-forced step: MoL loss, backward, all-reduce, Adam, EMA)",
                         "value": world * 64 * TT_ / qdt, "unit": "audio samples/s", "steps_per_s": 1.0 / qdt, "ms_per_step": qdt * 1e3,
                         "n_gpus": world, "scaling": "weak", "dtype": "f32",
                         "collective": "all-reduce(sum) of one flat f32 gradient buffer, %d elements, RCCL" % trn.n_params if world > 1 else "none (1 GPU)",
                         "config": {"workload": "configs[3]: train_vocoder.py step, 30 layers, per-GPU batch 64 x %d samples, random-init weights" % TT_},
                         "loss_first": l0, "loss_last": float(tl.item())}
            del trn, tnet, taudio, tlc
        except Exception as e:       # the headline metric must still be reported
            train_res = {"error": repr(e)}

    if rank == 0:
        total_samples = world * B * T * args.steps
        value = total_samples / dt
        # roofline: algorithmic HBM bytes per generation step for all B streams, weights streamed every step
        NL = len(dil)
        per_layer = 2 * (2 * 32 * 32 + 32) + 2 * 80 * 32 + (32 * 32 + 32) + (32 * 512 + 512)   # gc hoisted
        wfloats = NL * per_layer + (512 * 512 + 512) + (512 * 30 + 30) + 32 * 32
        bytes_per_step = wfloats * 4 + B * (80 + 1 + 1) * 4
        k_ms = float(np.mean(gen_ms))
        achieved = bytes_per_step * T / (k_ms * 1e-3) / 1e9
        res = {
            "metric": "WaveNet autoregressive audio samples/sec at 24 kHz, batch=8",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: WaveNet autoregressive synth, 30 dilated residual layers (3x[1..512]), R=D=32, S=512, "
                                   "MoL-30 output, gc+lc conditioning, 24 kHz, batch=%d x %.2f s (%d samples each) per GPU, random-init weights, "
                                   "injected uniforms" % (B, T / hp.sample_rate, T),
                       "batch_per_gpu": B, "samples_per_utterance": T, "sharding": "utterances, one batch of %d per GPU, no collective" % B},
            "realtime_factor_aggregate": value / hp.sample_rate,
            "roofline": {"bound": "hbm", "kernel": "wn_generate_kernel", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0,
                         # HBM bytes per launch from rocprofv3 PMC (separate --pmc passes, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE),
                         # measured per generated step on this kernel (profiles/r01_rocprofv3_pmc_fetch_write_generate_0.5s_v3.csv)
                         "traffic": (1.135e6 + 0.214e6) * T if (B == 8 and NL == 30) else None, "algorithmic_bytes_per_launch": bytes_per_step * T,
                         "kernel_ms": k_ms, "us_per_generation_step": k_ms * 1e3 / T},
        }
        if not args.no_cpu_baseline and world == 1:
            from helpers import make_case
            from oracle import oracle as O
            d = O.make_dims(dil)
            blob = O.blob_from_tensors(d, tensors)
            st = O.State(d, B)
            n = args.cpu_steps
            if not n:   # size the sample for about 15 s of CPU work
                probe = 200
                Up = rng.uniform(-1, 1, (B, probe, 80)).astype(np.float32)
                c0 = time.perf_counter()
                O.generate_mol(d, blob, st, Up, gc, seed_in, u[:, :probe].cpu().numpy())
                per = (time.perf_counter() - c0) / probe
                n = int(max(probe, min(T, 15.0 / max(per, 1e-9))))
                st.reset()
            Uc = rng.uniform(-1, 1, (B, n, 80)).astype(np.float32)
            uc = u[:, :n].cpu().numpy()
            c0 = time.perf_counter()
            O.generate_mol(d, blob, st, Uc, gc, seed_in, uc)
            cdt = time.perf_counter() - c0
            res["cpu_baseline"] = {"value": B * n / cdt, "unit": "samples/s", "cores": 1, "kind": "port",
                                   "sample": "build's CPU restatement (oracle/, plain C, 1 thread) of the same model on the GPU box's host: "
                                             "B=%d x %d generation steps = %.1f s of CPU work; NOT the reference generate.py "
                                             "(TensorFlow is absent; parity unpinned)" % (B, n, cdt)}
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
                res["tacotron"] = {"metric": "Tacotron mel frames/sec", "value": TN * hp.max_iters * hp.reduction_factor / qdt,
                                   "unit": "mel frames/s", "ms_per_pass": qdt * 1e3, "dtype": "f32",
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
