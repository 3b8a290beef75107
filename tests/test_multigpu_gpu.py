"""-m gpu, needs >= 2 visible GPUs (skipped on the one-GPU build box): the N > 1 paths on real devices -- utterance sharding with
torch.distributed's "nccl" backend (= RCCL over xGMI), the gradient all-reduce of the training step, every rank's samples against the
checker.  SURVEY.md 8(e): no collective on the generation data path; ONE all-reduce per training step."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _bench(*flags):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_on_two_gpus():
    """bench.py --gpus 2 (it starts its own two ranks): both ranks produce samples and both match the checker on their own batch
    (n_gpus == 2, checked_ranks == 2); the training secondary's gradient all-reduce really ran over RCCL"""
    if _visible_gpus() < 2:
        pytest.skip("needs >= 2 visible GPUs (the driver's multi-GPU node)")
    line = _bench("--gpus", "2", "--seconds", "0.25", "--steps", "1", "--warmup", "1", "--no-tacotron", "--no-sweep")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["checked_ranks"] == 2, line.get("checked_against_oracle")
    assert "RCCL" in line["train"]["collective"] and line["train"]["n_gpus"] == 2
    assert line["train"]["loss_last"] < line["train"]["loss_first"]
    assert line["value"] > 0 and line["config"]["batch_per_gpu"] == 8


def test_end_to_end_on_two_gpus():
    """bench.py --e2e --gpus 2: BASELINE configs[4]'s utterance sharding (8 utterances, 4 per GPU here) through Tacotron and the vocoder"""
    if _visible_gpus() < 2:
        pytest.skip("needs >= 2 visible GPUs (the driver's multi-GPU node)")
    line = _bench("--e2e", "--gpus", "2", "--e2e-frames", "40")
    assert line["n_gpus"] == 2 and line["finite"] is True and line["value"] > 0


def test_bench_line_on_one_gpu_has_every_contract_field():
    """the one-GPU line at a short utterance length: metric / roofline (HBM convention + latency floor + fp32 fraction) / cpu_baseline /
    secondaries with their own rooflines; end to end through the launcher code path (--gpus 1)"""
    if _visible_gpus() < 1:
        pytest.skip("needs a GPU")
    line = _bench("--seconds", "0.25", "--steps", "1", "--warmup", "1", "--cpu-seconds", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and 0 < rf["frac_of_floor"] <= 1.0 and 0 < rf["fp32_frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert line["checked_ranks"] == 1
    assert [e["streams"] for e in line["streams_sweep"]] == [8, 16, 32, 48, 64, 72, 96]
    assert all("error" not in e for e in line["streams_sweep"]), line["streams_sweep"]
    assert line["streams_sweep"][-1]["kernel"] == "wn_xcd_many_kernel"
    trf = line["tacotron"]["roofline"]                          # the dominant kernel (the decoder) first, the matrix-core kernels under "gemm"
    assert trf["bound"] == "hbm" and 0 < trf["frac"] < 1 and 0 < trf["frac_of_floor"] <= 1.0 and trf["kernel"].startswith("tc_decoder_x_kernel")
    assert trf["gemm"]["bound"] == "mfma" and 0 < trf["gemm"]["frac"] < 1
    assert [e["batch"] for e in line["tacotron"]["batch_sweep"]] == [8, 16, 64] and [e["decoder_kernel"] for e in line["tacotron"]["batch_sweep"]] == ["tc_decoder_x_kernel", "tc_decoder_x_kernel", "tc_decoder_g_kernel"]
    assert line["train"]["roofline"]["bound"] == "mfma" and 0 < line["train"]["roofline"]["frac"] < 1
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
