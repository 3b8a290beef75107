"""checkpoint bundles through the GPU-side callers: trainer save/restore, the generate CLI and Synthesizer.load"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import first_mismatch, make_model, mol_uniforms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _trainer(seed=0):
    import twvk_amd  # noqa: F401
    from twvk_amd import weights as W
    from twvk_amd.train import WaveNetTrainer
    dil = [1, 2, 4, 8]
    specs = W.tensor_specs(len(dil), S=64)
    tensors = W.random_tensors(specs, seed=seed)
    net = make_model(2, dil, tensors, S=64)
    tr = WaveNetTrainer(net, sample_size=600)
    tr.load_weights(tensors)
    return tr


def test_trainer_save_restore_resumes_identically(tmp_path):
    from twvk_amd import checkpoint as ck
    rng = np.random.RandomState(5)
    batches = [(((rng.rand(2, 600) - 0.5)).astype(np.float32), (rng.randn(2, 2, 80) * 0.5).astype(np.float32), np.array([0, 1], np.int32))
               for _ in range(5)]
    a = _trainer()
    for b in batches[:3]:
        a.step(*b)
    prefix = a.save(str(tmp_path))
    assert os.path.basename(prefix) == "model.ckpt-3" and ck.latest_checkpoint(str(tmp_path)) == prefix
    var = ck.read_bundle(prefix, verify=True)
    assert int(var["global_step"]) == 3 and "wavenet/conv1d/kernel/ExponentialMovingAverage" in var
    assert not any("queue" in k for k in var)
    b_ = _trainer(seed=9)                                            # different weights: everything must come from the bundle
    assert b_.restore(str(tmp_path), verify=True) == 3
    for name in ("params", "ema", "m", "v"):
        assert torch.equal(getattr(a, name), getattr(b_, name)), name
    la = [float(a.step(*b).item()) for b in batches[3:]]
    lb = [float(b_.step(*b).item()) for b in batches[3:]]
    assert la == lb and torch.equal(a.params, b_.params) and torch.equal(a.ema, b_.ema)
    # a weights-only bundle (what generate.py needs) restores with a fresh optimizer
    ck.write_bundle(str(tmp_path / "w" / "model.ckpt-40"), a.weights())
    c = _trainer(seed=4)
    assert c.restore(str(tmp_path / "w" / "model.ckpt-40")) == 40
    assert torch.equal(c.params, a.params) and torch.equal(c.ema, a.params) and float(c.m.abs().max()) == 0.0


def test_generate_cli_restores_a_bundle(torch_cuda, tmp_path):
    """`python generate.py <logdir>`: load() finds model.ckpt-N through the `checkpoint` state file (utils/__init__.py:75-90)"""
    from scipy.io import wavfile
    import twvk_amd
    from twvk_amd import checkpoint as ck, weights as W
    from twvk_amd.generate import main
    logdir = tmp_path / "ckpt"; logdir.mkdir()
    dil = [1, 2, 4, 8, 16, 32]
    json.dump({"dilations": dil, "skip_channels": 128}, open(logdir / "params.json", "w"))
    specs = W.tensor_specs(len(dil), S=128)
    tensors = W.random_tensors(specs, seed=11)
    var = dict(tensors)
    var["wavenet/queue/causal_queue"] = np.ones((1, 32, 1), np.float32)
    var["global_step"] = np.asarray(77, np.int32)
    ck.write_bundle(str(logdir / "model.ckpt-77"), var)
    ck.write_checkpoint_state(str(logdir), str(logdir / "model.ckpt-77"))
    mel = np.random.RandomState(0).uniform(-4, 4, (2, 80)).astype(np.float32)
    np.save(tmp_path / "mel.npy", mel)
    try:
        args = ["--mel", str(tmp_path / "mel.npy"), "--gc_cardinality", "2", "--gc_id", "1", "--batch_size", "1", "--seed", "3"]
        p1 = main([str(logdir)] + args + ["--logdir", str(tmp_path / "log1")])
        # the same weights through the .npz route give the same wave
        npz = tmp_path / "npz"; npz.mkdir()
        json.dump({"dilations": dil, "skip_channels": 128}, open(npz / "params.json", "w"))
        np.savez(npz / "wavenet_weights.npz", **tensors)
        p2 = main([str(npz)] + args + ["--logdir", str(tmp_path / "log2")])
        a, b = wavfile.read(p1[0])[1], wavfile.read(p2[0])[1]
        assert a.shape == (600,) and np.array_equal(a, b)
    finally:
        twvk_amd.hparams.__dict__.update(twvk_amd.default_hparams().__dict__)


def test_synthesizer_loads_a_bundle(torch_cuda, oracle, tmp_path):
    from twvk_amd import checkpoint as ck
    from twvk_amd.hparams import default_hparams
    from twvk_amd.tacotron import Synthesizer
    hp = default_hparams()
    hp.max_iters, hp.num_freq = 4, 65
    d = oracle.taco_dims(max_iters=4, num_freq=65)
    tensors = oracle.taco_random_tensors(d, seed=5)
    for step in (1000, 3000):
        ck.write_bundle(str(tmp_path / ("model.ckpt-%d" % step)),
                        ck.tacotron_variables(tensors if step == 3000 else {k: v * 0 + 1 for k, v in tensors.items()}))
    toks = [[5, 9, 33, 12, 1], [7, 7, 1]]
    ref = Synthesizer(); ref.load(tensors, num_speakers=2, hparams=hp)
    want = ref.infer(toks, speaker_ids=[1, 0])["mel"].cpu().numpy()
    syn = Synthesizer(); syn.load(str(tmp_path), num_speakers=2, hparams=hp)               # directory -> most recent step
    assert first_mismatch(syn.infer(toks, speaker_ids=[1, 0])["mel"].cpu().numpy(), want) is None
    syn = Synthesizer(); syn.load(str(tmp_path / "model.ckpt-3000"), num_speakers=2, hparams=hp)   # one bundle prefix
    assert first_mismatch(syn.infer(toks, speaker_ids=[1, 0])["mel"].cpu().numpy(), want) is None
