#!/usr/bin/env python
"""Generates tests/golden/restatement_*.npz: seeded inputs and the outputs of THIS REPO'S CPU restatement (oracle/).

These are NOT reference golden vectors: the reference (TensorFlow 1.x) cannot be run in this image and ships none
(SURVEY.md sections 0, 4).  They pin the restatement against drift, and give the GPU tests a fixture that does not need
the oracle library at all.  Re-run after any deliberate change of the arithmetic contract."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O           # noqa: E402
from helpers import make_case, mol_uniforms   # noqa: E402


def wavenet_mol_small():
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    B, Tm = 2, 1
    d, tensors, blob = make_case(O, dil, S=128, scale=0.15, seed=11)
    rng = np.random.RandomState(12)
    mel = rng.uniform(-4, 4, (B, Tm, 80)).astype(np.float32)
    U = O.upsample(d, blob, mel)
    T = 96
    gc = np.array([1, 0], np.int32)
    seed_in = rng.uniform(-1, 1, B).astype(np.float32)
    u = mol_uniforms(B, T, 10, seed=13)
    out = O.generate_mol(d, blob, O.State(d, B), U[:, :T], gc, seed_in, u)
    raw0 = O.step(d, blob, O.State(d, B), seed_in, U[:, 0], gc)
    np.savez_compressed(os.path.join(HERE, "restatement_wavenet_mol_small.npz"), dilations=np.array(dil), S=128, scale=0.15,
                        weight_seed=11, mel=mel, upsampled_head=U[:, :8], gc_ids=gc, first_input=seed_in, uniforms=u,
                        samples=out, raw_step0=raw0)


def codec_and_math():
    rng = np.random.RandomState(21)
    a = np.concatenate([rng.uniform(-1.2, 1.2, 2000), [0.0, 1.0, -1.0, 0.5, -0.25]]).astype(np.float32)
    x = np.linspace(-20, 20, 4001).astype(np.float32)
    xp = np.exp(np.linspace(-30, 30, 2001)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "restatement_codec_math.npz"), audio=a, q=O.mu_law_encode(a, 256),
                        dec=O.mu_law_decode(np.arange(256, dtype=np.int32), 256), x=x, tanh=O.elementwise("tanh", x),
                        sigmoid=O.elementwise("sigmoid", x), exp=O.elementwise("exp", x), xp=xp, log=O.elementwise("log", xp))


def wavenet_mulaw_small():
    """one-hot mu-law model (scalar_input False, Q = 256): float64 softmax, temperature rescale, legacy np.random.choice"""
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    B, T = 2, 48
    d, tensors, blob = make_case(O, dil, scalar_input=False, S=128, Q=256, scale=0.3, seed=31)
    rng = np.random.RandomState(32)
    U = rng.uniform(-4, 4, (B, T, 80)).astype(np.float32)
    gc = np.array([0, 1], np.int32)
    seed_in = rng.randint(256, size=B).astype(np.int32)
    u = np.random.RandomState(33).random_sample((B, T))
    out = {}
    for temp in (1.0, 0.8):
        out[temp] = O.generate_mulaw(d, blob, O.State(d, B), U, gc, seed_in, u, temp)
    np.savez_compressed(os.path.join(HERE, "restatement_wavenet_mulaw_small.npz"), dilations=np.array(dil), S=128, Q=256, scale=0.3,
                        weight_seed=31, upsampled=U, gc_ids=gc, first_input=seed_in, uniforms=u, samples_t10=out[1.0], samples_t08=out[0.8])


def tacotron_small():
    """Tacotron inference on a reduced configuration (banks 4 / 3, 6 decoder steps, 129 linear bins), ragged input lengths"""
    kw = dict(enc_bank=4, post_bank=3, max_iters=6, num_freq=129)
    d = O.taco_dims(**kw)
    tensors = O.taco_random_tensors(d, seed=41)
    blob = O.taco_blob(d, tensors)
    rng = np.random.RandomState(42)
    N, T, lengths = 3, 17, [17, 11, 6]
    tok = rng.randint(2, 80, (N, T)).astype(np.int32)
    for n, ln in enumerate(lengths):
        tok[n, ln - 1] = 1
        tok[n, ln:] = 0
    spk = np.array([0, 1, 0], np.int32)
    mel, lin, al = O.taco_infer(d, blob, tok, np.asarray(lengths, np.int32), spk)
    np.savez_compressed(os.path.join(HERE, "restatement_tacotron_small.npz"), weight_seed=41, tokens=tok, lengths=np.asarray(lengths, np.int32),
                        speaker_ids=spk, mel=mel, linear=lin, alignments=al, **{"dims_" + k: v for k, v in kw.items()})


if __name__ == "__main__":
    O.build()
    wavenet_mol_small()
    codec_and_math()
    wavenet_mulaw_small()
    tacotron_small()
    print("written", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
