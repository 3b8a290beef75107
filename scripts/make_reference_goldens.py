#!/usr/bin/env python
"""Generates tests/golden/reference_*.npz by RUNNING THE REFERENCE ITSELF under TensorFlow 1.x.

    python scripts/make_reference_goldens.py --reference /path/to/Tacotron-Wavenet-Vocoder-Korean [--out tests/golden]

NOT runnable in the build image: TensorFlow 1.x (tf.contrib: hparams.py:6), jamo and librosa are absent there and there is no
network, so this script has never been executed by this project -- it is the committed recipe a
maintainer with a TF 1.x environment runs ONCE; the .npz files it writes are data (seeded inputs + the reference's outputs) and
travel with the repo, the reference's sources never do.  tests/test_reference_goldens.py skips while the files are absent and,
once they exist, holds oracle/ (and through it the HIP path) to them: integer class ids bit-exact, floats within 1e-4
(north_star's bars).  Until then every parity claim of this repo stays "against the restatement" (DESIGN.md section 2).

What is pinned, with the same seeds / inputs as tests/golden/make_restatement.py:
  reference_codec.npz            wavenet/ops.py:22-47 mu_law_encode / mu_law_decode
  reference_wavenet_mol_small.npz   WaveNetModel(scalar_input=True, train_mode=False): create_upsample (model.py:102-111) and the RAW
                                 network output of _create_network (model.py:112-167) step by step for a teacher-forced input
                                 sequence (the MoL sampler draws unseeded tf.random_uniform inside the graph: its output cannot be
                                 reproduced, its input can), plus the full-convolution forward (train_mode=True)
  reference_wavenet_mulaw_small.npz predict_proba_incremental (model.py:215-245, float64 softmax) + the host loop of
                                 generate.py:199-233 with np.random seeded: class ids, both temperatures
  reference_tacotron_small.npz   Tacotron.initialize(..., rnn_decoder_test_mode=True) (tacotron.py:36-235): mel, linear, alignments
  reference_variable_names.json  the variable names TensorFlow actually gave each graph (closes the [RECALLED-TF] auto-naming
                                 question of checkpoint.py / weights.py)
  reference_ckpt_wavenet/        a REAL tf.train.Saver bundle (model.ckpt-7.index / .data-00000-of-00001 + the `checkpoint` state file)
  reference_ckpt_tacotron/       of a small WaveNet generation graph and of a small (1/8-width) two-speaker Tacotron graph, next to
                                 values.npz = the same variables read back through TensorFlow itself: what checkpoint.read_bundle /
                                 restore_variables (generate.py:157-161, synthesizer.py:69-70 Saver.restore) are held to bit for bit
                                 -- the importer has only ever read files it wrote itself
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _tensors_and_dims():
    """the seeded weights of tests/golden/make_restatement.py (pure numpy: the oracle's C library is not needed for them)"""
    from oracle import oracle as O
    from helpers import make_case
    return O, make_case


def _assign(tf, sess, tensors, prefix, report, key):
    """load `tensors` ({spec name: array}) into the graph's variables.  Names are matched exactly first; what TensorFlow named
    differently is matched through checkpoint.remap_names (scope path + shape + creation order) and REPORTED -- that report is
    itself a result: it says which recalled names were wrong."""
    import twvk_amd  # noqa: F401
    from twvk_amd import checkpoint as ck
    gvars = {v.name.split(":")[0]: v for v in tf.global_variables()}
    wanted = [(prefix + n, tuple(np.shape(a))) for n, a in tensors.items()]
    avail = {n: tuple(int(d) for d in v.shape) for n, v in gvars.items()}
    mapping = ck.remap_names(wanted, avail)
    missing = [n for n, _ in wanted if n not in mapping]
    if missing:
        raise SystemExit("graph has no variable for: %s\n  graph variables: %s" % (missing[:8], sorted(avail)[:40]))
    report[key] = {"graph_variables": sorted(avail), "renamed": {w: a for w, a in mapping.items() if w != a}}
    for (w, _), a in zip(wanted, tensors.values()):
        gvars[mapping[w]].load(np.asarray(a, np.float32), sess)
    unset = [n for n in avail if n not in set(mapping.values()) and "queue" not in n and "ExponentialMovingAverage" not in n]
    report[key]["graph_variables_left_at_init"] = unset


def codec(tf, ref, out):
    from wavenet.ops import mu_law_encode, mu_law_decode
    rng = np.random.RandomState(21)
    a = np.concatenate([rng.uniform(-1.2, 1.2, 2000), [0.0, 1.0, -1.0, 0.5, -0.25]]).astype(np.float32)
    with tf.Graph().as_default(), tf.Session() as sess:
        q = sess.run(mu_law_encode(tf.constant(a), 256))
        dec = sess.run(mu_law_decode(tf.constant(np.arange(256, dtype=np.int32)), 256))
        exp = sess.run(mu_law_decode(tf.constant(np.linspace(-1, 1, 513).astype(np.float32)), 256, quantization=False))
    np.savez_compressed(os.path.join(out, "reference_codec.npz"), audio=a, q=q.astype(np.int32), dec=dec.astype(np.float32),
                        expand_in=np.linspace(-1, 1, 513).astype(np.float32), expand=exp.astype(np.float32))


def wavenet_mol(tf, ref, out, report):
    from wavenet import WaveNetModel
    O, make_case = _tensors_and_dims()
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    B, Tm, T = 2, 1, 96
    d, tensors, _ = make_case(O, dil, S=128, scale=0.15, seed=11)
    rng = np.random.RandomState(12)
    mel = rng.uniform(-4, 4, (B, Tm, 80)).astype(np.float32)
    gc = np.array([1, 0], np.int32)
    forced = rng.uniform(-1, 1, (B, T)).astype(np.float32)           # teacher-forced inputs (the sampler is not reproducible)
    kw = dict(batch_size=B, dilations=dil, filter_width=2, residual_channels=32, dilation_channels=32, skip_channels=128,
              quantization_channels=256, out_channels=30, use_biases=True, scalar_input=True, initial_filter_width=32,
              global_condition_channels=32, global_condition_cardinality=2, local_condition_channels=80, upsample_factor=[5, 5, 12])
    with tf.Graph().as_default(), tf.Session() as sess:
        net = WaveNetModel(train_mode=False, **kw)
        x = tf.placeholder(tf.float32, [B, None]); lc = tf.placeholder(tf.float32, [B, 80])
        with tf.variable_scope("wavenet"):                           # predict_proba_incremental's scope (model.py:221), without its sampler
            enc = tf.reshape(x, [B, -1, 1])
            emb = net._embed_gc([int(v) for v in gc])
            raw = net._create_network(enc, tf.reshape(lc, [B, -1, 80]), emb)
        with tf.variable_scope("wavenet", reuse=tf.AUTO_REUSE):
            up = net.create_upsample(tf.constant(mel))              # generate.py:153-155
        sess.run(tf.global_variables_initializer())
        _assign(tf, sess, tensors, "", report, "wavenet_mol_incremental")
        sess.run(net.queue_initializer)
        U = sess.run(up)
        raws = np.stack([sess.run(raw, {x: forced[:, t:t + 1], lc: U[:, t]}).reshape(B, 30) for t in range(T)], axis=1)
    with tf.Graph().as_default(), tf.Session() as sess:                # the same weights through the full-convolution graph
        net = WaveNetModel(train_mode=True, **kw)
        rf = net.receptive_field
        xin = np.concatenate([np.zeros((B, rf - 1), np.float32), forced], axis=1)
        with tf.variable_scope("wavenet"):
            emb = net._embed_gc([int(v) for v in gc])
            up = net.create_upsample(tf.constant(mel))
            full = net._create_network(tf.reshape(tf.constant(xin), [B, -1, 1]), up[:, :xin.shape[1]], emb)   # lc as long as the input: model.py:79-80 slices its FRONT
        sess.run(tf.global_variables_initializer())
        _assign(tf, sess, tensors, "", report, "wavenet_mol_full")
        raw_full = sess.run(full)
    np.savez_compressed(os.path.join(out, "reference_wavenet_mol_small.npz"), dilations=np.array(dil), S=128, scale=0.15, weight_seed=11,
                        mel=mel, gc_ids=gc, forced=forced, upsampled=U, raw_incremental=raws, raw_full=raw_full, receptive_field=rf)


def wavenet_mulaw(tf, ref, out, report):
    from wavenet import WaveNetModel
    O, make_case = _tensors_and_dims()
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    B, T = 2, 48
    d, tensors, _ = make_case(O, dil, scalar_input=False, S=128, Q=256, scale=0.3, seed=31)
    rng = np.random.RandomState(32)
    U = rng.uniform(-4, 4, (B, T, 80)).astype(np.float32)
    gc = np.array([0, 1], np.int32)
    seed_in = rng.randint(256, size=B).astype(np.int32)
    res = {}
    with tf.Graph().as_default(), tf.Session() as sess:
        net = WaveNetModel(batch_size=B, dilations=dil, filter_width=2, residual_channels=32, dilation_channels=32, skip_channels=128,
                           quantization_channels=256, out_channels=30, use_biases=True, scalar_input=False, initial_filter_width=32,
                           global_condition_channels=32, global_condition_cardinality=2, local_condition_channels=80,
                           upsample_factor=[5, 5, 12], train_mode=False)
        samples = tf.placeholder(tf.int32, [B, None]); lc = tf.placeholder(tf.float32, [B, 80])
        nxt = net.predict_proba_incremental(samples, lc, [int(v) for v in gc])          # generate.py:147
        sess.run(tf.global_variables_initializer())
        _assign(tf, sess, tensors, "", report, "wavenet_mulaw")
        for temp, key in ((1.0, "t10"), (0.8, "t08")):
            sess.run(net.queue_initializer)
            np.random.seed(33)                                       # generate.py:231 draws from numpy's global RandomState
            waveform = seed_in.reshape(B, 1)
            probs, us = [], []
            state = np.random.get_state()
            for step in range(T):                                    # generate.py:199-233, the one-hot branch, verbatim in effect
                window = waveform[:, -1:]
                prediction = sess.run(nxt, {samples: window, lc: U[:, step]})
                probs.append(prediction)
                np.seterr(divide="ignore")
                scaled = np.log(prediction) / temp
                scaled = scaled - np.logaddexp.reduce(scaled, axis=-1, keepdims=True)
                scaled = np.exp(scaled)
                np.seterr(divide="warn")
                sample = [[np.random.choice(np.arange(256), p=p)] for p in scaled]
                waveform = np.concatenate([waveform, sample], axis=-1)
            # the uniform draws np.random.choice consumed (one random_sample() per call, batch order): what the injected-u path needs
            np.random.set_state(state)
            us = np.array([[np.random.random_sample() for _ in range(B)] for _ in range(T)]).T
            res["samples_" + key] = waveform[:, 1:].astype(np.int32)
            res["uniforms_" + key] = us
            res["proba_" + key] = np.stack(probs, axis=1)
    np.savez_compressed(os.path.join(out, "reference_wavenet_mulaw_small.npz"), dilations=np.array(dil), S=128, Q=256, scale=0.3, weight_seed=31,
                        upsampled=U, gc_ids=gc, first_input=seed_in, **res)


def tacotron(tf, ref, out, report):
    from hparams import hparams as hp
    from tacotron import create_model
    from text.symbols import symbols
    O, _ = _tensors_and_dims()
    kw = dict(enc_bank=4, post_bank=3, max_iters=6, num_freq=129)
    hp.enc_bank_size, hp.post_bank_size, hp.max_iters, hp.num_freq = 4, 3, 6, 129
    for n_speakers, name in ((2, "reference_tacotron_small.npz"), (1, "reference_tacotron_small_single_speaker.npz")):
        d = O.taco_dims(n_symbols=len(symbols), n_speakers=n_speakers, **kw)
        tensors = O.taco_random_tensors(d, seed=41)
        rng = np.random.RandomState(42)
        N, T, lengths = 3, 17, [17, 11, 6]
        tok = rng.randint(2, min(80, len(symbols)), (N, T)).astype(np.int32)
        for n, ln in enumerate(lengths):
            tok[n, ln - 1] = 1
            tok[n, ln:] = 0
        spk = np.array([0, 1, 0], np.int32)
        flat = {}
        for n, a in tensors.items():                                  # (4, C) batch-norm stacks -> TF's four variables
            if n.endswith("batch_normalization"):
                for i, part in enumerate(("gamma", "beta", "moving_mean", "moving_variance")):
                    flat[n + "/" + part] = a[i]
            else:
                flat[n] = a
        with tf.Graph().as_default(), tf.Session() as sess:
            inputs = tf.placeholder(tf.int32, [None, None]); ilen = tf.placeholder(tf.int32, [None]); sid = tf.placeholder(tf.int32, [None])
            with tf.variable_scope("model"):                          # synthesizer.py:52-56
                model = create_model(hp)
                model.initialize(inputs, ilen, n_speakers, sid, rnn_decoder_test_mode=True)
            sess.run(tf.global_variables_initializer())
            _assign(tf, sess, flat, "model/inference/", report, "tacotron_%d_speakers" % n_speakers)
            feed = {inputs: tok, ilen: np.asarray(lengths, np.int32), sid: spk,
                    model.is_manual_attention: False, model.manual_alignments: np.zeros([1, 1, 1], np.float32)}
            mel, lin, al = sess.run([model.mel_outputs, model.linear_outputs, model.alignments], feed)
        np.savez_compressed(os.path.join(out, name), weight_seed=41, n_symbols=len(symbols), n_speakers=n_speakers, tokens=tok,
                            lengths=np.asarray(lengths, np.int32), speaker_ids=spk, mel=mel, linear=lin, alignments=al,
                            **{"dims_" + k: v for k, v in kw.items()})


def _save_bundle(tf, sess, out_dir, step, dims):
    """tf.train.Saver over every non-queue global variable (what generate.py:157-161 / synthesizer.py:69-70 restore) + the values
    TensorFlow itself reads back, as data"""
    os.makedirs(out_dir, exist_ok=True)
    gvars = [v for v in tf.global_variables() if "queue" not in v.name]
    saver = tf.train.Saver(var_list=gvars, max_to_keep=3)
    prefix = saver.save(sess, os.path.join(out_dir, "model.ckpt"), global_step=step, write_meta_graph=False)
    names = [v.name.split(":")[0] for v in gvars]
    vals = sess.run(gvars)
    np.savez_compressed(os.path.join(out_dir, "values.npz"), **{"t%d" % i: np.asarray(a) for i, a in enumerate(vals)})
    with open(os.path.join(out_dir, "values.json"), "w") as fh:
        json.dump({"prefix": os.path.basename(prefix), "names": names, "dtypes": [str(np.asarray(a).dtype) for a in vals],
                   "shapes": [list(np.shape(a)) for a in vals], "dims": dims, "tensorflow": tf.__version__}, fh, indent=1)
    return prefix


def saver_bundles(tf, ref, out, report):
    from wavenet import WaveNetModel
    # ---- WaveNet: the generation graph of generate.py:117-147 (small), random initial values under a fixed graph seed
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    wdims = dict(dilations=dil, residual_channels=32, dilation_channels=32, skip_channels=128, quantization_channels=256, out_channels=30,
                 scalar_input=True, initial_filter_width=32, gc_channels=32, gc_cardinality=2, lc_channels=80, upsample_factor=[5, 5, 12])
    with tf.Graph().as_default(), tf.Session() as sess:
        tf.set_random_seed(51)
        net = WaveNetModel(batch_size=2, dilations=dil, filter_width=2, residual_channels=32, dilation_channels=32, skip_channels=128,
                           quantization_channels=256, out_channels=30, use_biases=True, scalar_input=True, initial_filter_width=32,
                           global_condition_channels=32, global_condition_cardinality=2, local_condition_channels=80,
                           upsample_factor=[5, 5, 12], train_mode=False)
        samples = tf.placeholder(tf.float32, [2, None]); lc = tf.placeholder(tf.float32, [2, 80])
        net.predict_proba_incremental(samples, lc, [0, 1])                               # generate.py:147
        with tf.variable_scope("wavenet", reuse=tf.AUTO_REUSE):
            net.create_upsample(tf.zeros([2, 1, 80]))                                    # generate.py:153-155
        tf.train.get_or_create_global_step()                                             # an int64 scalar, as train_vocoder.py's bundles carry
        sess.run(tf.global_variables_initializer())
        _save_bundle(tf, sess, os.path.join(out, "reference_ckpt_wavenet"), 7, wdims)
    # ---- Tacotron: synthesizer.py:52-56 with every width at 1/8 (a default-width bundle is ~28 MB: not a fixture)
    from hparams import hparams as hp
    from tacotron import create_model
    from text.symbols import symbols
    tdims = dict(embedding_size=32, enc_prenet_sizes=[32, 16], enc_bank_size=4, enc_bank_channel_size=16, enc_proj_sizes=[16, 16],
                 enc_rnn_size=16, attention_size=32, attention_state_size=32, dec_rnn_size=32, dec_prenet_sizes=[32, 16],
                 post_bank_size=3, post_bank_channel_size=16, post_proj_sizes=[32, 80], post_rnn_size=16, num_freq=129, max_iters=4)
    for k, v in tdims.items():
        setattr(hp, k, v)
    with tf.Graph().as_default(), tf.Session() as sess:
        tf.set_random_seed(52)
        inputs = tf.placeholder(tf.int32, [None, None]); ilen = tf.placeholder(tf.int32, [None]); sid = tf.placeholder(tf.int32, [None])
        with tf.variable_scope("model"):
            model = create_model(hp)
            model.initialize(inputs, ilen, 2, sid, rnn_decoder_test_mode=True)
        sess.run(tf.global_variables_initializer())
        _save_bundle(tf, sess, os.path.join(out, "reference_ckpt_tacotron"), 3, dict(tdims, n_symbols=len(symbols), num_speakers=2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference", help="checkout of hccho2/Tacotron-Wavenet-Vocoder-Korean")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="", help="comma list of: codec,mol,mulaw,tacotron,ckpt")
    args = ap.parse_args()
    try:
        import tensorflow as tf
    except ImportError:
        raise SystemExit("TensorFlow 1.x is required (the reference imports tf.contrib); this image has none -- run this where it exists")
    if not tf.__version__.startswith("1."):
        raise SystemExit("the reference needs TensorFlow 1.x (tf.contrib.training.HParams, hparams.py:6); found " + tf.__version__)
    sys.path.insert(0, os.path.abspath(args.reference))
    os.makedirs(args.out, exist_ok=True)
    only = set(filter(None, args.only.split(",")))
    report = {"tensorflow": tf.__version__, "numpy": np.__version__}
    if not only or "codec" in only:
        codec(tf, args.reference, args.out)
    if not only or "mol" in only:
        wavenet_mol(tf, args.reference, args.out, report)
    if not only or "mulaw" in only:
        wavenet_mulaw(tf, args.reference, args.out, report)
    if not only or "tacotron" in only:
        tacotron(tf, args.reference, args.out, report)
    if not only or "ckpt" in only:
        saver_bundles(tf, args.reference, args.out, report)      # last: it edits the reference's hparams singleton
    with open(os.path.join(args.out, "reference_variable_names.json"), "w") as fh:
        json.dump(report, fh, indent=1, sort_keys=True)
    print("wrote reference_*.npz to", args.out)
    for k, v in report.items():
        if isinstance(v, dict) and v.get("renamed"):
            print("  %s: %d variables carry another name than this repo recalled:" % (k, len(v["renamed"])))
            for w, a in sorted(v["renamed"].items()):
                print("      %s  <-  %s" % (w, a))


if __name__ == "__main__":
    main()
