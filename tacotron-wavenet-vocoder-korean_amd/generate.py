"""generate.py of the reference (generate.py:51-269) on the MI355X path: same positional argument and flags, same
output files (`<logdir>/generate/<timestamp>/test-{i}.wav`), the per-sample `sess.run` loop replaced by ONE persistent
kernel launch.

Differences that are deliberate and visible: the checkpoint is the `model.ckpt-N` bundle the directory's `checkpoint`
file names (read by checkpoint.py without TensorFlow; unpinned against a TF-written file) or, without one,
`<checkpoint_dir>/wavenet_weights.npz` (numpy arrays keyed by the TF variable names of wavenet/model.py); `--wav_seed` takes
a wav at the model's sample rate or a .npy (no librosa resampling / silence trimming, both are host DSP outside the
path); `--seed` makes the sampler's uniforms reproducible (the reference is unseeded)."""
import argparse
import os
import time
from datetime import datetime

import numpy as np

from .hparams import hparams, load_hparams


def get_arguments(argv=None):
    def _ensure_positive_float(f):
        if float(f) < 0:
            raise argparse.ArgumentTypeError('Argument must be greater than zero')
        return float(f)

    parser = argparse.ArgumentParser(description='WaveNet generation script')
    parser.add_argument('checkpoint_dir', type=str, help='Which model checkpoint to generate from')
    parser.add_argument('--temperature', type=_ensure_positive_float, default=1.0, help='Sampling temperature')
    parser.add_argument('--logdir', type=str, default='./logdir-wavenet', help='Directory in which to store the output')
    parser.add_argument('--wav_out_path', type=str, default=None, help='Path to output wav file')
    parser.add_argument('--batch_size', type=int, default=1, help='batch size')
    parser.add_argument('--wav_seed', type=str, default=None, help='The wav file to start generation from')
    parser.add_argument('--mel', type=str, default=None, help='mel input')
    parser.add_argument('--gc_cardinality', type=int, default=None, help='Number of categories upon which we globally condition.')
    parser.add_argument('--gc_id', type=int, default=None, help='ID of category to generate, if globally conditioned.')
    parser.add_argument('--seed', type=int, default=None, help='seed of the sampler uniforms (extension)')
    parser.add_argument('--random_init', action='store_true', help='random N(0,0.05^2) weights when no checkpoint exists (plumbing)')
    args = parser.parse_args(argv)
    return args


def save_wav(wav_int16, path, sr):
    """utils/audio.py:14-17: the peak normalisation / int16 conversion already ran on the device (ops.wav_to_int16)"""
    from scipy.io import wavfile
    wavfile.write(path, sr, np.asarray(wav_int16, dtype=np.int16))


def _load_seed(path, sr):
    if path.endswith('.npy'):
        return np.load(path).astype(np.float32)
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if rate != sr:
        raise ValueError('seed wav is %d Hz, the model runs at %d Hz (no resampling on this path)' % (rate, sr))
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    return data / 32768.0 if np.abs(data).max() > 1.0 else data


def main(argv=None):
    import torch
    from .wavenet import WaveNetModel
    from . import weights as W
    from .ops import mu_law_decode, mu_law_encode

    config = get_arguments(argv)
    started = "{0:%Y-%m-%dT%H-%M-%S}".format(datetime.now())
    logdir = os.path.join(config.logdir, 'generate', started)
    os.makedirs(logdir, exist_ok=True)
    if os.path.exists(os.path.join(config.checkpoint_dir, 'params.json')):
        load_hparams(hparams, config.checkpoint_dir)                                  # generate.py:114
    if hparams.gc_channels is not None:                                               # generate.py:72-77
        if config.gc_cardinality is None:
            raise ValueError("Globally conditioning but gc_cardinality not specified. Use --gc_cardinality=377 for full VCTK corpus.")
        if config.gc_id is None:
            raise ValueError("Globally conditioning, but global condition was not specified. Use --gc_id to specify global condition.")
    if config.mel is None:
        raise ValueError("--mel is required (generate.py:151)")

    B = config.batch_size
    scalar_input = hparams.scalar_input
    net = WaveNetModel(batch_size=B, dilations=hparams.dilations, filter_width=hparams.filter_width,
                       residual_channels=hparams.residual_channels, dilation_channels=hparams.dilation_channels,
                       quantization_channels=hparams.quantization_channels, out_channels=hparams.out_channels,
                       skip_channels=hparams.skip_channels, use_biases=hparams.use_biases, scalar_input=scalar_input,
                       initial_filter_width=hparams.initial_filter_width, global_condition_channels=hparams.gc_channels,
                       global_condition_cardinality=config.gc_cardinality, local_condition_channels=hparams.num_mels,
                       upsample_factor=hparams.upsample_factor, train_mode=False)      # generate.py:121-137
    from . import checkpoint as ckpt
    wpath = os.path.join(config.checkpoint_dir, 'wavenet_weights.npz')
    if ckpt.latest_checkpoint(config.checkpoint_dir):                                  # utils/__init__.py:75-90 load()
        prefix = ckpt.latest_checkpoint(config.checkpoint_dir)
        print('Restoring model from {}'.format(config.checkpoint_dir))
        print("  Checkpoint found: {}".format(prefix))
        print("  Global step was: {}".format(ckpt.checkpoint_step(prefix)))
        # raw variables by name (generate.py:157): only the model's own variables are read (not the Adam moments / EMA shadows the
        # training graph also saved), every tensor's CRC-32C is checked
        tensors = ckpt.wavenet_tensors(ckpt.restore_variables(prefix, net.specs, verify=True), net.specs)     # generate.py:157-161, name-tolerant
    elif os.path.exists(wpath):
        print('Restoring model from {}'.format(config.checkpoint_dir))
        tensors = dict(np.load(wpath))
    elif config.random_init:
        tensors = W.random_tensors(net.specs, seed=0, scale=0.05)
    else:
        raise FileNotFoundError(wpath)
    net.load_weights(tensors)                                                          # generate.py:157-163 (+ queue_initializer)

    mel_input = np.load(config.mel).astype(np.float32)                                 # generate.py:151
    sample_size = mel_input.shape[0] * hparams.hop_size
    mel_input = np.tile(mel_input, (B, 1, 1))
    upsampled = net.create_upsample(mel_input)                                         # generate.py:154,200
    gc = [config.gc_id] * B if hparams.gc_channels is not None else None
    Q = hparams.quantization_channels
    rng = np.random.RandomState(config.seed)

    if config.wav_seed:                                                                # generate.py:168-181
        seed = _load_seed(config.wav_seed, hparams.sample_rate)[:net.receptive_field]
        if not scalar_input:
            seed = mu_law_encode(seed, Q).cpu().numpy()
        seed = np.tile(seed[None, :], (B, 1))
        print('Priming generation...')
        if seed.shape[1] > 1:
            net.prime(seed[:, -net.receptive_field:-1], None, gc)
        print('Done.')
        first = seed[:, -1]
    elif scalar_input:                                                                 # generate.py:185-188
        first = (2 * rng.rand(B) - 1).astype(np.float32)
    else:                                                                              # generate.py:190-192
        first = rng.randint(Q, size=B).astype(np.int32)

    start_time = time.time()
    if scalar_input:
        nr = hparams.out_channels // 3
        lo, hi = np.float32(1e-5), np.float32(1. - 1e-5)                               # mixture.py:103,110
        u = (rng.random_sample((B, sample_size, nr + 1)).astype(np.float32) * (hi - lo) + lo).astype(np.float32)
    else:
        u = rng.random_sample((B, sample_size))                                        # generate.py:231
    out = net.generate(upsampled, gc, first, u, temperature=config.temperature)        # generate.py:202-233
    torch.cuda.synchronize()
    print('Sample {0}/{0}, ({1:.3f} sec)'.format(sample_size, time.time() - start_time))

    if hparams.input_type == 'raw':                                                    # generate.py:249-256
        wav = out
    elif hparams.input_type == 'mulaw':
        wav = mu_law_decode(out, Q, quantization=False)
    else:
        wav = mu_law_decode(out, Q, quantization=True)
    from .ops import wav_to_int16
    wav = wav_to_int16(wav).cpu().numpy()
    paths = []
    for i in range(B):                                                                 # generate.py:259-262
        path = config.wav_out_path if (config.wav_out_path and B == 1) else logdir + '/test-{}.wav'.format(i)
        save_wav(wav[i], path, hparams.sample_rate)
        paths.append(path)
    print('Finished generating.')
    return paths


if __name__ == '__main__':
    s = time.time()
    main()
    print(time.time() - s, 'sec')
