"""TensorFlow-free `hparams` with the reference's attribute names and defaults (hparams.py:6-176,190-192)
and the params.json round trip of utils/__init__.py:143-185.  Only the values are restated; the container
is a plain attribute bag instead of tf.contrib.training.HParams."""
import json
import os
import re

PARAMS_NAME = "params.json"   # utils/__init__.py:12


class HParams(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def values(self):
        return dict(self.__dict__)

    def to_json(self):
        return json.dumps(self.values(), sort_keys=True)

    def __repr__(self):
        return "HParams(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.values().items()))


def default_hparams():
    hp = HParams(
        name="Tacotron-Wavenet-Vocoder",
        cleaners='korean_cleaners', skip_path_filter=False, use_lws=False,
        # audio (hparams.py:18-46)
        sample_rate=24000, hop_size=300, fft_size=2048, win_size=1200, num_mels=80,
        preemphasize=True, preemphasis=0.97, min_level_db=-100, ref_level_db=20, signal_normalization=True,
        allow_clipping_in_normalization=True, symmetric_mels=True, max_abs_value=4.,
        rescaling=True, rescaling_max=0.999, trim_silence=True, trim_fft_size=512, trim_hop_size=128, trim_top_db=23,
        clip_mels_length=True, max_mel_frames=1000,
        l2_regularization_strength=0, sample_size=15000, silence_threshold=0,
        # wavenet (hparams.py:59-79)
        filter_width=2, gc_channels=32, input_type="raw", scalar_input=True,
        dilations=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 5,
        residual_channels=32, dilation_channels=32, quantization_channels=256, out_channels=30, skip_channels=512,
        use_biases=True, initial_filter_width=32, upsample_factor=[5, 5, 12],
        # wavenet training (hparams.py:84-100)
        wavenet_batch_size=8, store_metadata=False, num_steps=200000, wavenet_learning_rate=1e-3,
        wavenet_decay_rate=0.5, wavenet_decay_steps=300000, wavenet_clip_gradients=False,
        optimizer='adam', momentum=0.9, max_checkpoints=3,
        # tacotron (hparams.py:108-176)
        adam_beta1=0.9, adam_beta2=0.999, use_fixed_test_inputs=False, tacotron_initial_learning_rate=1e-3,
        decay_learning_rate_mode=0, initial_data_greedy=True, initial_phase_step=8000, main_data_greedy_factor=0,
        main_data=[''], prioritize_loss=False,
        model_type='deepvoice', speaker_embedding_size=16, embedding_size=256, dropout_prob=0.5,
        enc_prenet_sizes=[256, 128], enc_bank_size=16, enc_bank_channel_size=128, enc_maxpool_width=2,
        enc_highway_depth=4, enc_rnn_size=128, enc_proj_sizes=[128, 128], enc_proj_width=3,
        attention_type='bah_mon_norm', attention_size=256, attention_state_size=256,
        dec_layer_num=2, dec_rnn_size=256, dec_prenet_sizes=[256, 128],
        post_bank_size=8, post_bank_channel_size=128, post_maxpool_width=2, post_highway_depth=4, post_rnn_size=128,
        post_proj_sizes=[256, 80], post_proj_width=3, reduction_factor=5,
        min_tokens=30, min_iters=30, max_iters=200, skip_inadequate=False, griffin_lim_iters=60, power=1.5,
        recognition_loss_coeff=0.2, ignore_recognition_level=0,
    )
    # hparams.py:190-192 derived values
    hp.num_freq = int(hp.fft_size / 2 + 1)
    hp.frame_shift_ms = hp.hop_size * 1000.0 / hp.sample_rate
    hp.frame_length_ms = hp.win_size * 1000.0 / hp.sample_rate
    return hp


hparams = default_hparams()


def hparams_debug_string(hp=None):
    values = (hp or hparams).values()
    return 'Hyperparameters:\n' + '\n'.join('  %s: %s' % (name, values[name]) for name in sorted(values))


def save_hparams(model_dir, hp):
    """utils/__init__.py:143-154: params.json, indent 4, sorted keys."""
    path = os.path.join(model_dir, PARAMS_NAME)
    with open(path, 'w', encoding='utf-8') as f:
        json.dump(hp.values(), f, indent=4, sort_keys=True, ensure_ascii=False)
    return path


def load_json(path, encoding='euc-kr'):
    """utils/__init__.py:173-185 (tolerates trailing commas)."""
    with open(path, encoding=encoding) as f:
        content = f.read()
    content = re.sub(r",\s*}", "}", content)
    content = re.sub(r",\s*]", "]", content)
    return json.loads(content)


def load_hparams(hp, load_path, skip_list=()):
    """utils/__init__.py:156-172: override attribute by attribute; keys unknown to hparams are skipped."""
    new = load_json(os.path.join(load_path, PARAMS_NAME))
    keys = vars(hp).keys()
    for key, value in new.items():
        if key in skip_list or key not in keys:
            print("Skip {} because it not exists".format(key))
            continue
        if getattr(hp, key) != value:
            print("UPDATE {}: {} -> {}".format(key, getattr(hp, key), value))
            setattr(hp, key, value)
    return hp
