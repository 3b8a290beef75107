"""End-to-end text -> mel -> wave (BASELINE configs[4]): synthesizer.py's Tacotron pass chained into generate.py's WaveNet loop.

In the reference the hand-off is a file: synthesizer.py:279-280 saves the (optionally trimmed) `mel_outputs` as .npy and
generate.py:151-155 loads it as the local condition (sample_size = frames * hop_size).  Here the mel stays in HBM:
Tacotron.infer -> WaveNetModel.create_upsample -> WaveNetModel.generate, one utterance per batch lane; with several GPUs
the utterances are sharded over ranks (shard.shard_range), no collective on the data path."""
import numpy as np
import torch

from .shard import shard_range


def attention_trim_frames(alignment, sequence_length, reduction_factor):
    """synthesizer.py:232-256 (`attention_trim and end_of_sentence`): number of spectrogram frames to keep.
    alignment (T_in, T_dec) numpy; returns spec_end_idx = r * jdx + 3."""
    attention_argmax = alignment.argmax(0)
    end_idx = min(sequence_length - 1, int(attention_argmax.max()))
    max_counter = min(int((attention_argmax == end_idx).sum()), 5)
    end_idx_counter = 0
    jdx = 0
    for jdx, attend_idx in enumerate(attention_argmax):
        if len(attention_argmax) > jdx + 1:
            if attend_idx == end_idx:
                end_idx_counter += 1
            if attend_idx == end_idx and attention_argmax[jdx + 1] > end_idx:
                break
            if end_idx_counter >= max_counter:
                break
        else:
            break
    return reduction_factor * jdx + 3


def text_to_wave(synthesizer, vocoder, tokens, speaker_ids, uniforms, n_frames=None, first_input=None, temperature=1.0):
    """tokens: list of token-id lists (one per utterance; len == vocoder.batch_size), speaker_ids (B) are used both as
    Tacotron speaker ids and as the vocoder's global-condition ids (generate.py --gc_id).  n_frames: mel frames handed to the
    vocoder (default: all max_iters*r frames; the reference trims on the host, see attention_trim_frames).
    uniforms: (B, n_frames*hop, nr_mix+1) draws for the MoL sampler.  Returns dict(mel, alignments, audio)."""
    out = synthesizer.infer(tokens, speaker_ids=speaker_ids, want_linear=False)
    mel = out["mel"]                                             # (B, max_iters*r, num_mels) device tensor
    B = mel.shape[0]
    if B != vocoder.batch_size:
        raise ValueError("vocoder batch_size %d != %d utterances" % (vocoder.batch_size, B))
    if n_frames is not None:
        mel = mel[:, :n_frames].contiguous()
    vocoder.queue_initializer()                                  # generate.py:163
    up = vocoder.create_upsample(mel)                            # generate.py:154-155
    if first_input is None:
        first_input = np.zeros(B, np.float32)                    # generate.py:192 (scalar_input: silence seed)
    audio = vocoder.generate(up, np.asarray(speaker_ids, np.int32), first_input, uniforms, temperature=temperature)
    return {"mel": mel, "alignments": out["alignments"], "input_lengths": out["input_lengths"], "audio": audio}


def shard_utterances(n_utterances, world_size, rank):
    """contiguous shard of the utterance list for this rank (configs[4]: 8 utterances over 8 GPUs -> one each)"""
    return shard_range(n_utterances, world_size, rank)
