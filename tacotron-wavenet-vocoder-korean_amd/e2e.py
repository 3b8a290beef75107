"""End-to-end text -> mel -> wave (BASELINE configs[4]): synthesizer.py's Tacotron pass chained into generate.py's WaveNet loop.

In the reference the hand-off is a file: synthesizer.py:279-280 saves the (optionally trimmed) `mel_outputs` as .npy and
generate.py:151-155 loads it as the local condition (sample_size = frames * hop_size).  Here the mel stays in HBM:
Tacotron.infer -> WaveNetModel.create_upsample -> WaveNetModel.generate, one utterance per batch lane; with several GPUs
the utterances are sharded over ranks (shard.shard_range), no collective on the data path."""
import numpy as np
import torch

from .shard import shard_range


def attention_trim_frames(alignment, sequence_length, reduction_factor):
    """Number of spectrogram frames synthesizer.py:232-256 keeps when `attention_trim and end_of_sentence` (= r * jdx + 3).
    alignment (T_in, T_dec) numpy.  The rule, stated on arrays: `focus[j]` is the input position decoder step j attends to; the
    decode is cut at the first step j (not the last one) where either the focus sits on the final position `end` and moves past
    it at j + 1, or `end` has been in focus `quota` times (quota = how often it is in focus overall, at most 5)."""
    focus = np.asarray(alignment).argmax(axis=0)
    n = focus.size
    end = min(int(sequence_length) - 1, int(focus.max()))
    on_end = focus == end
    quota = min(int(on_end.sum()), 5)
    leaves_end = on_end[:-1] & (focus[1:] > end)
    quota_met = np.cumsum(on_end)[:-1] >= quota
    cut = np.flatnonzero(leaves_end | quota_met)
    jdx = int(cut[0]) if cut.size else n - 1
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
