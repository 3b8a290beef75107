"""synthesizer.py of the reference (synthesizer.py:29-388) on the MI355X path: same class, method signatures, output files and
CLI flags; `sess.run` of the Tacotron graph is ONE twv_tacotron_infer call and `inv_linear_spectrogram` (utils/audio.py:77-92,
Griffin-Lim) runs on the GPU.

    synthesizer = Synthesizer()
    synthesizer.load(load_path, num_speakers, checkpoint_step)                       # synthesizer.py:34-70
    synthesizer.synthesize(texts=[text], base_path=sample_path, speaker_ids=[0], attention_trim=True)   # :72-199

Per utterance `plot_graph_and_save_audio` (synthesizer.py:202-287) trims the spectrograms by the attention rule, inverts the
linear spectrogram and writes `<base_path>/<time>.wav` plus the mel hand-off file `<...>.npy` that generate.py --mel reads
(synthesizer.py:279-280 -> generate.py:151); without base_path / paths it returns the wav bytes.

Out of scope here, as in SURVEY.md section 8 (they raise instead of silently doing something else): the text frontend
(`texts` need a `text_to_sequence` callable -- jamo / g2p are not part of the hot path; pass `tokens`), the alignment plots
(.png), the manual-attention modes and short_concat (dead / broken in the reference), librosa_trim."""
import argparse
import io
import os
from datetime import datetime
from glob import glob

import numpy as np

from .hparams import hparams, load_hparams


def get_time():
    return datetime.now().strftime("%Y-%m-%d_%H-%M-%S")


def add_postfix(path, postfix):
    base, ext = path.rsplit('.', 1)
    return "{}.{}.{}".format(base, postfix, ext)


def str2bool(v):
    return str(v).lower() in ('true', '1')


def get_most_recent_checkpoint(checkpoint_dir, checkpoint_step=None):
    """synthesizer.py:289-299: the bundle prefix `model.ckpt-<N>` of a directory -- the step asked for, else the largest N that has
    a data shard on disk"""
    step = checkpoint_step
    if step is None:
        steps = []
        for shard in glob(os.path.join(checkpoint_dir, "*.ckpt-*.data-*")):
            tail = os.path.basename(shard).split(".ckpt-", 1)[1]           # "<N>.data-00000-of-00001"
            steps.append(int(tail.split(".", 1)[0]))
        if not steps:
            raise FileNotFoundError("no model.ckpt-* bundle in %s" % checkpoint_dir)
        step = max(steps)
    prefix = os.path.join(checkpoint_dir, "model.ckpt-%d" % int(step))
    print(" [*] Found latest checkpoint: %s" % prefix)
    return prefix


def _prepare_inputs(inputs):
    """datafeeder_tacotron.py:288-290: right-pad the token sequences with 0 to the longest"""
    max_len = max(len(x) for x in inputs)
    return np.stack([np.pad(np.asarray(x, np.int32), (0, max_len - len(x)), mode='constant') for x in inputs])


class Synthesizer(object):
    text_to_sequence = None        # optional callable str -> list of token ids (the reference's text frontend is not rebuilt)

    def close(self):
        self.model = None

    def load(self, checkpoint_path, num_speakers=2, checkpoint_step=None, model_name='tacotron', hparams=None, device="cuda:0"):
        """synthesizer.py:34-70.  checkpoint_path: a logdir (most recent `model.ckpt-N`, or checkpoint_step) or one bundle prefix,
        restored by variable name; also a dict / .npz of numpy arrays keyed by the same variable names (tests, tools)."""
        from .tacotron import Tacotron
        from .hparams import hparams as default_hp
        self.num_speakers = num_speakers
        hp = hparams or default_hp
        if isinstance(checkpoint_path, str) and not checkpoint_path.endswith(".npz"):
            if os.path.isdir(checkpoint_path):
                load_path = checkpoint_path
                checkpoint_path = get_most_recent_checkpoint(checkpoint_path, checkpoint_step)
            else:
                load_path = os.path.dirname(checkpoint_path)
            if os.path.exists(os.path.join(load_path, "params.json")):
                load_hparams(hp, load_path)
        print('Constructing model: %s' % model_name)
        self.hparams = hp
        self.model = Tacotron(hp, num_speakers, device=device)
        if isinstance(checkpoint_path, str) and not checkpoint_path.endswith(".npz"):
            from . import checkpoint as ckpt
            print('Loading checkpoint: %s' % checkpoint_path)
            # by variable name (synthesizer.py:69-70 Saver.restore); names TensorFlow auto-generated (dense_N, ...) may be shifted in a
            # checkpoint written by a differently-built graph: checkpoint.restore_variables falls back to (scope, shape, creation order)
            tensors = ckpt.tacotron_tensors(ckpt.restore_variables(ckpt.resolve(checkpoint_path), ckpt.tacotron_variable_specs(self.model.specs), verify=True),
                                            self.model.specs)
        else:
            tensors = dict(np.load(checkpoint_path)) if isinstance(checkpoint_path, str) else checkpoint_path
        self.model.load_weights(tensors)

    # ---- one sess.run(fetches) of synthesizer.py:160 ----
    def infer(self, tokens, speaker_ids=None, want_linear=True):
        seqs = [np.asarray(s, np.int32) for s in tokens]
        sequences = _prepare_inputs(seqs)
        input_lengths = [int(np.argmax(a == 1)) + 1 for a in sequences]                  # synthesizer.py:126
        if speaker_ids is None:
            speaker_ids = np.zeros(len(seqs), np.int32)                                  # synthesizer.py:49-50 default
        mel, lin, al = self.model.infer(sequences, input_lengths, speaker_ids, want_linear=want_linear)
        return {"mel": mel, "linear": lin, "alignments": al, "input_lengths": input_lengths, "sequences": sequences}

    def synthesize(self, texts=None, tokens=None, base_path=None, paths=None, speaker_ids=None, start_of_sentence=None,
                   end_of_sentence=True, pre_word_num=0, post_word_num=0, pre_surplus_idx=0, post_surplus_idx=1,
                   use_short_concat=False, manual_attention_mode=0, base_alignment_path=None, librosa_trim=False,
                   attention_trim=True, isKorean=True, seed=None):
        """synthesizer.py:72-199.  Returns one result per utterance: True when files were written, else the wav bytes."""
        if manual_attention_mode or base_alignment_path is not None:
            raise ValueError("manual attention (synthesizer.py:137-196) is out of scope of this path (SURVEY.md section 8)")
        if use_short_concat or librosa_trim:
            raise ValueError("short_concat / librosa_trim (host DSP on text-frontend data) are out of scope of this path")
        if type(texts) == str:
            texts = [texts]
        if texts is not None and tokens is None:
            if self.text_to_sequence is None:
                raise ValueError("texts need Synthesizer.text_to_sequence (the reference's text frontend is not part of this path); "
                                 "pass tokens=[[ids..., 1], ...]")
            tokens = [self.text_to_sequence(text) for text in texts]
        if tokens is None:
            raise ValueError("texts or tokens required")
        out = self.infer(tokens, speaker_ids, want_linear=True)
        sequences = out["sequences"]
        if paths is None:
            paths = [None] * len(sequences)
        if texts is None:
            texts = [None] * len(sequences)
        time_str = get_time()
        wavs = out["linear"].cpu().numpy()
        alignments = out["alignments"].cpu().numpy()
        mels = out["mel"].cpu().numpy()
        results = []
        for item in enumerate(zip(wavs, alignments, paths, texts, sequences, mels)):
            results.append(plot_graph_and_save_audio(item, base_path=base_path, start_of_sentence=start_of_sentence,
                                                     end_of_sentence=end_of_sentence, attention_trim=attention_trim, time_str=time_str,
                                                     isKorean=isKorean, hparams=self.hparams, seed=seed))
        return results


def plot_graph_and_save_audio(args, base_path=None, start_of_sentence=None, end_of_sentence=None, use_manual_attention=False,
                              save_alignment=False, attention_trim=False, time_str=None, isKorean=True, hparams=hparams, seed=None):
    """synthesizer.py:202-287 without the .png: attention trim -> Griffin-Lim (GPU) -> `<path>.wav` + the mel `<path>.npy`."""
    from .audio import inv_linear_spectrogram
    from .e2e import attention_trim_frames
    from .ops import wav_to_int16
    from scipy.io import wavfile
    idx, (wav, alignment, path, text, sequence, mel) = args
    if base_path:
        plot_path = "{}/{}.{}.png".format(base_path, time_str or get_time(), idx)     # one name per utterance of the batch
    elif path:
        plot_path = path.rsplit('.', 1)[0] + ".png"
    else:
        plot_path = None
    if use_manual_attention and plot_path:
        plot_path = add_postfix(plot_path, "manual")
    if attention_trim and end_of_sentence:
        spec_end_idx = attention_trim_frames(alignment, len(sequence), hparams.reduction_factor)     # synthesizer.py:232-256
        wav = wav[:spec_end_idx]
        mel = mel[:spec_end_idx]
    audio_out = inv_linear_spectrogram(wav[None], hparams, seed=seed)[0]          # synthesizer.py:258 inv_linear_spectrogram(wav.T)
    if save_alignment and base_path:
        np.save("{}/{}.npy".format(base_path, idx), alignment, allow_pickle=False)
    pcm = wav_to_int16(audio_out[None]).cpu().numpy().reshape(-1)                 # utils/audio.py:14-17 save_wav
    if path or base_path:
        current_path = add_postfix(path, idx) if path else plot_path.replace(".png", ".wav")
        os.makedirs(os.path.dirname(os.path.abspath(current_path)), exist_ok=True)
        wavfile.write(current_path, hparams.sample_rate, pcm)
        np.save(current_path.replace(".wav", ".npy"), mel)                         # synthesizer.py:279-280: the vocoder's --mel input
        return True
    io_out = io.BytesIO()
    wavfile.write(io_out, hparams.sample_rate, pcm)
    return io_out.getvalue()


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--load_path', required=True)
    parser.add_argument('--sample_path', default="logdir-tacotron/generate")
    parser.add_argument('--text', default=None)
    parser.add_argument('--tokens', default=None, help='comma-separated token ids ending in 1 (EOS): the text frontend is not part of this path')
    parser.add_argument('--num_speakers', default=1, type=int)
    parser.add_argument('--speaker_id', default=0, type=int)
    parser.add_argument('--checkpoint_step', default=None, type=int)
    parser.add_argument('--is_korean', default=True, type=str2bool)
    parser.add_argument('--base_alignment_path', default=None)
    parser.add_argument('--seed', default=None, type=int, help='seed of the Griffin-Lim initial phases (extension; the reference is unseeded)')
    config = parser.parse_args(argv)
    if config.text is None and config.tokens is None:
        parser.error("--text or --tokens required")
    os.makedirs(config.sample_path, exist_ok=True)
    synthesizer = Synthesizer()
    synthesizer.load(config.load_path, config.num_speakers, config.checkpoint_step)
    tokens = [[int(t) for t in config.tokens.split(",")]] if config.tokens else None
    return synthesizer.synthesize(texts=[config.text] if config.text else None, tokens=tokens, base_path=config.sample_path,
                                  speaker_ids=[config.speaker_id], attention_trim=True, base_alignment_path=config.base_alignment_path,
                                  isKorean=config.is_korean, seed=config.seed)[0]


if __name__ == "__main__":
    main()
