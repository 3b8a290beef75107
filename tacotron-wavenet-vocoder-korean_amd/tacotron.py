"""Host-side mirror of the reference's Tacotron inference surface (tacotron/tacotron.py `Tacotron.initialize(...,
rnn_decoder_test_mode=True)` and synthesizer.py `Synthesizer.load / synthesize`) over the HIP C-ABI.

Default hparams path (model_type 'deepvoice' with num_speakers > 1, or a single speaker; attention_type 'bah_mon_norm'); tokens in, mel /
linear / alignments out.  Text -> token ids (text/*, jamo) and Griffin-Lim are host DSP outside this path."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

BN_EPS = np.float32(1e-3)      # tf.layers.batch_normalization default epsilon


def _cbhg_specs(scope, cin, bank, bch, proj, pw, depth, rnn):
    s = []
    for k in range(1, bank + 1):
        p = "%s/conv_bank/conv1d_%d/" % (scope, k)
        s += [(p + "conv1d/kernel", (k, cin, bch)), (p + "conv1d/bias", (bch,)), (p + "batch_normalization", (4, bch))]
    c = bank * bch
    for i in range(2):
        p = "%s/proj_%d/" % (scope, i + 1)
        s += [(p + "conv1d/kernel", (pw, c, proj[i])), (p + "conv1d/bias", (proj[i],)), (p + "batch_normalization", (4, proj[i]))]
        c = proj[i]
    if proj[1] != rnn:
        s += [(scope + "/dense/kernel", (proj[1], rnn)), (scope + "/dense/bias", (rnn,))]
    for i in range(depth):
        p = "%s/highway_%d/" % (scope, i + 1)
        s += [(p + "H/kernel", (rnn, rnn)), (p + "H/bias", (rnn,)), (p + "T/kernel", (rnn, rnn)), (p + "T/bias", (rnn,))]
    for dr in ("fw", "bw"):
        p = "%s/bidirectional_rnn/%s/gru_cell/" % (scope, dr)
        s += [(p + "gates/kernel", (2 * rnn, 2 * rnn)), (p + "gates/bias", (2 * rnn,)),
              (p + "candidate/kernel", (2 * rnn, rnn)), (p + "candidate/bias", (rnn,))]
    return s


def tacotron_specs(hp, num_speakers, n_symbols=80):
    """checkpoint tensors (under 'model/inference/') in canonical blob order.  `batch_normalization` entries are (4, C):
    gamma, beta, moving_mean, moving_variance (four TF variables stacked)."""
    E, SE = hp.embedding_size, hp.speaker_embedding_size
    P0, P1 = hp.enc_prenet_sizes
    RN, A, AS, DR, M, R = hp.enc_rnn_size, hp.attention_size, hp.attention_state_size, hp.dec_rnn_size, hp.num_mels, hp.reduction_factor
    ENC = 2 * RN
    s = [("embedding", (n_symbols, E))]
    # tacotron.py:62-104: speaker embedding + the five deep_dense layers exist for num_speakers > 1 only; tf.layers.dense layers
    # are auto-named dense, dense_1, ... in creation order, so the linear-spectrogram layer is "dense" in a single-speaker graph
    dn = ([P1, 2 * RN, AS] + [DR] * hp.dec_layer_num) if num_speakers > 1 else []
    simple = num_speakers > 1 and getattr(hp, "model_type", "deepvoice") == "simple" and SE != 1
    SEc = SE if simple else 0
    if simple:
        # tacotron.py:85-90 model_type 'simple': the speaker embedding only -- it is concatenated to the decoder prenet's output
        # (rnn_wrappers.py:425-432) and to [output, attention] in front of the first projection (:455-463); no dense layers before "dense"
        s += [("speaker_embedding", (num_speakers, SE))]
        dn = []
    elif num_speakers > 1 and SE == 1:
        # tacotron.py:69-75: speaker_embedding_size == 1 -> the five speaker-dependent vectors are embedding tables of their own
        # (modules.py:10-12 get_embed, looked up by speaker id); no speaker embedding, no deep_dense layers (the linear layer is "dense")
        names = ["before_highway", "encoder_rnn_init_state", "attention_rnn_init_state"] + ["decoder_rnn_init_states%d" % (i + 1) for i in range(hp.dec_layer_num)]
        s += [(nm, (num_speakers, n)) for nm, n in zip(names, dn)]
        dn = []
    elif num_speakers > 1:
        s += [("speaker_embedding", (num_speakers, SE))]
    for i, n in enumerate(dn):
        nm = "dense" if i == 0 else "dense_%d" % i
        s += [(nm + "/kernel", (SE, n)), (nm + "/bias", (n,))]
    s += [("prenet/dense_1/kernel", (E, P0)), ("prenet/dense_1/bias", (P0,)), ("prenet/dense_2/kernel", (P0, P1)), ("prenet/dense_2/bias", (P1,))]
    s += _cbhg_specs("encoder_cbhg", P1, hp.enc_bank_size, hp.enc_bank_channel_size, tuple(hp.enc_proj_sizes), hp.enc_proj_width,
                     hp.enc_highway_depth, RN)
    s += [("memory_layer/kernel", (ENC, A)), ("decoder/bahdanau_monotonic_attention/query_layer/kernel", (AS, A)),
          ("decoder/bahdanau_monotonic_attention/attention_v", (A,)), ("decoder/bahdanau_monotonic_attention/attention_g", (1,)),
          ("decoder/bahdanau_monotonic_attention/attention_b", (A,)), ("decoder/bahdanau_monotonic_attention/attention_score_bias", (1,))]
    D0, D1 = hp.dec_prenet_sizes
    s += [("decoder/decoder_prenet/dense_1/kernel", (M, D0)), ("decoder/decoder_prenet/dense_1/bias", (D0,)),
          ("decoder/decoder_prenet/dense_2/kernel", (D0, D1)), ("decoder/decoder_prenet/dense_2/bias", (D1,))]
    ain = D1 + SEc + ENC
    p = "decoder/attention_wrapper/gru_cell/"
    s += [(p + "gates/kernel", (ain + AS, 2 * AS)), (p + "gates/bias", (2 * AS,)), (p + "candidate/kernel", (ain + AS, AS)), (p + "candidate/bias", (AS,))]
    s += [("decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/kernel", (AS + ENC + SEc, DR)),
          ("decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/bias", (DR,))]
    for i in range(hp.dec_layer_num):
        p = "decoder/output_projection_wrapper/multi_rnn_cell/cell_%d/gru_cell/" % (i + 1)
        s += [(p + "gates/kernel", (2 * DR, 2 * DR)), (p + "gates/bias", (2 * DR,)), (p + "candidate/kernel", (2 * DR, DR)), (p + "candidate/bias", (DR,))]
    s += [("decoder/output_projection_wrapper/kernel", (DR, M * R)), ("decoder/output_projection_wrapper/bias", (M * R,))]
    s += _cbhg_specs("post_cbhg", M, hp.post_bank_size, hp.post_bank_channel_size, tuple(hp.post_proj_sizes), hp.post_proj_width,
                     hp.post_highway_depth, hp.post_rnn_size)
    nm = "dense_%d" % len(dn) if dn else "dense"
    s += [(nm + "/kernel", (2 * hp.post_rnn_size, hp.num_freq)), (nm + "/bias", (hp.num_freq,))]
    return s


def bn_inference_vectors(bn):
    """tf.nn.batch_normalization at inference: inv = rsqrt(var + eps) * gamma ; y = x*inv + (beta - mean*inv)"""
    gamma, beta, mean, var = [np.asarray(v, np.float32) for v in bn]
    inv = (np.float32(1.0) / np.sqrt(var + BN_EPS)).astype(np.float32) * gamma
    shift = (beta - (mean * inv).astype(np.float32)).astype(np.float32)
    return inv.astype(np.float32), shift


def flatten(specs, tensors):
    parts = []
    for n, shp in specs:
        a = np.asarray(tensors[n], np.float32)
        if tuple(a.shape) != tuple(shp):
            raise ValueError("tensor %s has shape %s, expected %s" % (n, a.shape, shp))
        if n.endswith("batch_normalization"):
            parts += list(bn_inference_vectors(a))
        else:
            parts.append(a.reshape(-1))
    return np.concatenate(parts).astype(np.float32)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Tacotron(object):
    """tacotron/tacotron.py `Tacotron(hparams)` + `.initialize(inputs, input_lengths, num_speakers, speaker_id,
    rnn_decoder_test_mode=True)`: after `infer`, `.mel_outputs`, `.linear_outputs`, `.alignments` hold the same tensors."""

    def __init__(self, hparams, num_speakers=2, n_symbols=80, device="cuda:0"):
        hp = self._hparams = hparams
        if hp.attention_type != 'bah_mon_norm' or num_speakers < 1:
            raise NotImplementedError("built: attention_type 'bah_mon_norm' (hparams.py:145, the default)")
        if num_speakers > 1 and hp.model_type not in ('deepvoice', 'simple'):
            raise Exception(" [!] Unkown multi-speaker model type: {}".format(hp.model_type))       # tacotron.py:92
        self.num_speakers = num_speakers
        self.device = torch.device(device)
        self.specs = tacotron_specs(hp, num_speakers, n_symbols)
        d = _lib.TacoDims()
        d.n_symbols, d.embedding_size, d.num_speakers, d.speaker_embedding_size = n_symbols, hp.embedding_size, num_speakers, hp.speaker_embedding_size
        d.enc_prenet_sizes[0], d.enc_prenet_sizes[1] = hp.enc_prenet_sizes
        d.enc_bank_size, d.enc_bank_channel_size = hp.enc_bank_size, hp.enc_bank_channel_size
        d.enc_proj_sizes[0], d.enc_proj_sizes[1] = hp.enc_proj_sizes
        d.enc_proj_width, d.enc_highway_depth, d.enc_rnn_size = hp.enc_proj_width, hp.enc_highway_depth, hp.enc_rnn_size
        d.attention_size, d.attention_state_size = hp.attention_size, hp.attention_state_size
        d.dec_prenet_sizes[0], d.dec_prenet_sizes[1] = hp.dec_prenet_sizes
        d.dec_layer_num, d.dec_rnn_size = hp.dec_layer_num, hp.dec_rnn_size
        d.post_bank_size, d.post_bank_channel_size = hp.post_bank_size, hp.post_bank_channel_size
        d.post_proj_sizes[0], d.post_proj_sizes[1] = hp.post_proj_sizes
        d.post_proj_width, d.post_highway_depth, d.post_rnn_size = hp.post_proj_width, hp.post_highway_depth, hp.post_rnn_size
        d.num_mels, d.reduction_factor, d.num_freq, d.max_iters = hp.num_mels, hp.reduction_factor, hp.num_freq, hp.max_iters
        d.model_simple = 1 if (num_speakers > 1 and hp.model_type == 'simple') else 0
        self._dims = d
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.twv_tacotron_create(C.byref(d), C.byref(h)))
        self._h = h
        self._packed = None
        self._ws = None

    def set_option(self, name, value):
        """launch geometry (performance only, results are bit-identical): "decoder_groups" = workgroups per utterance in the decoder
        (0 auto, 1/2/4/8/16, -1 old kernel, 32 XCD-local kernel); "decoder_local" 1/0 = an utterance's workgroups on one XCD (default) or
        spread over the XCDs; "decoder_split_all" -1/0/1 = prenet and query layer split over the workgroups (-1: when local);
        "gemm_group", "gemm_valu", "gemm_timing", "highway_stack" (0: one launch per highway layer) """
        _lib.check(self._L.twv_tacotron_set_option(self._h, name.encode(), int(value)))

    def decoder_kernel_name(self, batch, t_in):
        """the decoder kernel infer() launches for this batch and input length with the options set (a measurement label)"""
        return self._L.twv_tacotron_decoder_kernel_name(self._h, int(batch), int(t_in)).decode()

    def gemm_stats(self):
        """after set_option("gemm_timing", 1): (useful FLOPs, summed kernel milliseconds, launches) of the dense contractions since then"""
        f, ms, n = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(self._L.twv_tacotron_gemm_stats(self._h, C.byref(f), C.byref(ms), C.byref(n)))
        return f.value, ms.value, n.value

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.twv_tacotron_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def load_weights(self, tensors):
        blob = flatten(self.specs, tensors)
        assert blob.size == self._L.twv_tacotron_blob_floats(self._h), (blob.size, self._L.twv_tacotron_blob_floats(self._h))
        with torch.cuda.device(self.device):
            dblob = torch.from_numpy(blob).to(self.device)
            self._packed = torch.empty(self._L.twv_tacotron_packed_bytes(self._h) // 4, dtype=torch.float32, device=self.device)
            _lib.check(self._L.twv_tacotron_pack(self._h, _ptr(dblob), _ptr(self._packed), _stream()))
            torch.cuda.current_stream().synchronize()

    def infer(self, inputs, input_lengths, speaker_id, want_linear=True, want_alignments=True):
        hp = self._hparams
        with torch.cuda.device(self.device):
            tok = torch.as_tensor(np.asarray(inputs, np.int32), device=self.device).contiguous()
            N, T = tok.shape
            ln = torch.as_tensor(np.asarray(input_lengths, np.int32), device=self.device).contiguous()
            sp = None                                        # synthesizer.py:150-151: no speaker_id feed for a single-speaker model
            if self.num_speakers > 1:
                sp = torch.as_tensor(np.asarray(speaker_id, np.int32), device=self.device).contiguous()
            need = self._L.twv_tacotron_workspace_bytes(self._h, N, T) // 4
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.float32, device=self.device)
            TO = hp.max_iters * hp.reduction_factor
            mel = torch.empty((N, TO, hp.num_mels), dtype=torch.float32, device=self.device)
            lin = torch.empty((N, TO, hp.num_freq), dtype=torch.float32, device=self.device) if want_linear else None
            al = torch.empty((N, T, hp.max_iters), dtype=torch.float32, device=self.device) if want_alignments else None
            status = torch.zeros(4, dtype=torch.int32, device=self.device)
            _lib.check(self._L.twv_tacotron_infer(self._h, _ptr(self._packed), _ptr(tok), _ptr(ln), _ptr(sp), N, T, _ptr(self._ws),
                                                  _ptr(mel), _ptr(lin), _ptr(al), _ptr(status), _stream()))
        self.inputs, self.input_lengths, self.speaker_id = tok, ln, sp
        self.mel_outputs, self.linear_outputs, self.alignments = mel, lin, al
        return mel, lin, al


def __getattr__(name):
    # the reference's Synthesizer (synthesizer.py) has its own module here too; kept importable from this one
    if name == "Synthesizer":
        from .synthesizer import Synthesizer
        return Synthesizer
    raise AttributeError(name)
