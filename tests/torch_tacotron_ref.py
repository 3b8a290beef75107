"""An INDEPENDENT float64 restatement of the reference's Tacotron inference graph in PyTorch -- the second opinion on
oracle/tacotron.c.  Test infrastructure only.

Written from the reference SOURCE (/root/reference/tacotron/tacotron.py:36-235, modules.py:10-96, rnn_wrappers.py:282-467,
helpers.py:10-41) and from TensorFlow 1.x's documented / published semantics of the library pieces the reference calls
(tf.layers.conv1d / max_pooling1d / batch_normalization, tf.contrib.rnn.GRUCell, tf.nn.bidirectional_dynamic_rnn,
tf.contrib.seq2seq.BahdanauMonotonicAttention + monotonic_attention(mode='parallel'), OutputProjectionWrapper,
ResidualWrapper, MultiRNNCell, dynamic_decode) -- NOT from oracle/tacotron.c, which was not consulted while writing this
file: the two restatements share only the checkpoint tensor names (the interface) and whatever both authors recall of
TensorFlow.  TensorFlow itself is absent from this image, so agreement of the two pins neither to TF; it does pin each
against a misreading the other does not share.

Everything runs in float64 with plain torch ops in whatever order torch picks: this is a tolerance checker (1e-5), the
bit-exact arithmetic contract lives in oracle/ and csrc/.
"""
import numpy as np
import torch

F64 = torch.float64
BN_EPSILON = 1e-3          # tf.layers.batch_normalization(epsilon=0.001) default


def _t(a):
    if torch.is_tensor(a):
        return a if a.dtype == F64 else a.to(F64)
    return torch.as_tensor(np.asarray(a), dtype=F64)


def _dense(x, w, name, activation=None):
    """tf.layers.dense: x @ kernel + bias"""
    y = x @ _t(w[name + "/kernel"])
    if name + "/bias" in w:
        y = y + _t(w[name + "/bias"])
    return activation(y) if activation is not None else y


def _batch_norm_inference(x, bn):
    """tf.layers.batch_normalization(training=False): (x - moving_mean) / sqrt(moving_variance + eps) * gamma + beta; the
    checkpoint entry is (4, C) = gamma, beta, moving_mean, moving_variance (tests' tensor convention)"""
    gamma, beta, mean, var = [_t(v) for v in bn]
    return (x - mean) / torch.sqrt(var + BN_EPSILON) * gamma + beta


def _conv1d_same(x, kernel, bias):
    """tf.layers.conv1d(padding='same', strides=1) on (N, T, Cin) with kernel (k, Cin, Cout).  TensorFlow's SAME padding puts
    floor((k-1)/2) zeros in front and the rest behind (the extra one goes to the END for even k): out[t] = sum_j K[j] x[t + j - left]"""
    k = kernel.shape[0]
    left = (k - 1) // 2
    right = (k - 1) - left
    N, T, Cin = x.shape
    xp = torch.cat([torch.zeros(N, left, Cin, dtype=F64), x, torch.zeros(N, right, Cin, dtype=F64)], dim=1)
    out = torch.zeros(N, T, kernel.shape[2], dtype=F64)
    for j in range(k):
        out = out + xp[:, j:j + T] @ kernel[j]
    return out + bias


def _conv1d_block(x, w, scope, activation):
    """modules.py:92-96: conv1d with the ACTIVATION INSIDE the conv layer, batch normalisation after it"""
    y = _conv1d_same(x, _t(w[scope + "/conv1d/kernel"]), _t(w[scope + "/conv1d/bias"]))
    if activation is not None:
        y = activation(y)
    return _batch_norm_inference(y, w[scope + "/batch_normalization"])


def _max_pool_same(x, width):
    """tf.layers.max_pooling1d(pool_size=width, strides=1, padding='same'): the window of output t covers inputs
    t - left .. t + right with the same left/right split as the convolution; padding never wins the max"""
    left = (width - 1) // 2
    right = (width - 1) - left
    N, T, Cc = x.shape
    neg = torch.full((N, 1, Cc), -float("inf"), dtype=F64)
    xp = torch.cat([neg.expand(N, left, Cc), x, neg.expand(N, right, Cc)], dim=1)
    out = xp[:, 0:T]
    for j in range(1, width):
        out = torch.maximum(out, xp[:, j:j + T])
    return out


def _gru_cell(x, h, w, scope):
    """tf.contrib.rnn.GRUCell.call: [r, u] = sigmoid([x, h] @ gates/kernel + gates/bias);
    c = tanh([x, r * h] @ candidate/kernel + candidate/bias); h' = u * h + (1 - u) * c"""
    gates = torch.sigmoid(torch.cat([x, h], dim=-1) @ _t(w[scope + "/gates/kernel"]) + _t(w[scope + "/gates/bias"]))
    n = h.shape[-1]
    r, u = gates[..., :n], gates[..., n:]
    c = torch.tanh(torch.cat([x, r * h], dim=-1) @ _t(w[scope + "/candidate/kernel"]) + _t(w[scope + "/candidate/bias"]))
    return u * h + (1.0 - u) * c


def _bidirectional_gru(x, lengths, w, scope, init_fw, init_bw):
    """tf.nn.bidirectional_dynamic_rnn(cell_fw, cell_bw, x, sequence_length=lengths): per example the forward cell runs over
    steps 0 .. len-1, the backward cell over len-1 .. 0 (array_ops.reverse_sequence); outputs past the length are zeros"""
    N, T, _ = x.shape
    n = w[scope + "/bidirectional_rnn/fw/gru_cell/candidate/bias"].shape[0]
    out = torch.zeros(N, T, 2 * n, dtype=F64)
    ln = torch.full((N,), T, dtype=torch.long) if lengths is None else torch.as_tensor(np.asarray(lengths), dtype=torch.long)
    # all examples step together; an example past its length keeps its state and emits zeros -- for the backward direction that
    # is exactly "start at step len-1" (its state does not move until t < len)
    h = init_fw if init_fw is not None else torch.zeros(N, n, dtype=F64)
    for t in range(T):
        live = (t < ln)[:, None]
        hn = _gru_cell(x[:, t], h, w, scope + "/bidirectional_rnn/fw/gru_cell")
        h = torch.where(live, hn, h)
        out[:, t, :n] = torch.where(live, hn, torch.zeros_like(hn))
    h = init_bw if init_bw is not None else torch.zeros(N, n, dtype=F64)
    for t in range(T - 1, -1, -1):
        live = (t < ln)[:, None]
        hn = _gru_cell(x[:, t], h, w, scope + "/bidirectional_rnn/bw/gru_cell")
        h = torch.where(live, hn, h)
        out[:, t, n:] = torch.where(live, hn, torch.zeros_like(hn))
    return out


def _cbhg(x, lengths, w, scope, bank_size, n_proj, depth, before_highway=None, rnn_init=None):
    """modules.py:25-74"""
    relu = torch.relu
    bank = torch.cat([_conv1d_block(x, w, "%s/conv_bank/conv1d_%d" % (scope, k), relu) for k in range(1, bank_size + 1)], dim=-1)
    y = _max_pool_same(bank, 2)                                     # maxpool_width = 2 (hparams.py)
    for i in range(n_proj):
        y = _conv1d_block(y, w, "%s/proj_%d" % (scope, i + 1), relu if i < n_proj - 1 else None)
    hw = y + x
    if before_highway is not None:
        hw = hw + before_highway[:, None, :]
    if scope + "/dense/kernel" in w:                                # modules.py:55-56 dimensionality mismatch
        hw = _dense(hw, w, scope + "/dense")
    for i in range(depth):
        p = "%s/highway_%d" % (scope, i + 1)
        H = _dense(hw, w, p + "/H", torch.relu)
        T_ = _dense(hw, w, p + "/T", torch.sigmoid)
        hw = H * T_ + hw * (1.0 - T_)
    fw = bw = None
    if rnn_init is not None:
        n = rnn_init.shape[1] // 2
        fw, bw = rnn_init[:, :n], rnn_init[:, n:]                  # tf.split(encoder_rnn_init_state, 2, 1)
    return _bidirectional_gru(hw, lengths, w, scope, fw, bw)


def _safe_cumprod_exclusive(x):
    """tf.contrib.seq2seq.safe_cumprod(x, axis=1, exclusive=True) = exp(cumsum(log(clip(x, tiny, 1)), exclusive=True))"""
    tiny = np.finfo(np.float32).tiny                              # the graph runs in float32: finfo(float32).tiny
    lg = torch.log(torch.clamp(x, tiny, 1.0))
    cs = torch.cumsum(lg, dim=1) - lg
    return torch.exp(cs)


def _monotonic_attention_parallel(p_choose, previous):
    """tf.contrib.seq2seq.monotonic_attention(mode='parallel')"""
    cp = _safe_cumprod_exclusive(1.0 - p_choose)
    return p_choose * cp * torch.cumsum(previous / torch.clamp(cp, 1e-10, 1.0), dim=1)


@torch.no_grad()
def infer(w, dims, tokens, lengths, speaker_ids):
    """tacotron.py:36-235 with rnn_decoder_test_mode=True, linear_targets=None (synthesizer.py:56): returns
    (mel (N, max_iters*r, num_mels), linear (N, max_iters*r, num_freq), alignments (N, T_in, max_iters)) as float64 numpy.
    `dims`: any object with n_speakers, enc_bank, post_bank, enc_hw_depth, post_hw_depth, dec_layers, num_mels, r, max_iters."""
    tokens = np.asarray(tokens); lengths = np.asarray(lengths)
    w = {k: _t(v) for k, v in w.items()}                             # float64 once, not per use
    N, T_in = tokens.shape
    multi = dims.n_speakers > 1
    # tacotron.py:51-60 embedding with the <PAD> row forced to zero
    table = _t(w["embedding"]).clone()
    table[0] = 0.0
    x = table[torch.as_tensor(tokens, dtype=torch.long)]
    before_highway = enc_init = att_init = None
    dec_init = [None] * dims.dec_layers
    embed_to_concat = None
    if multi and getattr(dims, "model_simple", 0) and "speaker_embedding" in w and "dense_1/kernel" not in w:
        # tacotron.py:85-90 model_type 'simple': before_highway and every initial state None; the speaker embedding itself goes into
        # DecoderPrenetWrapper (rnn_wrappers.py:425-432) and ConcatOutputAndAttentionWrapper (:455-463) as embed_to_concat
        embed_to_concat = w["speaker_embedding"][torch.as_tensor(np.asarray(speaker_ids), dtype=torch.long)]
        linear_name = "dense"
    elif multi and "speaker_embedding" not in w:
        # tacotron.py:69-75, speaker_embedding_size == 1: modules.py:10-12 get_embed -- five tables of their own, embedding_lookup by speaker id
        ids = torch.as_tensor(np.asarray(speaker_ids), dtype=torch.long)
        before_highway = w["before_highway"][ids]
        enc_init = w["encoder_rnn_init_state"][ids]
        att_init = w["attention_rnn_init_state"][ids]
        dec_init = [w["decoder_rnn_init_states%d" % (i + 1)][ids] for i in range(dims.dec_layers)]
        linear_name = "dense"                                        # no deep_dense layers were created before the linear one
    elif multi:
        # tacotron.py:63-86, model_type 'deepvoice', speaker_embedding_size != 1: softsign(dense(speaker_embed)); tf.layers.dense
        # layers are auto-named dense, dense_1, ... in creation order
        spk = _t(w["speaker_embedding"])[torch.as_tensor(np.asarray(speaker_ids), dtype=torch.long)]
        softsign = lambda v: v / (v.abs() + 1.0)
        names = ["dense"] + ["dense_%d" % i for i in range(1, 3 + dims.dec_layers)]
        before_highway = _dense(spk, w, names[0], softsign)
        enc_init = _dense(spk, w, names[1], softsign)
        att_init = _dense(spk, w, names[2], softsign)
        dec_init = [_dense(spk, w, names[3 + i], softsign) for i in range(dims.dec_layers)]
        linear_name = "dense_%d" % (3 + dims.dec_layers)
    else:
        linear_name = "dense"                                        # tacotron.py:97-104: nothing of the above exists
    # modules.py:15-23 prenet (dropout rate 0 at inference)
    h = _dense(x, w, "prenet/dense_1", torch.relu)
    h = _dense(h, w, "prenet/dense_2", torch.relu)
    enc = _cbhg(h, lengths, w, "encoder_cbhg", dims.enc_bank, 2, dims.enc_hw_depth, before_highway, enc_init)
    # BahdanauMonotonicAttention(normalize=True, memory_sequence_length=input_lengths): _prepare_memory zeroes the memory past
    # each length; keys = memory_layer(values) (no bias)
    mask = torch.as_tensor(np.arange(T_in)[None, :] < lengths[:, None])
    values = enc * mask[:, :, None].to(F64)
    keys = values @ _t(w["memory_layer/kernel"])
    ap = "decoder/bahdanau_monotonic_attention/"
    v, g, bvec = _t(w[ap + "attention_v"]), _t(w[ap + "attention_g"]), _t(w[ap + "attention_b"])
    score_bias = _t(w[ap + "attention_score_bias"])
    normed_v = g * v / torch.sqrt((v * v).sum())                   # _bahdanau_score(normalize=True): g * v * rsqrt(sum(v^2))
    wq = _t(w[ap + "query_layer/kernel"])
    R, M = dims.r, dims.num_mels
    att_h = att_init if att_init is not None else torch.zeros(N, wq.shape[0], dtype=F64)
    dec_h = [d if d is not None else None for d in dec_init]
    context = torch.zeros(N, enc.shape[2], dtype=F64)                # AttentionWrapper.zero_state: attention = zeros
    align = torch.zeros(N, T_in, dtype=F64); align[:, 0] = 1.0      # monotonic initial_alignments: one_hot(0)
    frame = torch.zeros(N, M, dtype=F64)                             # helpers.py _go_frames
    mel_steps, align_hist = [], []
    gp = "decoder/output_projection_wrapper/multi_rnn_cell/"
    for _step in range(dims.max_iters):
        # DecoderPrenetWrapper -> AttentionWrapper (rnn_wrappers.py:282-398)
        p = _dense(frame, w, "decoder/decoder_prenet/dense_1", torch.relu)
        p = _dense(p, w, "decoder/decoder_prenet/dense_2", torch.relu)
        if embed_to_concat is not None:
            p = torch.cat([p, embed_to_concat], dim=-1)              # rnn_wrappers.py:429-430 concat([prenet_out, embed_to_concat])
        att_h = _gru_cell(torch.cat([p, context], dim=-1), att_h, w, "decoder/attention_wrapper/gru_cell")
        q = att_h @ wq
        score = (normed_v * torch.tanh(keys + q[:, None, :] + bvec)).sum(dim=2) + score_bias
        score = torch.where(mask, score, torch.full_like(score, -float("inf")))       # _maybe_mask_score(-inf)
        align = _monotonic_attention_parallel(torch.sigmoid(score), align)            # sigmoid_noise = 0
        context = (align[:, None, :] @ values)[:, 0]
        align_hist.append(align)
        # ConcatOutputAndAttentionWrapper -> OutputProjectionWrapper(dec_rnn) -> ResidualWrapper(GRUCell) x dec_layers
        cat_out = [att_h, context] if embed_to_concat is None else [att_h, context, embed_to_concat]      # rnn_wrappers.py:458-462
        y = _dense(torch.cat(cat_out, dim=-1), w, gp + "cell_0/output_projection_wrapper")
        for i in range(dims.dec_layers):
            if dec_h[i] is None:
                dec_h[i] = torch.zeros(N, y.shape[1], dtype=F64)
            dec_h[i] = _gru_cell(y, dec_h[i], w, gp + "cell_%d/gru_cell" % (i + 1))
            y = y + dec_h[i]
        out = _dense(y, w, "decoder/output_projection_wrapper")
        mel_steps.append(out.reshape(N, R, M))
        frame = out[:, -M:]                                          # helpers.py:38-40 TacoTestHelper.next_inputs: the LAST of the r frames
    mel = torch.cat(mel_steps, dim=1)                                # tf.reshape(decoder_outputs, [N, -1, num_mels])
    post = _cbhg(mel, None, w, "post_cbhg", dims.post_bank, 2, dims.post_hw_depth)
    linear = _dense(post, w, linear_name)
    alignments = torch.stack(align_hist, dim=2)                      # transpose(alignment_history.stack(), [1, 2, 0])
    return mel.numpy(), linear.numpy(), alignments.numpy()
