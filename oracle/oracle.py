"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker (see oracle/README.md).  PARITY UNPINNED: the reference (Python on
TensorFlow 1.x) cannot be imported here and ships no golden vectors; this restates its source.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libtwv_oracle.so")
MAX_LAYERS = 64


class Dims(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("dilations", C.c_int * MAX_LAYERS), ("R", C.c_int), ("D", C.c_int),
                ("S", C.c_int), ("Q", C.c_int), ("O", C.c_int), ("scalar_input", C.c_int), ("ifw", C.c_int),
                ("use_bias", C.c_int), ("G", C.c_int), ("gc_card", C.c_int), ("L", C.c_int), ("n_up", C.c_int),
                ("up", C.c_int * 4)]


def make_dims(dilations, R=32, D=32, S=512, Q=256, out_channels=30, scalar_input=True, ifw=32, use_bias=True,
              G=32, gc_card=2, L=80, up=(5, 5, 12)):
    d = Dims()
    d.n_layers = len(dilations)
    for i, v in enumerate(dilations):
        d.dilations[i] = int(v)
    d.R, d.D, d.S, d.Q = R, D, S, Q
    d.scalar_input = 1 if scalar_input else 0
    d.O = out_channels if scalar_input else Q
    d.ifw = ifw if scalar_input else 2
    d.use_bias = 1 if use_bias else 0
    d.G = G or 0
    d.gc_card = (gc_card or 0) if G else 0          # 0 with G > 0: the caller supplies the embedding (model.py:199-207)
    d.L = L or 0
    d.n_up = len(up) if L else 0
    for i, v in enumerate(up if L else ()):
        d.up[i] = int(v)
    return d


def build(force=False):
    """compile oracle/*.c -> oracle/_build/libtwv_oracle.so (gcc)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))] + [os.path.join(_HERE, "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
        P = C.POINTER(Dims)
        L.twvo_blob_floats.restype = C.c_size_t; L.twvo_blob_floats.argtypes = [P]
        L.twvo_receptive_field.restype = C.c_int; L.twvo_receptive_field.argtypes = [P]
        L.twvo_hop.restype = C.c_int; L.twvo_hop.argtypes = [P]
        for n in ("twvo_tanh", "twvo_sigmoid", "twvo_exp", "twvo_log", "twvo_log1p"):
            getattr(L, n).restype = C.c_float; getattr(L, n).argtypes = [C.c_float]
        for n in ("twvo_exp64", "twvo_log64"):
            getattr(L, n).restype = C.c_double; getattr(L, n).argtypes = [C.c_double]
        L.twvo_cdot.restype = C.c_float; L.twvo_cdot.argtypes = [fp, C.c_int, fp, C.c_int]
        L.twvo_mu_law_encode.argtypes = [fp, C.c_int, C.c_int, ip]
        L.twvo_mu_law_decode.argtypes = [ip, C.c_int, C.c_int, fp]
        L.twvo_mu_law_expand.argtypes = [fp, C.c_int, C.c_int, fp]
        L.twvo_upsample.argtypes = [P, fp, fp, C.c_int, C.c_int, fp]
        L.twvo_state_new.restype = C.c_void_p; L.twvo_state_new.argtypes = [P, C.c_int]
        L.twvo_state_reset.argtypes = [C.c_void_p]
        L.twvo_state_free.argtypes = [C.c_void_p]
        L.twvo_step.argtypes = [P, fp, C.c_void_p, fp, ip, fp, ip, fp, fp, fp]
        L.twvo_sample_mol.restype = C.c_float; L.twvo_sample_mol.argtypes = [fp, C.c_int, fp]
        L.twvo_set_threads.argtypes = [C.c_int]; L.twvo_set_threads.restype = None
        L.twvo_max_threads.restype = C.c_int
        L.twvo_generate_mol.argtypes = [P, fp, C.c_void_p, fp, ip, fp, fp, C.c_int, C.c_int, fp]
        L.twvo_sample_categorical.restype = C.c_int
        L.twvo_sample_categorical.argtypes = [fp, C.c_int, C.c_double, C.c_double, fp]
        L.twvo_sample_categorical_sequential.restype = C.c_int
        L.twvo_sample_categorical_sequential.argtypes = [fp, C.c_int, C.c_double, C.c_double, fp]
        L.twvo_scan64.argtypes = [dp]; L.twvo_scan64.restype = None
        L.twvo_generate_mulaw.argtypes = [P, fp, C.c_void_p, fp, ip, ip, dp, C.c_double, C.c_int, C.c_int, ip]
        L.twvo_forward_full.argtypes = [P, fp, C.c_int, C.c_int, fp, ip, fp, C.c_int, ip, fp]
    return _lib


def _f(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _d(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _c32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _ci(a):
    if a is not None and np.asarray(a).dtype.kind == 'f':      # a (B, G) embedding instead of ids: same pointer, float payload
        return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def blob_floats(d):
    return int(lib().twvo_blob_floats(C.byref(d)))


def receptive_field(d):
    return int(lib().twvo_receptive_field(C.byref(d)))


def tensor_specs(d):
    """(name, shape) of every checkpoint tensor in canonical blob order (SURVEY.md 8a names)."""
    specs = []
    if d.scalar_input:
        specs.append(("wavenet/conv1d/kernel", (d.ifw, 1, d.R)))
    else:
        specs.append(("wavenet/conv1d/kernel", (2, d.Q, d.R)))
    if d.G and d.gc_card:
        specs.append(("wavenet/gc_embedding", (d.gc_card, d.G)))
    for i in range(d.n_layers):
        p = "wavenet/dilated_stack/layer%d/dilation_layer/" % i
        for nm in ("conv_filter", "conv_gate"):
            specs.append((p + nm + "/kernel", (2, d.R, d.D)))
            if d.use_bias:
                specs.append((p + nm + "/bias", (d.D,)))
        if d.G:
            specs.append((p + "gc_filter/kernel", (1, d.G, d.D)))
            specs.append((p + "gc_gate/kernel", (1, d.G, d.D)))
        if d.L:
            specs.append((p + "lc_filter/kernel", (1, d.L, d.D)))
            specs.append((p + "lc_gate/kernel", (1, d.L, d.D)))
        specs.append((p + "dense/kernel", (1, d.D, d.R)))
        if d.use_bias:
            specs.append((p + "dense/bias", (d.R,)))
        specs.append((p + "skip/kernel", (1, d.D, d.S)))
        if d.use_bias:
            specs.append((p + "skip/bias", (d.S,)))
    specs.append(("wavenet/conv1d_1/kernel", (1, d.S, d.S)))
    if d.use_bias:
        specs.append(("wavenet/conv1d_1/bias", (d.S,)))
    specs.append(("wavenet/conv1d_2/kernel", (1, d.S, d.O)))
    if d.use_bias:
        specs.append(("wavenet/conv1d_2/bias", (d.O,)))
    for i in range(d.n_up):
        specs.append(("wavenet/upsample%d/kernel" % i, (d.up[i], 2, 1, 1)))
    return specs


def blob_from_tensors(d, tensors):
    parts = [np.asarray(tensors[n], dtype=np.float32).reshape(-1) for n, shp in tensor_specs(d)]
    blob = np.concatenate(parts).astype(np.float32)
    assert blob.size == blob_floats(d), (blob.size, blob_floats(d))
    return blob


def random_tensors(d, seed=0, scale=0.05):
    rng = np.random.RandomState(seed)
    return {n: (rng.randn(*shp) * scale).astype(np.float32) for n, shp in tensor_specs(d)}


def elementwise(name, x):
    fn = getattr(lib(), "twvo_" + name)
    x = np.asarray(x)
    flat = x.reshape(-1)
    if name.endswith("64"):
        return np.array([fn(float(v)) for v in flat], dtype=np.float64).reshape(x.shape)
    return np.array([fn(float(np.float32(v))) for v in flat], dtype=np.float32).reshape(x.shape)


def mu_law_encode(audio, Q=256):
    a = _c32(audio).reshape(-1)
    out = np.empty(a.size, np.int32)
    lib().twvo_mu_law_encode(_f(a), a.size, Q, _i(out))
    return out.reshape(np.shape(audio))


def mu_law_decode(q, Q=256):
    a = _ci(q).reshape(-1)
    out = np.empty(a.size, np.float32)
    lib().twvo_mu_law_decode(_i(a), a.size, Q, _f(out))
    return out.reshape(np.shape(q))


def mu_law_expand(y, Q=256):
    a = _c32(y).reshape(-1)
    out = np.empty(a.size, np.float32)
    lib().twvo_mu_law_expand(_f(a), a.size, Q, _f(out))
    return out.reshape(np.shape(y))


def upsample(d, blob, mel):
    mel = _c32(mel)
    B, Tm, L = mel.shape
    hop = int(lib().twvo_hop(C.byref(d)))
    out = np.empty((B, Tm * hop, L), np.float32)
    lib().twvo_upsample(C.byref(d), _f(blob), _f(mel), B, Tm, _f(out))
    return out


class State:
    def __init__(self, d, B):
        self.d, self.B = d, B
        self.h = lib().twvo_state_new(C.byref(d), B)

    def reset(self):
        lib().twvo_state_reset(self.h)

    def __del__(self):
        try:
            lib().twvo_state_free(self.h)
        except Exception:
            pass


def step(d, blob, st, inp, lc=None, gc_ids=None, debug=False):
    B = st.B
    raw = np.empty((B, d.O), np.float32)
    lc = _c32(lc); gc = _ci(gc_ids)
    dz = np.empty((B, d.n_layers, d.D), np.float32) if debug else None
    dx = np.empty((B, d.n_layers, d.R), np.float32) if debug else None
    if d.scalar_input:
        a = _c32(inp).reshape(B)
        lib().twvo_step(C.byref(d), _f(blob), st.h, _f(a), None, _f(lc), _i(gc), _f(raw), _f(dz), _f(dx))
    else:
        a = _ci(inp).reshape(B)
        lib().twvo_step(C.byref(d), _f(blob), st.h, None, _i(a), _f(lc), _i(gc), _f(raw), _f(dz), _f(dx))
    return (raw, dz, dx) if debug else raw


def sample_mol(y, u):
    y = _c32(y); u = _c32(u)
    return float(lib().twvo_sample_mol(_f(y), y.size // 3, _f(u)))


def set_threads(n):
    """threads for generate_mol (one stream per thread); returns the host core count"""
    lib().twvo_set_threads(int(n))
    return int(lib().twvo_max_threads())


def generate_mol(d, blob, st, U, gc_ids, seed, u):
    U = _c32(U); u = _c32(u); seed = _c32(seed); gc = _ci(gc_ids)
    B, T = u.shape[0], u.shape[1]
    out = np.empty((B, T), np.float32)
    lib().twvo_generate_mol(C.byref(d), _f(blob), st.h, _f(U), _i(gc), _f(seed), _f(u), B, T, _f(out))
    return out


def sample_categorical(logits, temperature, u, sequential=False):
    """sequential=True: round 3's all-sequential form (kept for the comparison against numpy only)"""
    logits = _c32(logits)
    p = np.empty(logits.size, np.float32)
    fn = lib().twvo_sample_categorical_sequential if sequential else lib().twvo_sample_categorical
    k = fn(_f(logits), logits.size, float(temperature), float(u), _f(p))
    return int(k), p


def scan64(v):
    v = np.ascontiguousarray(v, dtype=np.float64).copy()
    assert v.size == 64
    lib().twvo_scan64(_d(v))
    return v


def generate_mulaw(d, blob, st, U, gc_ids, seed, u, temperature=1.0):
    U = _c32(U); seed = _ci(seed); gc = _ci(gc_ids)
    u = np.ascontiguousarray(u, dtype=np.float64)
    B, T = u.shape
    out = np.empty((B, T), np.int32)
    lib().twvo_generate_mulaw(C.byref(d), _f(blob), st.h, _f(U), _i(gc), _i(seed), _d(u), float(temperature), B, T, _i(out))
    return out


def forward_full(d, blob, inp, lc_up=None, gc_ids=None):
    B, Tin = np.shape(inp)
    rf = receptive_field(d)
    out = np.empty((B, Tin - rf + 1, d.O), np.float32)
    lc = _c32(lc_up); gc = _ci(gc_ids)
    Tlc = 0 if lc is None else lc.shape[1]
    if d.scalar_input:
        a = _c32(inp)
        lib().twvo_forward_full(C.byref(d), _f(blob), B, Tin, _f(a), None, _f(lc), Tlc, _i(gc), _f(out))
    else:
        a = _ci(inp)
        lib().twvo_forward_full(C.byref(d), _f(blob), B, Tin, None, _i(a), _f(lc), Tlc, _i(gc), _f(out))
    return out


# =====================================================================================================================
#  Tacotron text -> mel inference (oracle/tacotron.c)
# =====================================================================================================================
class TacoDims(C.Structure):
    _fields_ = [("n_symbols", C.c_int), ("emb", C.c_int), ("n_speakers", C.c_int), ("spk_emb", C.c_int),
                ("enc_prenet", C.c_int * 2), ("enc_bank", C.c_int), ("enc_bank_ch", C.c_int), ("enc_proj", C.c_int * 2),
                ("enc_proj_w", C.c_int), ("enc_hw_depth", C.c_int), ("enc_rnn", C.c_int), ("att", C.c_int), ("att_state", C.c_int),
                ("dec_prenet", C.c_int * 2), ("dec_layers", C.c_int), ("dec_rnn", C.c_int), ("post_bank", C.c_int),
                ("post_bank_ch", C.c_int), ("post_proj", C.c_int * 2), ("post_proj_w", C.c_int), ("post_hw_depth", C.c_int),
                ("post_rnn", C.c_int), ("num_mels", C.c_int), ("r", C.c_int), ("num_freq", C.c_int), ("max_iters", C.c_int),
                ("model_simple", C.c_int)]


def taco_dims(n_symbols=80, emb=256, n_speakers=2, spk_emb=16, enc_prenet=(256, 128), enc_bank=16, enc_bank_ch=128,
              enc_proj=(128, 128), enc_proj_w=3, enc_hw_depth=4, enc_rnn=128, att=256, att_state=256, dec_prenet=(256, 128),
              dec_layers=2, dec_rnn=256, post_bank=8, post_bank_ch=128, post_proj=(256, 80), post_proj_w=3, post_hw_depth=4,
              post_rnn=128, num_mels=80, r=5, num_freq=1025, max_iters=200, model_simple=False):
    """defaults = hparams.py:126-165; model_simple: hparams.model_type == 'simple' (tacotron.py:85-90), multi-speaker only"""
    d = TacoDims()
    d.model_simple = 1 if model_simple else 0
    d.n_symbols, d.emb, d.n_speakers, d.spk_emb = n_symbols, emb, n_speakers, spk_emb
    d.enc_prenet[0], d.enc_prenet[1] = enc_prenet
    d.enc_bank, d.enc_bank_ch, d.enc_proj_w, d.enc_hw_depth, d.enc_rnn = enc_bank, enc_bank_ch, enc_proj_w, enc_hw_depth, enc_rnn
    d.enc_proj[0], d.enc_proj[1] = enc_proj
    d.att, d.att_state = att, att_state
    d.dec_prenet[0], d.dec_prenet[1] = dec_prenet
    d.dec_layers, d.dec_rnn = dec_layers, dec_rnn
    d.post_bank, d.post_bank_ch, d.post_proj_w, d.post_hw_depth, d.post_rnn = post_bank, post_bank_ch, post_proj_w, post_hw_depth, post_rnn
    d.post_proj[0], d.post_proj[1] = post_proj
    d.num_mels, d.r, d.num_freq, d.max_iters = num_mels, r, num_freq, max_iters
    assert enc_proj[1] == enc_prenet[1] and post_proj[1] == num_mels        # the residual connections of modules.py:47-53
    assert att_state == dec_rnn and 2 * enc_rnn <= 1024
    return d


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


def taco_tensor_specs(d):
    """checkpoint tensors in canonical blob order.  Names follow the variable scopes of tacotron.py / modules.py / rnn_wrappers.py
    with TF's automatic numbering ([RECALLED-TF], unverified).  `batch_normalization` entries are (4, C): gamma, beta,
    moving_mean, moving_variance; the blob stores the derived (inv, shift) pair instead (see taco_blob)."""
    E, SE, P0, P1, RN, A, AS, DR, M, R = d.emb, d.spk_emb, d.enc_prenet[0], d.enc_prenet[1], d.enc_rnn, d.att, d.att_state, d.dec_rnn, d.num_mels, d.r
    ENC = 2 * RN
    s = [("embedding", (d.n_symbols, E))]
    dn = ([P1, 2 * RN, AS] + [DR] * d.dec_layers) if d.n_speakers > 1 else []      # tacotron.py:62-104
    simple = d.n_speakers > 1 and bool(d.model_simple) and SE != 1
    SEc = SE if simple else 0
    if simple:
        # tacotron.py:85-90 model_type 'simple': the speaker embedding only, concatenated inside the decoder (rnn_wrappers.py:425-432, 455-463)
        s += [("speaker_embedding", (d.n_speakers, SE))]
        dn = []
    elif d.n_speakers > 1 and SE == 1:
        # tacotron.py:69-75: speaker_embedding_size == 1 -> five embedding tables of their own (modules.py:10-12 get_embed), no dense layers
        names = ["before_highway", "encoder_rnn_init_state", "attention_rnn_init_state"] + ["decoder_rnn_init_states%d" % (i + 1) for i in range(d.dec_layers)]
        s += [(nm, (d.n_speakers, n)) for nm, n in zip(names, dn)]
        dn = []
    elif d.n_speakers > 1:
        s += [("speaker_embedding", (d.n_speakers, SE))]
    for i, n in enumerate(dn):
        nm = "dense" if i == 0 else "dense_%d" % i
        s += [(nm + "/kernel", (SE, n)), (nm + "/bias", (n,))]
    s += [("prenet/dense_1/kernel", (E, P0)), ("prenet/dense_1/bias", (P0,)), ("prenet/dense_2/kernel", (P0, P1)), ("prenet/dense_2/bias", (P1,))]
    s += _cbhg_specs("encoder_cbhg", P1, d.enc_bank, d.enc_bank_ch, tuple(d.enc_proj), d.enc_proj_w, d.enc_hw_depth, RN)
    s += [("memory_layer/kernel", (ENC, A)), ("decoder/bahdanau_monotonic_attention/query_layer/kernel", (AS, A)),
          ("decoder/bahdanau_monotonic_attention/attention_v", (A,)), ("decoder/bahdanau_monotonic_attention/attention_g", (1,)),
          ("decoder/bahdanau_monotonic_attention/attention_b", (A,)), ("decoder/bahdanau_monotonic_attention/attention_score_bias", (1,))]
    D0, D1 = d.dec_prenet[0], d.dec_prenet[1]
    s += [("decoder/decoder_prenet/dense_1/kernel", (M, D0)), ("decoder/decoder_prenet/dense_1/bias", (D0,)),
          ("decoder/decoder_prenet/dense_2/kernel", (D0, D1)), ("decoder/decoder_prenet/dense_2/bias", (D1,))]
    ain = D1 + SEc + ENC                                              # 'simple': [prenet_out, speaker embed, context]
    p = "decoder/attention_wrapper/gru_cell/"
    s += [(p + "gates/kernel", (ain + AS, 2 * AS)), (p + "gates/bias", (2 * AS,)), (p + "candidate/kernel", (ain + AS, AS)), (p + "candidate/bias", (AS,))]
    s += [("decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/kernel", (AS + ENC + SEc, DR)),
          ("decoder/output_projection_wrapper/multi_rnn_cell/cell_0/output_projection_wrapper/bias", (DR,))]
    for i in range(d.dec_layers):
        p = "decoder/output_projection_wrapper/multi_rnn_cell/cell_%d/gru_cell/" % (i + 1)
        s += [(p + "gates/kernel", (2 * DR, 2 * DR)), (p + "gates/bias", (2 * DR,)), (p + "candidate/kernel", (2 * DR, DR)), (p + "candidate/bias", (DR,))]
    s += [("decoder/output_projection_wrapper/kernel", (DR, M * R)), ("decoder/output_projection_wrapper/bias", (M * R,))]
    s += _cbhg_specs("post_cbhg", M, d.post_bank, d.post_bank_ch, tuple(d.post_proj), d.post_proj_w, d.post_hw_depth, d.post_rnn)
    nm = "dense_%d" % len(dn) if dn else "dense"
    s += [(nm + "/kernel", (2 * d.post_rnn, d.num_freq)), (nm + "/bias", (d.num_freq,))]
    return s


BN_EPS = np.float32(1e-3)      # tf.layers.batch_normalization default epsilon [RECALLED-TF]


def bn_inference_vectors(bn):
    """(4,C) gamma, beta, moving_mean, moving_variance -> (inv, shift) in float32, as tf.nn.batch_normalization does:
    inv = rsqrt(var + eps) * gamma ; y = x*inv + (beta - mean*inv)   [RECALLED-TF]"""
    gamma, beta, mean, var = [np.asarray(v, np.float32) for v in bn]
    inv = (np.float32(1.0) / np.sqrt(var + BN_EPS)).astype(np.float32) * gamma
    shift = (beta - (mean * inv).astype(np.float32)).astype(np.float32)
    return inv.astype(np.float32), shift


def taco_random_tensors(d, seed=0, scale=0.08):
    rng = np.random.RandomState(seed)
    t = {}
    for n, shp in taco_tensor_specs(d):
        if n.endswith("batch_normalization"):
            C_ = shp[1]
            t[n] = np.stack([1 + 0.1 * rng.randn(C_), 0.1 * rng.randn(C_), 0.1 * rng.randn(C_), 1 + 0.2 * rng.rand(C_)]).astype(np.float32)
        elif n.endswith("gates/bias"):
            t[n] = (1.0 + 0.05 * rng.randn(*shp)).astype(np.float32)          # GRUCell gate bias init 1.0
        elif n.endswith("T/bias"):
            t[n] = (-1.0 + 0.05 * rng.randn(*shp)).astype(np.float32)         # modules.py:87 highway T bias init -1
        elif n.endswith("attention_g"):
            t[n] = np.array([np.sqrt(1.0 / d.att)], np.float32)
        elif n.endswith("attention_score_bias"):
            t[n] = np.array([0.0], np.float32)
        elif n in ("embedding", "speaker_embedding"):
            t[n] = (rng.randn(*shp) * 0.5).astype(np.float32)                 # truncated_normal(stddev=0.5), tacotron.py:51,67
        else:
            fan_in = int(np.prod(shp[:-1])) if len(shp) > 1 else 1
            t[n] = (rng.randn(*shp) * (scale if len(shp) == 1 else min(1.0, 1.2 / np.sqrt(fan_in)))).astype(np.float32)
    return t


def taco_blob(d, tensors):
    parts = []
    for n, shp in taco_tensor_specs(d):
        a = np.asarray(tensors[n], np.float32)
        assert tuple(a.shape) == tuple(shp), (n, a.shape, shp)
        if n.endswith("batch_normalization"):
            inv, shift = bn_inference_vectors(a)
            parts += [inv, shift]
        else:
            parts.append(a.reshape(-1))
    blob = np.concatenate(parts).astype(np.float32)
    L = lib()
    L.twvo_taco_blob_floats.restype = C.c_size_t
    L.twvo_taco_blob_floats.argtypes = [C.POINTER(TacoDims)]
    assert blob.size == L.twvo_taco_blob_floats(C.byref(d)), (blob.size, L.twvo_taco_blob_floats(C.byref(d)))
    return blob


def taco_infer(d, blob, tokens, lengths, speaker_ids, want_linear=True, want_align=True):
    tokens = _ci(tokens); lengths = _ci(lengths)
    speaker_ids = _ci(speaker_ids) if d.n_speakers > 1 else None
    N, T = tokens.shape
    TO = d.max_iters * d.r
    mel = np.empty((N, TO, d.num_mels), np.float32)
    lin = np.empty((N, TO, d.num_freq), np.float32) if want_linear else None
    al = np.empty((N, T, d.max_iters), np.float32) if want_align else None
    L = lib()
    L.twvo_taco_infer.argtypes = [C.POINTER(TacoDims), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.twvo_taco_infer(C.byref(d), _f(blob), _i(tokens), _i(lengths), _i(speaker_ids), N, T, _f(mel), _f(lin), _f(al))
    return mel, lin, al
