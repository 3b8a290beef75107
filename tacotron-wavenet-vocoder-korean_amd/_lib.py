"""ctypes binding of the C-ABI in include/twv_amd.h (the drop-in boundary).

The HIP library is the product: if libtwv_amd.so is missing or fails to load this module raises --
there is NO CPU / PyTorch fallback anywhere in this package.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TWV_AMD_LIB") or os.path.join(_HERE, "libtwv_amd.so")      # TWV_AMD_LIB: A/B-test another build of the library
CSRC = os.path.join(_HERE, "csrc")
MAX_LAYERS = 64
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


class Dims(C.Structure):
    """twv_wavenet_dims (include/twv_amd.h) = WaveNetModel.__init__ arguments (wavenet/model.py:8-10)."""
    _fields_ = [("n_layers", C.c_int32), ("dilations", C.c_int32 * MAX_LAYERS), ("residual_channels", C.c_int32),
                ("dilation_channels", C.c_int32), ("skip_channels", C.c_int32), ("quantization_channels", C.c_int32),
                ("out_channels", C.c_int32), ("scalar_input", C.c_int32), ("initial_filter_width", C.c_int32),
                ("use_biases", C.c_int32), ("gc_channels", C.c_int32), ("gc_cardinality", C.c_int32),
                ("lc_channels", C.c_int32), ("n_upsample", C.c_int32), ("upsample_factor", C.c_int32 * 4)]


GENERATION_SOURCES = ("twv_wavenet.hip", "twv_wavenet_xcd.hip", "twv_xcd.hpp", "twv_dpp.hpp", "twv_dev.hpp", "twv_categorical.hpp", "twv_math.hpp", "twv_layout.hpp")


def _code_only(text):
    """C / C++ source with comments removed and white space collapsed: what the compiler sees, give or take a string literal"""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return " ".join(text.split())


def _hash_files(files, code_only=False):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            data = fh.read()
        h.update(_code_only(data.decode("utf-8")).encode() if code_only else data)
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def source_hash():
    """sha256 over the library's sources (csrc/, include/twv_amd.h) and the compile flags: what the binary is stamped with
    (twv_version() ends in it) and what decides whether a build is stale."""
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h"))]
    files.append(os.path.join(_HERE, "..", "include", "twv_amd.h"))
    return _hash_files(files)


def generation_hash():
    """sha256 over the sources the two generation kernels are compiled from (and the flags): the key of profiles/traffic.json --
    counter measurements of the generation kernel stay valid while only the Tacotron / training sources change -- or only COMMENTS do
    (the hash is taken over the code with comments and white space removed: an edit of a comment does not orphan a measurement)."""
    return _hash_files([os.path.join(CSRC, f) for f in GENERATION_SOURCES], code_only=True)


TACOTRON_SOURCES = ("twv_tacotron.hip", "twv_dev.hpp", "twv_dpp.hpp", "twv_math.hpp", "twv_layout.hpp")


def tacotron_hash():
    """the same for the Tacotron kernels: key of the `tacotron` entries of profiles/traffic.json (scripts/pmc_to_tacotron_traffic.py)"""
    return _hash_files([os.path.join(CSRC, f) for f in TACOTRON_SOURCES], code_only=True)


TRAIN_SOURCES = ("twv_train.hip", "twv_dev.hpp", "twv_math.hpp", "twv_layout.hpp")      # twv_dev.hpp includes the other two headers


def train_hash():
    """the same for the training step: key of the `train` entries of profiles/traffic.json (scripts/train_traffic.sh)"""
    return _hash_files([os.path.join(CSRC, f) for f in TRAIN_SOURCES], code_only=True)


def build_stamp():
    """(extra hipcc flags, stamp) of the build the environment asks for.  TWV_EXTRA_HIPCC_FLAGS (a tuning aid, e.g. -DTWV_TRPROF) is part of
    the stamp, so a variant build never passes for the plain one and the next plain import rebuilds."""
    extra = os.environ.get("TWV_EXTRA_HIPCC_FLAGS", "").split()
    return extra, source_hash() + ("+" + " ".join(extra) if extra else "")


_toolchain = None


def toolchain_id():
    """what `hipcc --version` prints (part of every object's cache key: a ROCm upgrade must not link objects of the old compiler)"""
    global _toolchain
    if _toolchain is None:
        try:
            _toolchain = subprocess.check_output(["hipcc", "--version"], stderr=subprocess.STDOUT).decode("utf-8", "replace")
        except Exception as e:                           # noqa: BLE001 -- no compiler: the build below fails with its own message
            _toolchain = "hipcc unavailable: %r" % (e,)
    return _toolchain


def build(force=False, verbose=False):
    """hipcc cross-compiles the gfx950 library in-tree (works without a GPU).  The source hash is compiled into the binary and kept
    in a sidecar file: a library whose stamp differs from the tree's hash is rebuilt (file times say nothing after a checkout).
    Several processes may call this at once (torchrun ranks, pytest-xdist workers on a stale tree): the whole build runs under an
    exclusive file lock, and whoever comes second finds the stamp in place."""
    want = source_hash()
    extra, stamp_want = build_stamp()
    stamp = LIB_PATH + ".srchash"

    def fresh():
        return os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == stamp_want
    if not force and fresh():
        return LIB_PATH
    import fcntl
    os.makedirs(os.path.join(_HERE, "_objcache"), exist_ok=True)
    with open(os.path.join(_HERE, "_objcache", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():                    # another process built it while this one waited
                return LIB_PATH
            return _build_locked(force, verbose, want, extra, stamp_want, stamp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, want, extra, stamp_want, stamp):
    hip = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    # one hipcc per source file, side by side (the generation kernels alone take minutes), then one link.  Objects are cached per
    # file under _objcache/ (git- and gpurun-ignored), keyed by the file's text + every header + the flags: an edit of one kernel
    # file recompiles that file only.  The source-hash stamp goes into twv_ckpt.hip alone (it exports twv_version).
    from concurrent.futures import ThreadPoolExecutor
    base_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + extra
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hpp", ".h"))] + [os.path.join(_HERE, "..", "include", "twv_amd.h")]
    cache = os.path.join(_HERE, "_objcache")
    os.makedirs(cache, exist_ok=True)

    def compile_one(src):
        flags = list(base_flags)
        if os.path.basename(src) == "twv_ckpt.hip":
            flags.append('-DTWV_SRC_HASH="%s"' % want)
        import hashlib
        key = hashlib.sha256((_hash_files([src] + headers) + " ".join(flags) + toolchain_id()).encode()).hexdigest()[:20]
        obj = os.path.join(cache, "%s.%s.o" % (os.path.basename(src), key))
        if force or not os.path.exists(obj):
            for old in os.listdir(cache):                       # one object per source file is kept
                if old.startswith(os.path.basename(src) + "."):
                    os.remove(os.path.join(cache, old))
            tmp = os.path.join(cache, "tmp.%d.%s.o" % (os.getpid(), os.path.basename(src)))   # (outside the prefix the loop above evicts)
            cmd = ["hipcc"] + flags + ["-c", src, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, obj)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(hip), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, hip))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH, "-L/opt/rocm/lib", "-lrocblas", "-lhipfft", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(stamp_want + "\n")
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libtwv_amd.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and must be the FIRST HIP runtime in the process:
    # loaded after it, this library binds to the same copy; loaded before it, /opt/rocm's copy comes in, torch then adds
    # its own, and the second runtime in one process fails with "no ROCm-capable device is detected".
    try:
        import torch                                 # noqa: F401
    except ImportError:
        pass                                         # symbol checks / builds still work without torch
    L = C.CDLL(LIB_PATH)
    vp, ip, fp, dp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.twv_last_error.restype = C.c_char_p
    L.twv_version.restype = C.c_char_p
    L.twv_wavenet_create.argtypes = [C.POINTER(Dims), C.POINTER(C.c_void_p)]
    L.twv_wavenet_destroy.argtypes = [vp]; L.twv_wavenet_destroy.restype = None
    L.twv_wavenet_receptive_field.argtypes = [vp]
    L.twv_wavenet_hop_size.argtypes = [vp]
    for n in ("twv_wavenet_blob_floats", "twv_wavenet_packed_bytes"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = C.c_size_t
    L.twv_wavenet_state_bytes.argtypes = [vp, C.c_int]; L.twv_wavenet_state_bytes.restype = C.c_size_t
    L.twv_wavenet_cond_bytes.argtypes = [vp, C.c_int, C.c_int]; L.twv_wavenet_cond_bytes.restype = C.c_size_t
    L.twv_wavenet_pack.argtypes = [vp, fp, vp, vp]
    L.twv_wavenet_reset_state.argtypes = [vp, vp, C.c_int, vp]
    L.twv_wavenet_upsample.argtypes = [vp, vp, fp, C.c_int, C.c_int, fp, fp, vp]
    L.twv_wavenet_condition.argtypes = [vp, vp, fp, ip, C.c_int, C.c_int, vp, vp]
    L.twv_wavenet_fused_conditioning.argtypes = [vp, C.c_int]
    L.twv_wavenet_kernel_name.argtypes = [vp, C.c_int]; L.twv_wavenet_kernel_name.restype = C.c_char_p
    L.twv_wavenet_cond_bytes_mel.argtypes = [vp, C.c_int, C.c_int]; L.twv_wavenet_cond_bytes_mel.restype = C.c_size_t
    L.twv_wavenet_condition_mel.argtypes = [vp, vp, fp, ip, C.c_int, C.c_int, vp, vp]
    L.twv_wavenet_generate.argtypes = [vp, vp, vp, vp, vp, vp, C.c_double, C.c_int, C.c_int, vp, ip, fp, C.c_int, vp]
    L.twv_wavenet_prime.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, ip, vp]
    L.twv_wavenet_status.argtypes = [ip, vp]
    L.twv_wavenet_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.twv_wavenet_set_profile_buffer.argtypes = [vp, vp, C.c_int]
    L.twv_mu_law_encode.argtypes = [fp, C.c_int64, C.c_int, ip, vp]
    L.twv_mu_law_decode.argtypes = [ip, C.c_int64, C.c_int, fp, vp]
    L.twv_mu_law_expand.argtypes = [fp, C.c_int64, C.c_int, fp, vp]
    L.twv_griffin_lim_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.twv_griffin_lim_destroy.argtypes = [vp]; L.twv_griffin_lim_destroy.restype = None
    L.twv_griffin_lim_samples.argtypes = [vp]
    L.twv_griffin_lim_workspace_bytes.argtypes = [vp]; L.twv_griffin_lim_workspace_bytes.restype = C.c_size_t
    L.twv_inv_linear_spectrogram.argtypes = [vp, fp, fp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, vp, fp, vp]
    L.twv_wav_to_int16.argtypes = [fp, C.c_int, C.c_int64, vp, fp, vp]
    L.twv_eval_elementwise.argtypes = [C.c_int, fp, C.c_int64, fp, vp]
    L.twv_eval_elementwise64.argtypes = [C.c_int, dp, C.c_int64, dp, vp]
    L.twv_selftest.argtypes = [fp, vp]
    L.twv_sample_categorical.argtypes = [fp, C.c_int64, C.c_int, C.c_double, dp, ip, fp, vp]
    L.twv_debug_occupy.argtypes = [C.c_int, C.c_int, C.c_double, vp]
    L.twv_tacotron_create.argtypes = [C.POINTER(TacoDims), C.POINTER(C.c_void_p)]
    L.twv_tacotron_destroy.argtypes = [vp]; L.twv_tacotron_destroy.restype = None
    for n in ("twv_tacotron_blob_floats", "twv_tacotron_packed_bytes"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = C.c_size_t
    L.twv_tacotron_workspace_bytes.argtypes = [vp, C.c_int, C.c_int]; L.twv_tacotron_workspace_bytes.restype = C.c_size_t
    L.twv_tacotron_pack.argtypes = [vp, fp, vp, vp]
    L.twv_tacotron_infer.argtypes = [vp, vp, ip, ip, ip, C.c_int, C.c_int, vp, fp, fp, fp, ip, vp]
    L.twv_tacotron_set_profile_buffer.argtypes = [vp, vp]
    L.twv_tacotron_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.twv_tacotron_gemm_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.twv_tacotron_decoder_kernel_name.argtypes = [vp, C.c_int, C.c_int]; L.twv_tacotron_decoder_kernel_name.restype = C.c_char_p
    L.twv_wavenet_train_create.argtypes = [C.POINTER(Dims), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.twv_wavenet_train_destroy.argtypes = [vp]; L.twv_wavenet_train_destroy.restype = None
    for n in ("twv_wavenet_train_param_floats", "twv_wavenet_train_workspace_bytes"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = C.c_size_t
    L.twv_wavenet_train_output_width.argtypes = [vp]
    L.twv_wavenet_train_reset_workspace.argtypes = [vp]
    L.twv_wavenet_train_loss_grad.argtypes = [vp, fp, fp, fp, ip, vp, fp, fp, vp]
    L.twv_wavenet_train_l2.argtypes = [vp, fp, C.c_double, vp, fp, fp, vp]
    L.twv_clip_by_global_norm.argtypes = [fp, C.c_int64, C.c_double, C.c_double, vp, vp]
    L.twv_adam_ema_step.argtypes = [fp, fp, fp, fp, fp, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                                    C.c_double, C.c_double, vp]
    L.twv_crc32c.argtypes = [vp, C.c_size_t, C.c_uint32]; L.twv_crc32c.restype = C.c_uint32
    _lib = L
    return L


EXPORTS = ["twv_last_error", "twv_version", "twv_wavenet_create", "twv_wavenet_destroy", "twv_wavenet_receptive_field",
           "twv_wavenet_hop_size", "twv_wavenet_blob_floats", "twv_wavenet_packed_bytes", "twv_wavenet_state_bytes",
           "twv_wavenet_cond_bytes", "twv_wavenet_pack", "twv_wavenet_reset_state", "twv_wavenet_upsample",
           "twv_wavenet_condition", "twv_wavenet_fused_conditioning", "twv_wavenet_kernel_name", "twv_wavenet_cond_bytes_mel", "twv_wavenet_condition_mel", "twv_wavenet_generate", "twv_wavenet_prime", "twv_wavenet_status", "twv_wavenet_set_option", "twv_wavenet_set_profile_buffer",
           "twv_mu_law_encode", "twv_mu_law_decode", "twv_mu_law_expand", "twv_wav_to_int16", "twv_eval_elementwise",
           "twv_eval_elementwise64", "twv_sample_categorical", "twv_selftest", "twv_debug_occupy", "twv_tacotron_create", "twv_tacotron_destroy", "twv_tacotron_blob_floats",
           "twv_tacotron_packed_bytes", "twv_tacotron_workspace_bytes", "twv_tacotron_pack", "twv_tacotron_infer", "twv_tacotron_set_profile_buffer", "twv_tacotron_set_option", "twv_tacotron_gemm_stats", "twv_tacotron_decoder_kernel_name",
           "twv_wavenet_train_create", "twv_wavenet_train_destroy", "twv_wavenet_train_param_floats", "twv_wavenet_train_workspace_bytes",
           "twv_wavenet_train_output_width", "twv_wavenet_train_reset_workspace", "twv_wavenet_train_loss_grad", "twv_adam_ema_step", "twv_wavenet_train_l2",
           "twv_clip_by_global_norm", "twv_griffin_lim_create", "twv_griffin_lim_destroy", "twv_griffin_lim_samples",
           "twv_griffin_lim_workspace_bytes", "twv_inv_linear_spectrogram", "twv_crc32c"]


class TacoDims(C.Structure):
    """twv_tacotron_dims (include/twv_amd.h) = the Tacotron hyper-parameters of hparams.py:126-165"""
    _fields_ = [("n_symbols", C.c_int32), ("embedding_size", C.c_int32), ("num_speakers", C.c_int32), ("speaker_embedding_size", C.c_int32),
                ("enc_prenet_sizes", C.c_int32 * 2), ("enc_bank_size", C.c_int32), ("enc_bank_channel_size", C.c_int32),
                ("enc_proj_sizes", C.c_int32 * 2), ("enc_proj_width", C.c_int32), ("enc_highway_depth", C.c_int32), ("enc_rnn_size", C.c_int32),
                ("attention_size", C.c_int32), ("attention_state_size", C.c_int32),
                ("dec_prenet_sizes", C.c_int32 * 2), ("dec_layer_num", C.c_int32), ("dec_rnn_size", C.c_int32),
                ("post_bank_size", C.c_int32), ("post_bank_channel_size", C.c_int32), ("post_proj_sizes", C.c_int32 * 2),
                ("post_proj_width", C.c_int32), ("post_highway_depth", C.c_int32), ("post_rnn_size", C.c_int32),
                ("num_mels", C.c_int32), ("reduction_factor", C.c_int32), ("num_freq", C.c_int32), ("max_iters", C.c_int32),
                ("model_simple", C.c_int32)]


TWV_E_BUSY = 5          # include/twv_amd.h: a persistent kernel found the device occupied: nothing was done, retry later


class TwvError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise TwvError("twv_amd error %d: %s" % (rc, lib().twv_last_error().decode()))
