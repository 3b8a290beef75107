"""Host-side mirror of the reference's WaveNetModel (wavenet/model.py) over the HIP C-ABI.

Same constructor arguments, `receptive_field`, `create_upsample`, `predict_proba_incremental`,
`queue_initializer`; plus `generate`, the generate.py:199-233 sample loop as ONE persistent kernel launch.
PyTorch is used for device memory and streams only.  No fallback: without the HIP library this raises."""
import ctypes as C
import time

import numpy as np
import torch

from . import _lib
from . import weights as W


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Upsampled(object):
    """What WaveNetModel.create_upsample returns under fused conditioning: the mel frames plus the shape of the upsampled tensor
    they stand for.  generate() takes it as `upsampled_local_condition` and upsamples inside its launch, ALWAYS FROM ROW 0: for
    chunked generation hand over `rows_from(start)` (or slices of `tensor()`), not the same object twice.  Anything else that is
    done to it -- indexing, `.to()`, `.float()`, `.cpu()`, `np.asarray`, `torch.cat`, arithmetic -- works on the materialised
    (B, T_mel*hop, lc) tensor of model.py:102-111, built on first use by the stand-alone upsampling kernel (same bits)."""

    def __init__(self, model, mel):
        self.model, self.mel = model, mel
        self.shape = (mel.shape[0], mel.shape[1] * model.hop_size, mel.shape[2])
        self._t = None

    def tensor(self):
        if self._t is None:
            self._t = self.model._upsample_now(self.mel)
        return self._t

    def rows_from(self, start):
        """the condition from upsampled row `start` on: still lazy (fused) when `start` is a whole number of hops, else the
        materialised rows"""
        hop = self.model.hop_size
        if start % hop == 0:
            return Upsampled(self.model, self.mel[:, start // hop:].contiguous())
        return self.tensor()[:, start:]

    def __getitem__(self, idx):
        return self.tensor()[idx]

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        a = self.tensor().cpu().numpy()
        return a if dtype is None else a.astype(dtype)

    def __getattr__(self, name):                 # everything a tensor has that this handle does not: on the materialised tensor
        if name.startswith("__") or name in ("model", "mel", "shape", "_t"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def conv(x):
            if isinstance(x, Upsampled):
                return x.tensor()
            if isinstance(x, (list, tuple)):
                return type(x)(conv(v) for v in x)
            return x
        return func(*conv(args), **{k: conv(v) for k, v in (kwargs or {}).items()})


def _forward_operators():
    import operator
    for name in ("add", "sub", "mul", "truediv", "matmul", "pow"):
        op = getattr(operator, name)
        setattr(Upsampled, "__%s__" % name, lambda self, other, op=op: op(self.tensor(), other.tensor() if isinstance(other, Upsampled) else other))
        setattr(Upsampled, "__r%s__" % name, lambda self, other, op=op: op(other, self.tensor()))
    Upsampled.__neg__ = lambda self: -self.tensor()


_forward_operators()


class WaveNetModel(object):
    def __init__(self, batch_size, dilations, filter_width, residual_channels, dilation_channels, skip_channels,
                 quantization_channels=2 ** 8, out_channels=30, use_biases=False, scalar_input=False,
                 initial_filter_width=32, global_condition_channels=None, global_condition_cardinality=None,
                 local_condition_channels=80, upsample_factor=None, train_mode=True, device="cuda:0"):
        if filter_width != 2:
            raise ValueError("filter_width must be 2 (hparams.py:59)")
        self.batch_size = batch_size
        self.dilations = list(dilations)
        self.filter_width = filter_width
        self.residual_channels = residual_channels
        self.dilation_channels = dilation_channels
        self.quantization_channels = quantization_channels
        self.use_biases = use_biases
        self.skip_channels = skip_channels
        self.scalar_input = scalar_input
        self.initial_filter_width = initial_filter_width
        self.global_condition_channels = global_condition_channels
        self.global_condition_cardinality = global_condition_cardinality
        self.local_condition_channels = local_condition_channels
        self.upsample_factor = list(upsample_factor) if upsample_factor else []
        self.train_mode = train_mode
        self.out_channels = out_channels
        self.receptive_field = WaveNetModel.calculate_receptive_field(filter_width, self.dilations, scalar_input,
                                                                      initial_filter_width)
        self.device = torch.device(device)
        self.specs = W.tensor_specs(len(self.dilations), residual_channels, dilation_channels, skip_channels,
                                    quantization_channels, out_channels, scalar_input, initial_filter_width, use_biases,
                                    global_condition_channels or 0, global_condition_cardinality or 0,
                                    local_condition_channels or 0, self.upsample_factor)
        d = _lib.Dims()
        d.n_layers = len(self.dilations)
        for i, v in enumerate(self.dilations):
            d.dilations[i] = int(v)
        d.residual_channels, d.dilation_channels, d.skip_channels = residual_channels, dilation_channels, skip_channels
        d.quantization_channels, d.out_channels = quantization_channels, out_channels
        d.scalar_input, d.initial_filter_width, d.use_biases = int(bool(scalar_input)), initial_filter_width, int(bool(use_biases))
        d.gc_channels = global_condition_channels or 0
        d.gc_cardinality = global_condition_cardinality or 0
        d.lc_channels = local_condition_channels or 0
        d.n_upsample = len(self.upsample_factor) if local_condition_channels else 0
        for i, v in enumerate(self.upsample_factor[:4]):
            d.upsample_factor[i] = int(v)
        self._dims = d
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.twv_wavenet_create(C.byref(d), C.byref(h)))
        self._h = h
        self.hop_size = self._L.twv_wavenet_hop_size(h)
        self._packed = None
        self._state = None
        self._status = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.twv_wavenet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def calculate_receptive_field(filter_width, dilations, scalar_input, initial_filter_width):
        """wavenet/model.py:31-39"""
        receptive_field = (filter_width - 1) * sum(dilations) + 1
        receptive_field += (initial_filter_width - 1) if scalar_input else (filter_width - 1)
        return receptive_field

    def set_option(self, name, value):
        """launch-geometry options of the C-ABI (twv_wavenet_set_option).  "xcd" / "xcd_many" / "groups" select the kernel and with it
        the SIZE and layout of the state buffer (the XCD kernels append their exchange area): the queues are re-created, i.e. reset as by
        queue_initializer (generate.py:163) -- set such options before priming / generating, not between chunks of one utterance."""
        _lib.check(self._L.twv_wavenet_set_option(self._h, name.encode(), int(value)))
        if name in ("xcd", "xcd_many", "groups") and self._state is not None:
            self.queue_initializer()

    # ---- weights (tf.train.Saver.restore of generate.py:157-161) ----
    def load_weights(self, tensors):
        blob = W.flatten(self.specs, tensors)
        assert blob.size == self._L.twv_wavenet_blob_floats(self._h), (blob.size, self._L.twv_wavenet_blob_floats(self._h))
        with torch.cuda.device(self.device):
            dblob = torch.from_numpy(blob).to(self.device)
            self._packed = torch.empty(self._L.twv_wavenet_packed_bytes(self._h) // 4, dtype=torch.float32, device=self.device)
            _lib.check(self._L.twv_wavenet_pack(self._h, _ptr(dblob), _ptr(self._packed), _stream()))
            torch.cuda.current_stream().synchronize()
        if self._state is None:
            self.queue_initializer()

    # ---- net.queue_initializer (model.py:64) ----
    def queue_initializer(self):
        with torch.cuda.device(self.device):
            n = self._L.twv_wavenet_state_bytes(self._h, self.batch_size) // 4
            if self._state is None or self._state.numel() != n:          # (the size depends on the kernel selection: see set_option)
                self._state = torch.empty(n, dtype=torch.float32, device=self.device)
                self._status = torch.zeros(4, dtype=torch.int32, device=self.device)
            _lib.check(self._L.twv_wavenet_reset_state(self._h, _ptr(self._state), self.batch_size, _stream()))

    # ---- model.py:102-111 ----
    def fused_conditioning(self):
        """True when create_upsample + the lc projections run inside the generation launch (XCD-per-stream kernel)."""
        with torch.cuda.device(self.device):             # the answer depends on THIS model's device (its CU count)
            return bool(self._L.twv_wavenet_fused_conditioning(self._h, self.batch_size))

    def kernel_name(self):
        """the generation kernel `generate` launches for this model, batch and options (twv_wavenet_kernel_name): measurement label"""
        with torch.cuda.device(self.device):
            return self._L.twv_wavenet_kernel_name(self._h, self.batch_size).decode()

    def create_upsample(self, local_condition_batch, materialize=None):
        """net.create_upsample(mel) (generate.py:200).  With fused conditioning the returned `Upsampled` only holds the mel frames:
        `generate` upsamples row by row inside its launch; `.tensor()` (or materialize=True) builds the (B, T_mel*hop, lc) tensor
        with the stand-alone kernel -- same bits either way."""
        mel = torch.as_tensor(local_condition_batch, dtype=torch.float32, device=self.device).contiguous()
        B, Tm, Lc = mel.shape
        assert Lc == self.local_condition_channels
        up = Upsampled(self, mel)
        if materialize is None:
            materialize = not self.fused_conditioning()
        return up.tensor() if materialize else up

    def _upsample_now(self, mel):
        B, Tm, Lc = mel.shape
        with torch.cuda.device(self.device):
            out = torch.empty((B, Tm * self.hop_size, Lc), dtype=torch.float32, device=self.device)
            scratch = torch.empty_like(out)
            _lib.check(self._L.twv_wavenet_upsample(self._h, _ptr(self._packed), _ptr(mel), B, Tm, _ptr(out), _ptr(scratch), _stream()))
        return out

    def _condition(self, upsampled, gc_ids, n_steps):
        B = self.batch_size
        with torch.cuda.device(self.device):
            gc = None
            if self.global_condition_channels:
                if self.global_condition_cardinality:
                    gc = torch.as_tensor(np.asarray(gc_ids, dtype=np.int32), device=self.device).contiguous()
                else:      # model.py:199-207: the global condition IS the embedding (B, gc_channels)
                    gc = torch.as_tensor(np.asarray(gc_ids, dtype=np.float32), device=self.device).reshape(B, -1).contiguous()
                    if gc.shape[1] != self.global_condition_channels:
                        raise ValueError('Shape of global_condition {} does not match global_condition_channels {}.'.format(tuple(gc.shape), self.global_condition_channels))
            if isinstance(upsampled, Upsampled) and self.fused_conditioning() and n_steps <= upsampled.shape[1]:
                mel = upsampled.mel
                cond = torch.empty(self._L.twv_wavenet_cond_bytes_mel(self._h, B, mel.shape[1]) // 4, dtype=torch.float32, device=self.device)
                _lib.check(self._L.twv_wavenet_condition_mel(self._h, _ptr(self._packed), _ptr(mel), _ptr(gc), B, mel.shape[1], _ptr(cond), _stream()))
                self._keep = (gc, mel)
                return cond
            if isinstance(upsampled, Upsampled):
                upsampled = upsampled.tensor()[:, :n_steps]
            cond = torch.empty(self._L.twv_wavenet_cond_bytes(self._h, B, n_steps) // 4, dtype=torch.float32, device=self.device)
            up = None
            if self.local_condition_channels:
                up = torch.as_tensor(upsampled, dtype=torch.float32, device=self.device).contiguous()
                assert tuple(up.shape) == (B, n_steps, self.local_condition_channels), (tuple(up.shape), (B, n_steps))
            _lib.check(self._L.twv_wavenet_condition(self._h, _ptr(self._packed), _ptr(up), _ptr(gc), B, n_steps, _ptr(cond), _stream()))
            self._keep = (gc, up)
        return cond

    # ---- the generic kernel's hoisted conditioning, bounded ----
    MAX_COND_BYTES = 1 << 30
    BUSY_RETRIES = 3                 # generate(check=True): relaunches after TWV_E_BUSY (device shared with another kernel at launch time)

    def _steps_per_call(self, n_steps):
        """The generic kernel reads a hoisted projection table cond[B][T][layers][64] (4 * B * T * layers * 64 bytes: 11.8 GB at BASELINE
        configs[1], linear in B*T).  Longer requests are cut into calls of at most MAX_COND_BYTES of it -- a whole number of hops, so the
        upsampled rows of a chunk come from whole mel frames; the state carries over between calls (as between the sess.run calls of
        generate.py:211), so the samples do not change.  The fused (XCD) path has no such table: one call."""
        if self.fused_conditioning():
            return n_steps
        # what one more step adds to the table of the kernel in use (0 on the XCD kernels without local conditioning; the generic
        # kernel's table holds a row per step even for a gc-only model)
        per_step = int(self._L.twv_wavenet_cond_bytes(self._h, self.batch_size, 2)) - int(self._L.twv_wavenet_cond_bytes(self._h, self.batch_size, 1))
        if per_step <= 0:
            return n_steps
        hop = max(1, self.hop_size)
        steps = max(hop, (self.MAX_COND_BYTES // per_step) // hop * hop)
        return min(n_steps, steps)

    def _lc_rows(self, upsampled, p, n):
        """rows [p, p+n) of the upsampled local condition; a lazy handle is upsampled chunk by chunk (a transposed conv with
        stride == kernel size in time: every output row depends on one mel frame only, model.py:102-111)"""
        if upsampled is None:
            return None
        if isinstance(upsampled, Upsampled):
            hop = self.hop_size
            f0, f1 = p // hop, (p + n + hop - 1) // hop
            rows = self._upsample_now(upsampled.mel[:, f0:f1].contiguous())
            return rows[:, p - f0 * hop:p - f0 * hop + n].contiguous()
        return upsampled[:, p:p + n]

    # ---- generate.py:199-233 as one persistent launch ----
    def generate(self, upsampled_local_condition, global_condition, first_input, uniforms, temperature=1.0,
                 debug_steps=0, check=True):
        """upsampled_local_condition (B,T,lc) | None; global_condition: (B) ids | None; first_input (B); uniforms
        (B,T,nr_mix+1) float32 (scalar_input) or (B,T) float64.  Returns (B,T) samples (and the debug dump)."""
        B = self.batch_size
        T_all = np.shape(uniforms)[1] if not torch.is_tensor(uniforms) else uniforms.shape[1]
        per_call = self._steps_per_call(T_all)
        if per_call < T_all and not debug_steps:
            outs, fi, p = [], first_input, 0
            while p < T_all:
                n = min(per_call, T_all - p)
                o = self.generate(self._lc_rows(upsampled_local_condition, p, n), global_condition, fi, uniforms[:, p:p + n],
                                  temperature=temperature, check=check)
                outs.append(o)
                fi = o[:, -1].cpu().numpy()                  # generate.py:204: the next window is the sample just appended
                p += n
            return torch.cat(outs, dim=1)
        with torch.cuda.device(self.device):
            if self.scalar_input:
                u = torch.as_tensor(uniforms, dtype=torch.float32, device=self.device).contiguous()
                T = u.shape[1]
                fi = torch.as_tensor(np.asarray(first_input, dtype=np.float32).reshape(B), device=self.device)
                out = torch.zeros((B, T), dtype=torch.float32, device=self.device)      # zeros: with check=False a launch that found the device busy writes nothing (status() tells)
            else:
                u = torch.as_tensor(uniforms, dtype=torch.float64, device=self.device).contiguous()
                T = u.shape[1]
                fi = torch.as_tensor(np.asarray(first_input, dtype=np.int32).reshape(B), device=self.device)
                out = torch.zeros((B, T), dtype=torch.int32, device=self.device)
            cond = self._condition(upsampled_local_condition, global_condition, T)
            dbg = None
            if debug_steps:
                opad = (((self.out_channels if self.scalar_input else self.quantization_channels) + 63) // 64) * 64
                dbg = torch.zeros((B, debug_steps, len(self.dilations) * 64 + opad), dtype=torch.float32, device=self.device)
            # TWV_E_BUSY (the persistent kernel's role workgroups were not all resident within ~50 ms: another kernel held CUs) means
            # NOTHING was done -- no sample written, state unchanged -- so the launch is simply repeated, a few times, with a pause
            for attempt in range(self.BUSY_RETRIES + 1):
                _lib.check(self._L.twv_wavenet_generate(self._h, _ptr(self._packed), _ptr(self._state), _ptr(cond), _ptr(fi), _ptr(u),
                                                        float(temperature), B, T, _ptr(out), _ptr(self._status), _ptr(dbg),
                                                        int(debug_steps), _stream()))
                if not check:
                    break
                rc = self._L.twv_wavenet_status(_ptr(self._status), _stream())
                if rc == _lib.TWV_E_BUSY and attempt < self.BUSY_RETRIES:
                    time.sleep(0.05 * (attempt + 1))
                    continue
                _lib.check(rc)
                break
        return (out, dbg) if debug_steps else out

    def status(self):
        """The status word of the launches issued so far on this model (generate / prime with check=False return without reading it):
        waits for the stream, raises TwvError for a watchdog abort, a launch that found the device busy (nothing was written: the
        zeros of `out` are not samples) or non-finite logits; returns 0 otherwise.  The firm way to end a check=False sequence."""
        with torch.cuda.device(self.device):
            rc = self._L.twv_wavenet_status(_ptr(self._status), _stream())
        _lib.check(rc)
        return 0

    # ---- generate.py:168-180 priming loop: feed the seed samples, discard the predictions ----
    def prime(self, inputs, upsampled_local_condition=None, global_condition=None, check=True):
        """inputs (B, n): n teacher-forced steps; upsampled_local_condition defaults to zeros like generate.py:180."""
        B = self.batch_size
        with torch.cuda.device(self.device):
            dt = torch.float32 if self.scalar_input else torch.int32
            x = torch.as_tensor(inputs if torch.is_tensor(inputs) else np.asarray(inputs), dtype=dt, device=self.device).reshape(B, -1).contiguous()
            n = x.shape[1]
            per_call = self._steps_per_call(n)
            if per_call < n:                                 # bounded conditioning table (see _steps_per_call)
                for p in range(0, n, per_call):
                    m_ = min(per_call, n - p)
                    self.prime(x[:, p:p + m_], self._lc_rows(upsampled_local_condition, p, m_), global_condition, check=check)
                return
            if upsampled_local_condition is None and self.local_condition_channels:
                upsampled_local_condition = torch.zeros((B, n, self.local_condition_channels), dtype=torch.float32, device=self.device)
            cond = self._condition(upsampled_local_condition, global_condition, n)
            _lib.check(self._L.twv_wavenet_prime(self._h, _ptr(self._packed), _ptr(self._state), _ptr(cond), _ptr(x), B, n,
                                                 _ptr(self._status), _stream()))
            if check:
                _lib.check(self._L.twv_wavenet_status(_ptr(self._status), _stream()))

    # ---- model.py:215-245: one step (a single sess.run of generate.py:211) ----
    def predict_proba_incremental(self, waveform, upsampled_local_condition=None, global_condition=None, uniforms=None,
                                  temperature=1.0):
        """One step of the incremental network (the queues advance by one sample, like one sess.run of generate.py:211).
        scalar_input (model.py:229-231): returns the sample drawn by sample_from_discretized_mix_logistic, (B, 1) float -- the two
        uniform draws must be injected.  One-hot model (model.py:241-243): returns tf.cast(softmax(float64(logits)), float32), (B, Q)
        probabilities, as the reference does (sampling is the caller's, generate.py:219-231); with `uniforms` given it returns the
        category drawn by that host rule instead (an extension: generate()'s one-step form)."""
        B = self.batch_size
        lc = None
        if upsampled_local_condition is not None:
            lc = torch.as_tensor(upsampled_local_condition, dtype=torch.float32, device=self.device).reshape(B, 1, -1)
        if uniforms is None and not self.scalar_input:
            from .ops import eval_elementwise
            Q = self.quantization_channels
            _ids, dump = self.generate(lc, global_condition, np.asarray(waveform).reshape(B), np.full((B, 1), 0.5), debug_steps=1)
            logits = dump[:, 0, len(self.dilations) * 64:len(self.dilations) * 64 + Q].to(torch.float64)
            e = eval_elementwise("exp64", logits - logits.max(dim=1, keepdim=True)[0], device=self.device)      # the kernel's own float64 exp
            return (e / e.sum(dim=1, keepdim=True)).to(torch.float32)
        if uniforms is None:
            raise ValueError("uniforms must be injected (the reference draws unseeded tf.random_uniform / np.random)")
        u = np.asarray(uniforms)
        u = u.reshape(B, 1, -1) if self.scalar_input else u.reshape(B, 1)
        return self.generate(lc, global_condition, np.asarray(waveform).reshape(B), u, temperature)

    # ---- model.py:247-346: the training-graph builders, eagerly (the reference builds graph nodes `net.loss` / `net.optimize` that
    # train_vocoder.py:163 runs once per step; here the calls ARE the step) ----
    def _trainer_for(self, sample_size, hparams=None):
        from .train import WaveNetTrainer, crop_length
        tr = getattr(self, "_trainer", None)
        if tr is None or tr.sample_size != crop_length(sample_size, self.hop_size):
            old = tr
            tr = WaveNetTrainer(self, hparams, sample_size=sample_size)
            if old is not None and old.params is not None:
                tr.params, tr.m, tr.v, tr.ema, tr.global_step = old.params, old.m, old.v, old.ema, old.global_step
            self._trainer = tr
        return tr

    def add_loss(self, input_batch, local_condition=None, global_condition_batch=None, l2_regularization_strength=None, name='wavenet'):
        """model.py:247-312: teacher-forced loss of one batch (B, T[, 1]) -> `self.loss` (device scalar); the gradients stay in the
        trainer for add_optimizer.  Weights: `net._trainer.load_weights(...)` / `init_weights()` once before the first call."""
        audio = torch.as_tensor(input_batch, dtype=torch.float32)
        if audio.dim() == 3:
            audio = audio[:, :, 0]
        tr = self._trainer_for(audio.shape[1])
        tr.l2 = float(l2_regularization_strength or 0.0)              # train_vocoder.py:118-119: 0 -> None
        if tr.params is None:
            tr.init_weights(seed=0)
        self.loss = tr.loss_and_gradients(audio, local_condition, global_condition_batch)
        return self.loss

    def add_optimizer(self, hparams, global_step=None):
        """model.py:314-346: exponential-decay learning rate, Adam, optional clip_by_global_norm, EMA of the variables; applies the
        gradients add_loss left (all-reduced over the process group when there is one).  Returns the learning rate used."""
        from .train import allreduce_sum_
        tr = self._trainer
        tr.lr0 = getattr(hparams, "wavenet_learning_rate", tr.lr0)
        tr.decay_steps = getattr(hparams, "wavenet_decay_steps", tr.decay_steps)
        tr.decay_rate = getattr(hparams, "wavenet_decay_rate", tr.decay_rate)
        tr.clip_gradients = bool(getattr(hparams, "wavenet_clip_gradients", tr.clip_gradients))
        if global_step is not None:
            tr.global_step = int(global_step)
        world = allreduce_sum_(tr.grads, tr.group)
        self.optimize = tr.apply_gradients(world)
        return self.optimize
