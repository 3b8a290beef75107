"""Host-side mirror of the WaveNet training step (train_vocoder.py:86-182) over the HIP C-ABI.

    net = WaveNetModel(batch_size=hparams.wavenet_batch_size, ..., train_mode=True)   # train_vocoder.py:100-115
    trainer = WaveNetTrainer(net, hparams, sample_size=7800)
    trainer.load_weights(tensors)                       # or trainer.init_weights(seed)
    loss = trainer.step(audio, local_condition, gc_ids) # == sess.run([net.loss, net.optimize]), train_vocoder.py:163/174

`add_loss` (wavenet/model.py:247-312) and `add_optimizer` (model.py:314-346) run as twv_wavenet_train_loss_grad and
twv_adam_ema_step.  The reference trains on one device; with an initialised torch.distributed process group the flat
gradient buffer is all-reduced (sum, then 1/world folded into the Adam kernel) between the two calls -- one process per
GPU, backend "nccl" (RCCL over xGMI); replicas must start from identical weights.
PyTorch is used for device memory, streams and the collective only."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import weights as W
from .wavenet import _ptr, _stream


def allreduce_sum_(flat, group=None):
    """in-place sum of one flat gradient buffer over all ranks (no-op without a process group); returns world size."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return dist.get_world_size(group)


def exponential_decay(lr0, global_step, decay_steps, decay_rate):
    """tf.train.exponential_decay (model.py:320), staircase=False"""
    return float(lr0) * float(decay_rate) ** (float(global_step) / float(decay_steps))


def crop_length(sample_size, hop_size):
    """datafeeder_wavenet.py:41-47: sample_size is floored to a multiple of the hop size"""
    return int(sample_size) // int(hop_size) * int(hop_size)


class WaveNetTrainer(object):
    def __init__(self, net, hparams=None, sample_size=None, learning_rate=1e-3, decay_steps=300000, decay_rate=0.5,
                 ema_decay=0.9999, beta1=0.9, beta2=0.999, epsilon=1e-8, group=None, clip_gradients=False, l2_regularization_strength=0.0):
        self.net = net
        if hparams is not None:
            learning_rate = getattr(hparams, "wavenet_learning_rate", learning_rate)
            decay_steps = getattr(hparams, "wavenet_decay_steps", decay_steps)
            decay_rate = getattr(hparams, "wavenet_decay_rate", decay_rate)
            clip_gradients = bool(getattr(hparams, "wavenet_clip_gradients", clip_gradients))
            l2_regularization_strength = getattr(hparams, "l2_regularization_strength", l2_regularization_strength) or 0.0
            if sample_size is None:
                sample_size = hparams.sample_size
            self.max_checkpoints = getattr(hparams, "max_checkpoints", 3)
        self.lr0, self.decay_steps, self.decay_rate = learning_rate, decay_steps, decay_rate
        self.ema_decay, self.beta1, self.beta2, self.epsilon = ema_decay, beta1, beta2, epsilon
        self.group = group
        self.clip_gradients = clip_gradients                      # model.py:330-331 clip_by_global_norm(gradients, 1.)
        self.l2 = float(l2_regularization_strength or 0.0)       # train_vocoder.py:118-119: 0 -> None
        self.batch_size = net.batch_size
        self.sample_size = crop_length(sample_size, net.hop_size)
        self.device = net.device
        self._L = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._L.twv_wavenet_train_create(C.byref(net._dims), self.batch_size, self.sample_size, C.byref(h)))
        self._h = h
        self.n_params = self._L.twv_wavenet_train_param_floats(h)
        self.output_width = self._L.twv_wavenet_train_output_width(h)
        self.global_step = 0
        self.params = None
        with torch.cuda.device(self.device):
            self._ws = torch.empty(self._L.twv_wavenet_train_workspace_bytes(h) // 4, dtype=torch.float32, device=self.device)
            # the gradient buffer carries ONE extra word behind the gradients: a failure flag that rides in the same all-reduce
            # (a rank that cannot produce its batch must not leave the others waiting in the collective, and a separate flag
            # all-reduce + host sync in front of every step would cost a round trip per step)
            # clip_by_global_norm's scratch (n squares + partial sums): NOT the training workspace -- rows of that one must stay zero
            self._clip_scratch = None                        # (allocated on the first clipped step)
            self._gbuf = torch.zeros(self.n_params + 1, dtype=torch.float32, device=self.device)
            self.grads = self._gbuf[:self.n_params]
            self._flag = self._gbuf[self.n_params:]
            self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.twv_wavenet_train_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- variables: tf.global_variables_initializer / saver.restore (train_vocoder.py:139-152) ----
    def load_weights(self, tensors):
        blob = W.flatten(self.net.specs, tensors)
        assert blob.size == self.n_params, (blob.size, self.n_params)
        self.params = torch.from_numpy(blob).to(self.device)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.ema = self.params.clone()                      # EMA shadow variables start at the variable's value
        self.global_step = 0

    def init_weights(self, seed=0, scale=0.05):
        self.load_weights(W.random_tensors(self.net.specs, seed=seed, scale=scale))

    def _named(self, flat):
        out, o = {}, 0
        host = flat.detach().cpu().numpy()
        for name, shape in self.net.specs:
            n = int(np.prod(shape))
            out[name] = host[o:o + n].reshape(shape).copy()
            o += n
        return out

    def weights(self):
        return self._named(self.params)

    def ema_weights(self):
        """the ExponentialMovingAverage shadows (model.py:30,346).  NOTE: the reference's generate.py:157-158 restores the RAW
        variables by name, not these -- `weights()` is what its generation path loads."""
        return self._named(self.ema)

    # ---- saver.save / load (utils/__init__.py:62-90; train_vocoder.py:133-152,175-176) ----
    def save(self, logdir, step=None, include_optimizer=True, max_to_keep=None):
        """writes `logdir/model.ckpt-<step>` as a TensorFlow V2 bundle + the `checkpoint` state file (checkpoint.py).
        Variables carry the reference graph's names: the weights under their own names, their EMA shadows under
        `<name>/ExponentialMovingAverage`, `global_step`; with include_optimizer the Adam moments under
        `optimizer/<name>/Adam`, `/Adam_1` and `optimizer/beta{1,2}_power` ([RECALLED-TF] slot naming, unverified)."""
        import os
        from . import checkpoint as ckpt
        step = self.global_step if step is None else int(step)
        var = dict(self.weights())
        for k, v in self.ema_weights().items():
            var[k + ckpt.EMA_SUFFIX] = v
        var["global_step"] = np.asarray(step, np.int32)
        if include_optimizer:
            for k, v in self._named(self.m).items():
                var["optimizer/" + k + "/Adam"] = v
            for k, v in self._named(self.v).items():
                var["optimizer/" + k + "/Adam_1"] = v
            var["optimizer/beta1_power"] = np.asarray(self.beta1 ** (self.global_step + 1), np.float32)
            var["optimizer/beta2_power"] = np.asarray(self.beta2 ** (self.global_step + 1), np.float32)
        prefix = os.path.join(logdir, "model.ckpt-%d" % step)
        ckpt.write_bundle(prefix, var)
        # tf.train.Saver(max_to_keep=hparams.max_checkpoints) (train_vocoder.py:133): the state file lists every kept bundle, the
        # oldest ones beyond the limit are deleted
        if max_to_keep is None:
            max_to_keep = getattr(self, "max_checkpoints", 3)
        kept = [q for q in ckpt.all_checkpoint_paths(logdir) if q != prefix and os.path.exists(q + ".index")] + [prefix]
        while max_to_keep and len(kept) > max_to_keep:
            ckpt.delete_bundle(kept.pop(0))
        ckpt.write_checkpoint_state(logdir, prefix, kept)
        return prefix

    def restore(self, path, verify=False):
        """`load(saver, sess, logdir)`: logdir or bundle prefix -> global step.  Weights are required; EMA shadows / Adam
        moments are taken when the bundle has them (else EMA = weights, moments = 0, like a fresh optimizer)."""
        from . import checkpoint as ckpt
        prefix = ckpt.resolve(path)
        var = ckpt.read_bundle(prefix, verify=verify)
        self.load_weights(ckpt.wavenet_tensors(var, self.net.specs))

        def opt(fmt):
            got = {n: var.get(fmt % n) for n, _ in self.net.specs}
            if any(v is None for v in got.values()):
                return None
            return torch.from_numpy(W.flatten(self.net.specs, got)).to(self.device)
        ema, m, v = opt("%s" + ckpt.EMA_SUFFIX), opt("optimizer/%s/Adam"), opt("optimizer/%s/Adam_1")
        if ema is not None:
            self.ema = ema
        if m is not None and v is not None:
            self.m, self.v = m, v
        self.global_step = int(var["global_step"]) if "global_step" in var else ckpt.checkpoint_step(prefix)
        return self.global_step

    def gradients(self):
        return self._named(self.grads)

    def reset_workspace(self):
        """the next step clears the workspace again (twv_wavenet_train_reset_workspace): for a caller that let anything else write into
        `_ws`, or replaced it by a buffer a caching allocator may have handed out at the same address"""
        _lib.check(self._L.twv_wavenet_train_reset_workspace(self._h))

    # ---- add_loss + compute_gradients ----
    def loss_and_gradients(self, audio, local_condition, gc_ids):
        """audio (B, sample_size) float32, local_condition (B, sample_size/hop, num_mels), gc_ids (B) -> loss (device scalar);
        self.grads holds d loss / d params in the canonical order."""
        B, T = self.batch_size, self.sample_size
        audio = torch.as_tensor(audio, dtype=torch.float32).to(self.device).contiguous()
        lc = torch.as_tensor(local_condition, dtype=torch.float32).to(self.device).contiguous()
        gc = torch.as_tensor(gc_ids, dtype=torch.int32).to(self.device).contiguous()
        if tuple(audio.shape) != (B, T):
            raise ValueError("audio must be (%d, %d), got %s" % (B, T, tuple(audio.shape)))
        if tuple(lc.shape) != (B, T // self.net.hop_size, self.net.local_condition_channels):
            raise ValueError("local_condition must be (%d, %d, %d), got %s" % (B, T // self.net.hop_size,
                                                                              self.net.local_condition_channels, tuple(lc.shape)))
        if int(gc.numel()) != B:
            raise ValueError("gc_ids must hold %d ids" % B)
        if self.params is None:
            raise RuntimeError("no weights: call load_weights / init_weights first")
        with torch.cuda.device(self.device):
            _lib.check(self._L.twv_wavenet_train_loss_grad(self._h, _ptr(self.params), _ptr(audio), _ptr(lc), _ptr(gc), _ptr(self._ws),
                                                          _ptr(self.loss), _ptr(self.grads), _stream()))
            if self.l2:
                _lib.check(self._L.twv_wavenet_train_l2(self._h, _ptr(self.params), self.l2, _ptr(self._ws), _ptr(self.loss), _ptr(self.grads),
                                                       _stream()))
        return self.loss

    # ---- apply_gradients + ema.apply ----
    def apply_gradients(self, world=1):
        lr = exponential_decay(self.lr0, self.global_step, self.decay_steps, self.decay_rate)
        with torch.cuda.device(self.device):
            if self.clip_gradients:
                if self._clip_scratch is None:
                    self._clip_scratch = torch.empty(self.n_params + 2048, dtype=torch.float32, device=self.device)
                _lib.check(self._L.twv_clip_by_global_norm(_ptr(self.grads), self.n_params, 1.0 / world, 1.0, _ptr(self._clip_scratch), _stream()))
                world = 1
            _lib.check(self._L.twv_adam_ema_step(_ptr(self.params), _ptr(self.grads), _ptr(self.m), _ptr(self.v), _ptr(self.ema),
                                                self.n_params, lr, self.beta1, self.beta2, self.epsilon, self.global_step + 1,
                                                self.ema_decay, 1.0 / world, _stream()))
        self.global_step += 1
        return lr

    def step(self, audio, local_condition, gc_ids, failed=False):
        """one sess.run([net.loss, net.optimize]) (train_vocoder.py:163); returns the (local) loss as a device tensor.
        failed=True: this rank has no batch (its feeder raised): it still joins the gradient all-reduce -- with zero gradients and the
        failure flag set -- so that no rank is left waiting, and applies nothing; every rank sees `peer_failure()` afterwards.
        A rank whose loss / gradient computation raises joins the all-reduce the same way before re-raising, and NO rank applies a step
        whose reduced gradient contains a failed rank's zeros: with more than one rank the flag is read (one host sync per step)
        before Adam / EMA / global_step move, so a caller that never polls peer_failure() cannot train on a corrupted step."""
        err = None
        if failed:
            loss = self.loss
        else:
            self._flag.zero_()
            try:
                loss = self.loss_and_gradients(audio, local_condition, gc_ids)
            except Exception as e:                     # noqa: BLE001 -- re-raised below, after the peers have been released
                err, failed, loss = e, True, self.loss
        if failed:
            self._gbuf.zero_()
            self._flag.fill_(1.0)
        world = allreduce_sum_(self._gbuf, self.group)
        if err is not None:
            raise err
        if not failed and (world == 1 or not self.peer_failure()):
            self.apply_gradients(world)
        return loss

    def peer_failure(self):
        """True when some rank joined the last step's all-reduce with failed=True (host sync: call it where the loss is read anyway)"""
        return bool(float(self._flag.item()) > 0.0)
