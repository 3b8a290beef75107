"""train_vocoder.py of the reference (train_vocoder.py:26-200) on the MI355X path: same flags (--data_dir, --logdir,
--logdir_root, --restore_from, --checkpoint_every), same directory rules (utils/__init__.py:100-142 validate_directories),
same loop: `sess.run([global_step, net.loss, net.optimize])` per step is WaveNetTrainer.step -- one teacher-forced
forward/backward (twv_wavenet_train_loss_grad), the gradient all-reduce over RCCL when launched with several ranks, Adam + EMA
(twv_adam_ema_step) -- and a TensorFlow-V2 bundle every `checkpoint_every` steps, at most hparams.max_checkpoints kept.

The feeder is datafeeder_wavenet.py's batch rule without its thread and TF queue: per data directory (= speaker) npz files with
'audio' (T,) and 'mel' (T/hop, num_mels); every example is cut to `sample_size` (floored to a hop multiple) at a RANDOM FRAME
offset, audio and mel in step (datafeeder_wavenet.py:153-156, which draws from the global np.random, not the feeder's rng).
Dataset preparation and the text side of those npz files are out of scope; `--synthetic` trains on
random data of the right shapes (no dataset exists in this image)."""
import argparse
import os
import time
from collections import defaultdict
from datetime import datetime
from glob import glob

import numpy as np

from .hparams import hparams, load_hparams, save_hparams

LOGDIR_ROOT_Wavenet = './logdir-wavenet'
EPSILON = 0.001


def ensure_divisible(length, divisible_by=256, lower=True):
    """datafeeder_wavenet.py:41-47"""
    if length % divisible_by == 0:
        return length
    if lower:
        return length - length % divisible_by
    return length + (divisible_by - length % divisible_by)


def crop_example(input_wav, local_condition, max_frames, hop_size, randint=None):
    """datafeeder_wavenet.py:150-156: a random frame offset s in [0, frames - max_frames]; audio[s*hop : (s+max_frames)*hop],
    mel[s : s+max_frames].  `randint(low, high)` defaults to np.random.randint (the reference uses the GLOBAL generator here)."""
    input_wav = np.asarray(input_wav).reshape(-1)
    local_condition = np.asarray(local_condition)
    assert len(input_wav) % len(local_condition) == 0 and len(input_wav) // len(local_condition) == hop_size     # assert_ready_for_upsampling
    s = (randint or np.random.randint)(0, len(local_condition) - max_frames + 1)
    ts = s * hop_size
    return input_wav[ts:ts + hop_size * max_frames], local_condition[s:s + max_frames, :]


class DataFeederWavenet(object):
    """datafeeder_wavenet.py:50-158 as a synchronous iterator: next_batch() -> (audio (B, sample_size), mel (B, frames, num_mels),
    speaker ids (B)).  Same bookkeeping: per-directory offsets starting at 2, reshuffle on wrap-around, 32 batches' worth of
    examples drawn evenly from the directories, shuffled, cut into batches."""

    def __init__(self, data_dirs, batch_size, receptive_field, gc_enable=False, hp=hparams, rank=0, world=1):
        self.data_dirs = list(data_dirs)
        self.batch_size = batch_size
        self.receptive_field = receptive_field
        self.hop_size = hp.hop_size
        self.sample_size = ensure_divisible(hp.sample_size, self.hop_size, True)
        self.max_frames = self.sample_size // self.hop_size
        self.gc_enable = gc_enable
        self.rng = np.random.RandomState(123 + rank)                # (the reference has one feeder: RandomState(123))
        self._offset = defaultdict(lambda: 2)
        self.data_dir_to_id = {d: i for i, d in enumerate(self.data_dirs)}
        self.path_dict = {d: sorted(os.path.basename(p) for p in glob("{}/*.npz".format(d))) for d in self.data_dirs}
        for d, paths in self.path_dict.items():
            if not paths:
                raise ValueError("no .npz examples in %s" % d)
            # data-parallel ranks read DISJOINT examples (every world-th file of a directory), so that the all-reduced gradient is the
            # gradient of world x batch_size different crops; a directory with fewer files than ranks is shared whole
            if world > 1 and len(paths) >= world:
                self.path_dict[d] = paths[rank::world]
        self._batches = []

    def _get_next_example(self, data_dir):
        data_paths = self.path_dict[data_dir]
        min_len = max(self.sample_size, self.receptive_field)
        for _ in range(4 * len(data_paths) + 8):
            if self._offset[data_dir] >= len(data_paths):
                self._offset[data_dir] = 0
                self.rng.shuffle(data_paths)
            data_path = os.path.join(data_dir, data_paths[self._offset[data_dir]])
            self._offset[data_dir] += 1
            if not os.path.exists(data_path):
                continue
            data = np.load(data_path)
            if len(data['audio'].reshape(-1)) > min_len:                # get_path_dict's time_steps > min_length filter
                wav, lc = crop_example(data['audio'], data['mel'], self.max_frames, self.hop_size)
                return wav.astype(np.float32), lc.astype(np.float32), self.data_dir_to_id[data_dir]
        raise ValueError("no example in %s is longer than %d samples" % (data_dir, min_len))

    def next_batch(self):
        if not self._batches:
            n = self.batch_size
            examples = []
            for d in self.data_dirs:
                examples.extend(self._get_next_example(d) for _ in range(int(n * 32 // len(self.data_dirs))))
            self.rng.shuffle(examples)
            self._batches = [examples[i:i + n] for i in range(0, len(examples) - n + 1, n)]
        batch = self._batches.pop(0)
        return (np.stack([b[0] for b in batch]), np.stack([b[1] for b in batch]), np.asarray([b[2] for b in batch], np.int32))


class SyntheticFeeder(object):
    """random data of the feeder's shapes (audio ~ U(-1,1)*0.5, mel, alternating speakers)"""

    def __init__(self, batch_size, num_speakers, hp=hparams, seed=0):
        self.rng = np.random.RandomState(seed)
        self.batch_size, self.num_speakers = batch_size, num_speakers
        self.sample_size = ensure_divisible(hp.sample_size, hp.hop_size, True)
        self.frames, self.num_mels = self.sample_size // hp.hop_size, hp.num_mels

    def next_batch(self):
        B = self.batch_size
        return (((self.rng.rand(B, self.sample_size) - 0.5)).astype(np.float32),
                (self.rng.randn(B, self.frames, self.num_mels) * 0.5).astype(np.float32),
                (np.arange(B) % self.num_speakers).astype(np.int32))


def get_default_logdir(logdir_root):
    return os.path.join(logdir_root, 'train', datetime.now().strftime('%Y-%m-%dT%H-%M-%S'))


def validate_directories(args, hp):
    """utils/__init__.py:100-142"""
    if args.logdir and args.logdir_root:
        raise ValueError("--logdir and --logdir_root cannot be specified at the same time.")
    if args.logdir and args.restore_from:
        raise ValueError("--logdir and --restore_from cannot be specified at the same time. This is to keep your previous model from "
                         "unexpected overwrites.\nUse --logdir_root to specify the root of the directory which will be automatically "
                         "created with current date and time, or use only --logdir to just continue the training from the last checkpoint.")
    logdir_root = args.logdir_root or LOGDIR_ROOT_Wavenet
    logdir = args.logdir
    if logdir is None:
        logdir = get_default_logdir(logdir_root)
        print('Using default logdir: {}'.format(logdir))
        os.makedirs(logdir, exist_ok=True)
        save_hparams(logdir, hp)
    else:
        os.makedirs(logdir, exist_ok=True)
        if os.path.exists(os.path.join(logdir, "params.json")):
            load_hparams(hp, logdir)
        else:
            save_hparams(logdir, hp)
    restore_from = args.restore_from if args.restore_from is not None else logdir
    return {'logdir': logdir, 'logdir_root': args.logdir_root, 'restore_from': restore_from}


def get_arguments(argv=None):
    parser = argparse.ArgumentParser(description='WaveNet example network')
    parser.add_argument('--data_dir', type=str, default='./data/moon,./data/son', help='The directories (one per speaker) containing the npz examples.')
    parser.add_argument('--logdir', type=str, default=None, help='Directory in which to store the model. If the model already exists, it will '
                        'restore the state and will continue training. Cannot use with --logdir_root and --restore_from.')
    parser.add_argument('--logdir_root', type=str, default=None, help='Root directory to place the generated model under a dated subdirectory. Cannot use with --logdir.')
    parser.add_argument('--restore_from', type=str, default=None, help='Directory in which to restore the model from. Cannot use with --logdir.')
    parser.add_argument('--checkpoint_every', type=int, default=1000, help='How many steps to save each checkpoint after. Default: 1000.')
    parser.add_argument('--num_steps', type=int, default=None, help='stop after this many steps (extension; default hparams.num_steps)')
    parser.add_argument('--synthetic', action='store_true', help='random data of the right shapes instead of --data_dir (extension)')
    return parser.parse_args(argv)


def main(argv=None, log=print):
    import torch
    from .wavenet import WaveNetModel
    from .train import WaveNetTrainer
    from . import checkpoint as ckpt
    config = get_arguments(argv)
    config.data_dir = config.data_dir.split(",")
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:                                                   # one process per GPU; gradients meet in WaveNetTrainer.step
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda:%d" % local_rank))
    # rank 0 decides the directories (a default logdir carries a timestamp: every rank computing its own would give N directories)
    # and creates them / writes params.json; the others take its answer
    directories, setup_error = None, None
    if rank == 0:
        try:
            directories = validate_directories(config, hparams)
        except ValueError as e:
            print("Some arguments are wrong:")
            print(str(e))
        except Exception as e:                                      # makedirs / save_hparams: the other ranks are waiting in the broadcast
            setup_error = e
    if dist is not None:
        box = [directories]
        dist.broadcast_object_list(box, src=0)                      # None = rank 0 could not set the run up: every rank leaves
        directories = box[0]
        if directories is not None and rank != 0 and os.path.exists(os.path.join(directories['logdir'], "params.json")):
            load_hparams(hparams, directories['logdir'])
    if setup_error is not None:
        raise setup_error
    if directories is None:
        return None
    logdir, restore_from = directories['logdir'], directories['restore_from']
    is_overwritten_training = logdir != restore_from
    num_speakers = len(config.data_dir)
    gc_enable = num_speakers > 1
    rf = WaveNetModel.calculate_receptive_field(hparams.filter_width, hparams.dilations, hparams.scalar_input, hparams.initial_filter_width)
    if config.synthetic:
        reader = SyntheticFeeder(hparams.wavenet_batch_size, max(num_speakers, 1), hparams, seed=rank)
    else:
        reader = DataFeederWavenet(config.data_dir, hparams.wavenet_batch_size, rf, gc_enable=gc_enable, hp=hparams, rank=rank, world=world)
        np.random.seed(1234 + rank)                                 # ranks must not draw the same crop offsets
    net = WaveNetModel(batch_size=hparams.wavenet_batch_size, dilations=hparams.dilations, filter_width=hparams.filter_width,
                       residual_channels=hparams.residual_channels, dilation_channels=hparams.dilation_channels,
                       quantization_channels=hparams.quantization_channels, out_channels=hparams.out_channels,
                       skip_channels=hparams.skip_channels, use_biases=hparams.use_biases, scalar_input=hparams.scalar_input,
                       initial_filter_width=hparams.initial_filter_width, global_condition_channels=hparams.gc_channels,
                       global_condition_cardinality=num_speakers, local_condition_channels=hparams.num_mels,
                       upsample_factor=hparams.upsample_factor, train_mode=True, device="cuda:%d" % local_rank)
    trainer = WaveNetTrainer(net, hparams)
    trainer.init_weights(seed=0)                                    # tf.global_variables_initializer (identical on every rank)
    start_step = None
    if ckpt.latest_checkpoint(restore_from) is not None:            # load(saver, sess, restore_from)
        start_step = trainer.restore(restore_from, verify=True)
        log("  Global step was: {}".format(start_step))
    if is_overwritten_training or start_step is None:
        trainer.global_step = 0                                     # train_vocoder.py:141-145
    num_steps = config.num_steps if config.num_steps is not None else hparams.num_steps
    step, loss_value = trainer.global_step, float("nan")
    while step < num_steps:
        start_time = time.time()
        failure = None
        try:
            audio, lc, gc = reader.next_batch()
        except Exception as e:                                      # a rank that cannot feed must not leave the others in the all-reduce
            failure = e
        # the failure flag rides in the gradient all-reduce itself (WaveNetTrainer.step): a failing rank still joins the collective
        loss = trainer.step(None, None, None, failed=True) if failure is not None else trainer.step(audio, lc, gc)
        if failure is not None:
            raise failure
        step = trainer.global_step
        loss_value = float(loss.item())
        if dist is not None and trainer.peer_failure():             # read where the loss is read: the stream is drained already
            raise RuntimeError("another rank's feeder failed at step %d" % step)
        log('step {:d} - loss = {:.3f}, ({:.3f} sec/step)'.format(step, loss_value, time.time() - start_time))
        if step % config.checkpoint_every == 0 and rank == 0:
            log('Storing checkpoint to {} ...'.format(logdir))
            trainer.save(logdir, step)
    return {"logdir": logdir, "step": step, "loss": loss_value}


if __name__ == '__main__':
    main()
    print('Done')
