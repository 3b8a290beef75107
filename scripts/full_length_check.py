#!/usr/bin/env python
"""Full BASELINE length (B=8 x 192000 steps): the helper-workgroup launch and the plain launch must produce the same bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import twvk_amd
from twvk_amd.wavenet import WaveNetModel
from twvk_amd import weights as W

hp = twvk_amd.default_hparams()
dil = [2 ** i for i in range(10)] * 3
B, Tm = 8, 640
T = Tm * hp.hop_size
outs = []
for helpers in (1, 0):
    m = WaveNetModel(B, dil, 2, 32, 32, 512, out_channels=30, use_biases=True, scalar_input=True, initial_filter_width=32,
                     global_condition_channels=32, global_condition_cardinality=2, local_condition_channels=80,
                     upsample_factor=[5, 5, 12], train_mode=False)
    m.set_option("helpers", helpers)
    m.load_weights(W.random_tensors(m.specs, 0, 0.05))
    rng = np.random.RandomState(1)
    mel = torch.from_numpy(rng.uniform(-4, 4, (B, Tm, 80)).astype(np.float32)).cuda()
    u = torch.from_numpy(rng.uniform(1e-5, 1 - 1e-5, (B, T, 11)).astype(np.float32)).cuda()
    U = m.create_upsample(mel)
    out = m.generate(U, (np.arange(B) % 2).astype(np.int32), (2 * rng.rand(B) - 1).astype(np.float32), u).cpu().numpy()
    outs.append(out)
    del m, U
    torch.cuda.empty_cache()
same = np.array_equal(outs[0], outs[1])
print("helpers on/off identical over %d x %d samples: %s; finite %s; |x|max %.3f" % (B, T, same, np.isfinite(outs[0]).all(), np.abs(outs[0]).max()))
sys.exit(0 if same else 1)
