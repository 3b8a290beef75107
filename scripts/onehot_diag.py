"""diagnostic: tiny one-hot launches with progress prints (tuning aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import twvk_amd
from twvk_amd import weights as W
from twvk_amd.wavenet import WaveNetModel
hp = twvk_amd.default_hparams()
B = int(os.environ.get("DB", 1)); T = int(os.environ.get("DT", 4)); NLY = int(os.environ.get("DL", 30))
dil = ([2 ** i for i in range(10)] * 5)[:NLY]
m = WaveNetModel(B, dil, 2, 32, 32, 512, quantization_channels=256, out_channels=30, use_biases=True, scalar_input=False, initial_filter_width=32,
                 global_condition_channels=32, global_condition_cardinality=2, local_condition_channels=80, upsample_factor=[5, 5, 12], train_mode=False, device="cuda:0")
m.load_weights(W.random_tensors(m.specs, seed=0, scale=0.05))
rq = np.random.RandomState(91)
mel = rq.uniform(-4, 4, (B, (T + 299) // 300, 80)).astype(np.float32)
u = rq.random_sample((B, T)); fq = rq.randint(256, size=B).astype(np.int32); gc = (np.arange(B) % 2).astype(np.int32)
U = m.create_upsample(torch.from_numpy(mel).cuda())
print("fused:", m.fused_conditioning(), "launching T =", T, flush=True)
try:
    out = m.generate(U[:, :T].contiguous() if hasattr(U, "shape") else U, gc, fq, torch.from_numpy(u).cuda())
    torch.cuda.synchronize()
    print("done:", out.cpu().numpy()[:, :8], flush=True)
except Exception as e:
    print("error:", e, flush=True)
    st = m._state.cpu().numpy().view(np.uint32)
    Ls = 64
    ZX = 0; PG = ZX + Ls * 128; LG = PG + Ls * 64; H1 = LG + Ls * 64; PT = H1 + 512; LCR = PT + 512; CTRL = LCR + 16 * Ls * 64
    MARK = CTRL + 64; SEG = MARK + 256; DONE = SEG + 64; SKT = DONE + 64; H2 = SKT + 3 * 2 * 256; QL = H2 + 512; WORDS = QL + 256
    xbytes = B * WORDS * 8 + 64
    base = st.size - xbytes // 4
    ex = st[base:base + WORDS * 2].reshape(-1, 2)      # stream 0: [word] = (value bits, tag)
    zx = ex[ZX:ZX + NLY * 128].reshape(NLY, 64, 2, 2)
    print("ZX z-tag  min/max per layer:", [(int(zx[l, :, 0, 1].min()), int(zx[l, :, 0, 1].max())) for l in range(NLY)])
    print("ZX x-tag  min/max per layer:", [(int(zx[l, :, 1, 1].min()), int(zx[l, :, 1, 1].max())) for l in range(NLY)])
    pg = ex[PG:PG + NLY * 64].reshape(NLY, 64, 2)
    print("PG tag    min/max per layer:", [(int(pg[l, :, 1].min()), int(pg[l, :, 1].max())) for l in range(NLY)])
    print("H1 tags min/max", int(ex[H1:H1 + 512, 1].min()), int(ex[H1:H1 + 512, 1].max()), " H2", int(ex[H2:H2 + 512, 1].min()), int(ex[H2:H2 + 512, 1].max()),
          " QL", int(ex[QL:QL + 256, 1].min()), int(ex[QL:QL + 256, 1].max()))
    for k in range(4):
        tr = ex[MARK + 128 + k * 8:MARK + 128 + k * 8 + 8]
        print("sampler wave block %d trace (polls:dead per phase: entry, logits, max1, e, max2, x, totals, end):" % k, ["%d:%g" % (tr[i][1], tr[i][0:1].view(np.float32)[0]) for i in range(8)])
    print("progress tag", ex[CTRL][1], "abort", ex[CTRL + 1][1], "status code in the error above")
