"""Bring-up aid: run the instrumented XCD-per-stream kernel and, on a watchdog abort, print where every wave of stream 0 sits."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_case, make_model, mol_uniforms
from oracle import oracle as O
from twvk_amd import _lib
B = int(os.environ.get("XB", 8)); T = int(os.environ.get("XT", 600)); NLAY = int(os.environ.get("XL", 30))
dil = ([1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3)[:NLAY]
d, tensors, blob = make_case(O, dil)
m = make_model(B, dil, tensors)
rng = np.random.RandomState(1)
mel = rng.uniform(-4, 4, (B, (T + 299) // 300, 80)).astype(np.float32)
U = m.create_upsample(mel)
u = mol_uniforms(B, T, 10)
prof = torch.zeros((T, 64), dtype=torch.int64, device="cuda")
if os.environ.get("XINSTR", "1") == "1":
    _lib.check(m._L.twv_wavenet_set_profile_buffer(m._h, C.c_void_p(prof.data_ptr()), T))
try:
    out = m.generate(U, (np.arange(B) % 2).astype(np.int32), np.zeros(B, np.float32), u)
    print("ok", float(out.abs().max()))
except Exception as e:
    print("FAILED:", e)
    st = m._state.cpu().numpy().view(np.uint32)
    L = m._L
    # exchange area sits behind the per-stream state and the generic kernel's exchange words
    total = st.size
    WORDS = (32 * 128 + 2 * 32 * 64 + 512 + 512 + 16 * 32 * 64 + 64 + 256)
    xbytes = B * WORDS * 8 + 64
    base = total - xbytes // 4
    ex = st[base:base + WORDS * 2].reshape(-1, 2)      # stream 0: [word] = (value bits, tag)
    mark = ex[WORDS - 256:]
    names = {0: "chain", 1: "service"}
    for r in range(22):
        row = ["%d:%g" % (mark[r * 8 + w][1], mark[r * 8 + w][0:1].view(np.float32)[0]) for w in range(8)]
        nm = names.get(r, "skip%d" % (r - 2) if r < 10 else ("conv%d" % (r - 10) if r < 18 else "lc%d" % (r - 18)))
        print("%-8s (tag:stage per wave) %s" % (nm, "  ".join(row)))
    zx = ex[:32 * 128].reshape(32, 64, 2, 2)
    print("ZX z-tag min/max per layer:", [(int(zx[l, :, 0, 1].min()), int(zx[l, :, 0, 1].max())) for l in range(NLAY)])
    for l in range(NLAY):
        if zx[l, :, 0, 1].min() != zx[l, :, 0, 1].max() or zx[l, :, 1, 1].min() != zx[l, :, 1, 1].max():
            print("ZX layer", l, "z tags", zx[l, :, 0, 1].tolist(), "x tags", zx[l, :, 1, 1].tolist())
    for v in range(8):
        r = mark[176 + v * 4:176 + v * 4 + 3]
        print("skip0 wave %d abandoned poll: saw tags qa %d qb %d, layer %d, polls %d, wanted tag %d, own index %d" % (v, r[0][1], r[0][0], r[1][1], r[1][0], r[2][1], r[2][0]))
    ctrl = ex[WORDS - 256 - 64:WORDS - 256]
    print("progress tag", ctrl[0][1], "abort", ctrl[1][1])
