#!/usr/bin/env python
"""Phase stamps of the fused training-layer kernels (tuning aid).  Needs the variant build:
    TWV_EXTRA_HIPCC_FLAGS=-DTWV_TRPROF python scripts/train_trprof.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "1"]
exec(open(os.path.join(ROOT, "scripts", "train_bench.py")).read())
from twvk_amd import _lib
L = _lib.lib()
buf = (C.c_ulonglong * (3 * 8 * 16))()
L.twv_debug_trprof.argtypes = [C.c_void_p]
assert L.twv_debug_trprof(buf) == 0
p = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(3, 8, 16)
for k in (0, 1, 2):
    if not p[k].any(): continue
    print("kernel", k, "(block 0 / wave 0; s_memtime ticks between consecutive stamps of a tile | from this tile's first stamp to the next tile's)")
    for t in range(8):
        row = p[k, t]
        nz = [i for i in range(16) if row[i]]
        if not nz: continue
        d = [int(row[nz[i + 1]] - row[nz[i]]) for i in range(len(nz) - 1)]
        nxt = int(p[k, t + 1, 0] - row[0]) if t + 1 < 8 and p[k, t + 1, 0] > row[0] else -1
        print("  tile %d:" % t, " ".join("%6d" % v for v in d), "|", nxt)
