#!/usr/bin/env python
"""Bring-up / timing aid for the many-streams XCD kernel (wn_xcd_many_kernel): layer dumps, samples and step time against the oracle
and the one-chain-per-stream kernel.  `python scripts/many_check.py [stage ...]` on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import make_case, make_model, mol_uniforms, first_mismatch
from oracle import oracle as O
O.build()
stages = sys.argv[1:] or ["dump", "small", "b64", "prime", "time"]


def autopsy(m, B, NL, streams=(0,)):
    """after a watchdog abort: the tags left in the streams' exchange areas (twv_xcd.hpp XcdExch) say where the step stopped"""
    ZX, PG, LG, H1, PT, LCR = 0, 8192, 12288, 16384, 16896, 17408
    CTRL = LCR + 16 * 64 * 64; SKT = CTRL + 64 + 256 + 64 + 64; WORDS = SKT + 1536
    L = m._L
    st = m._state.cpu().numpy().view(np.uint32)
    total = st.size
    ex0 = total - (B * WORDS * 2 + 16)
    for b in streams:
        e = st[ex0 + b * WORDS * 2: ex0 + (b + 1) * WORDS * 2].reshape(-1, 2)      # [granule][value, tag]
        tg = e[:, 1]
        print("   stream", b, "CTRL step", tg[CTRL], "abort", tg[CTRL + 1])
        print("    z tags  ", [int(tg[ZX + l * 128]) for l in range(NL)])
        print("    x tags  ", [int(tg[ZX + l * 128 + 1]) for l in range(NL)])
        print("    PG tags ", [int(tg[PG + l * 64]) for l in range(NL)])
        print("    LG tags ", [int(tg[LG + l * 64]) for l in range(NL)])
        print("    LCR[1] tags", [int(tg[LCR + (1 * 64 + l) * 64]) for l in range(NL)])
        print("    H1 tags ", [int(tg[H1 + g * 64]) for g in range(8)], " PT tags", [int(tg[PT + c * 64]) for c in range(8)])
        print("    SKT tags", [int(tg[SKT + h * 256]) for h in range(6)])
        MARK = CTRL + 64
        fv = e[:, 0].view(np.float32)
        print("    chain waves {stage@step}:", ["%d@%d" % (fv[MARK + w], tg[MARK + w]) for w in range(8)])
        print("    sampler autopsy {tag wanted, tag seen q0 | polls, tag seen q7} lanes 0 / 32:", [(int(tg[MARK + 240 + i]), int(e[MARK + 240 + i, 0])) for i in (0, 1, 4, 5)])
        print("    skip waves  {stage@step}:", [["%d@%d" % (fv[MARK + 16 + 8 * r + w], tg[MARK + 16 + 8 * r + w]) for w in range(8)] for r in range(8)])


def run(tag, dil, B, T, many, dbg=0, chunks=None, scale=0.05, seed=0, **kw):
    d, tensors, blob = make_case(O, dil, seed=seed, scale=scale, **kw)
    m = make_model(B, dil, tensors, **kw)
    m.set_option("xcd_many", many)
    rng = np.random.RandomState(1)
    Tm = (T + 299) // 300
    mel = rng.uniform(-4, 4, (B, Tm, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, d.O // 3)
    O.set_threads(min(B, O.set_threads(1)))
    Uo = O.upsample(d, blob, mel)[:, :T]
    want = O.generate_mol(d, blob, O.State(d, B), Uo, gc, seed_in, u)
    O.set_threads(1)
    U = m.create_upsample(mel)
    try:
        if dbg:
            got, dump = m.generate(U, gc, seed_in, u, debug_steps=dbg, check=False)
            torch.cuda.synchronize()
            print(tag, "status", m._status.cpu().numpy())
            got = got.cpu().numpy(); dump = dump.cpu().numpy()
            st = O.State(d, B); inp = seed_in.copy(); NL = len(dil)
            for t in range(dbg):
                raw, dz, dx = O.step(d, blob, st, inp, Uo[:, t], gc, debug=True)
                gz = dump[:, t, :NL * 64].reshape(B, NL, 2, 32)
                print(tag, "step", t, "z", first_mismatch(gz[:, :, 0], dz), "x", first_mismatch(gz[:, :, 1], dx), "raw", first_mismatch(dump[:, t, NL * 64:NL * 64 + d.O], raw),
                      "| got sample", got[0, t], "want", want[0, t])
                inp = want[:, t]
            if m._status.cpu().numpy()[0] != 0:
                raise RuntimeError("watchdog %d" % m._status.cpu().numpy()[0])
        elif chunks:
            outs = []; fi = seed_in; p = 0
            Ut = U.tensor() if hasattr(U, "tensor") else U
            for n in chunks:
                o = m.generate(Ut[:, p:p + n].contiguous(), gc, fi, u[:, p:p + n]).cpu().numpy()
                outs.append(o); fi = o[:, -1]; p += n
            got = np.concatenate(outs, axis=1)
        else:
            got = m.generate(U, gc, seed_in, u).cpu().numpy()
        mm = first_mismatch(got, want)
        bad = sorted(set(np.argwhere(got != want)[:, 0].tolist()))
        print(tag, "B", B, "T", T, "many", many, "MISMATCH first %s streams %s" % (mm, bad) if mm else "bit-exact")
    except Exception as e:
        print(tag, "B", B, "T", T, "many", many, "ERROR", repr(e)[:300])
        if "watchdog" in repr(e):
            autopsy(m, B, len(dil), streams=[b for b in (0, 8, 16, 24, 32) if b < B])


def exch_values(m, B):
    H1, PT = 16384, 16896
    WORDS = 17408 + 16 * 64 * 64 + 64 + 256 + 64 + 64 + 1536
    st = m._state.cpu().numpy().view(np.uint32)
    ex0 = st.size - (B * WORDS * 2 + 16)
    e = st[ex0: ex0 + B * WORDS * 2].reshape(B, WORDS, 2)
    return e[:, H1:H1 + 512, 0].copy().view(np.float32), e[:, PT:PT + 512, 0].copy().view(np.float32), e[:, H1:H1 + 512, 1].copy(), e[:, PT:PT + 512, 1].copy()


if "h1" in stages:
    for name, dil, B, scale in (("h1-8", [1, 2, 4, 8, 1, 2, 4, 8], 1, 0.2), ("h1-8b40", [1, 2, 4, 8, 1, 2, 4, 8], 40, 0.2), ("h1-30", [2 ** i for i in range(10)] * 3, 40, 0.05)):
        d, tensors, blob = make_case(O, dil, scale=scale)
        rng = np.random.RandomState(1)
        mel = rng.uniform(-4, 4, (B, 1, 80)).astype(np.float32); gc = (np.arange(B) % 2).astype(np.int32)
        seed_in = (2 * rng.rand(B) - 1).astype(np.float32); u = mol_uniforms(B, 1, 10)
        res = []
        for many in (0, 1):
            if many == 0 and B > 32:
                # reference for the many kernel at B > 32: the first 32 streams on the one-chain-per-stream kernel
                m = make_model(32, dil, tensors); m.set_option("xcd_many", 0)
                m.generate(m.create_upsample(mel[:32]), gc[:32], seed_in[:32], u[:32], check=False); torch.cuda.synchronize()
                res.append(exch_values(m, 32))
            else:
                m = make_model(B, dil, tensors); m.set_option("xcd_many", many)
                m.generate(m.create_upsample(mel), gc, seed_in, u, check=False); torch.cuda.synchronize()
                res.append(exch_values(m, B))
            print(name, "many", many, "status", m._status.cpu().numpy()[0])
        n = min(res[0][0].shape[0], res[1][0].shape[0])
        for b in range(n):
            h_bad = np.flatnonzero(res[0][0][b] != res[1][0][b]); p_bad = np.flatnonzero(res[0][1][b] != res[1][1][b])
            if h_bad.size or p_bad.size or b < 2:
                print(name, "stream", b, "H1 mismatches", h_bad.size, h_bad[:6], "tags", np.unique(res[1][2][b]), "| PT mismatches", p_bad.size, p_bad[:6], "tags", np.unique(res[1][3][b]))
                if h_bad.size:
                    i = h_bad[0]; print("     H1[%d] want %r got %r" % (i, res[0][0][b][i], res[1][0][b][i]))


if "dump" in stages:
    run("dump8b1", [1, 2, 4, 8, 1, 2, 4, 8], 1, 24, 1, dbg=4, scale=0.2)
    if "more" in stages:
        run("dump8", [1, 2, 4, 8, 1, 2, 4, 8], 40, 24, 1, dbg=4, scale=0.2)
        run("dump30", [2 ** i for i in range(10)] * 3, 40, 12, 1, dbg=3)
if "small" in stages:
    for B in (1, 9, 33, 40, 64):
        run("small", [1, 2, 4, 8, 16, 32], B, 600, 1, scale=0.1)
    for nl, kw in ((1, {}), (5, {}), (9, dict(use_bias=False)), (17, dict(G=0)), (28, dict(use_bias=False, G=0, out_channels=3))):
        run("shape%d" % nl, ([1, 2, 4, 8, 16, 32, 64] * 5)[:nl], 43, 450, 1, scale=0.1, **kw)
if "b64" in stages:
    run("b64", [2 ** i for i in range(10)] * 3, 64, 900, 1, chunks=[500, 1, 399], seed=5)
    run("b64s", [1, 2, 4, 8, 16, 32], 64, 40, 1, chunks=[15, 1, 1, 2, 21], scale=0.1)
    run("b8s", [1, 2, 4, 8, 16, 32], 8, 40, 0, chunks=[15, 1, 1, 2, 21], scale=0.1)
    run("b32s", [1, 2, 4, 8, 16, 32], 32, 40, 0, chunks=[15, 1, 1, 2, 21], scale=0.1)
    run("b48", [2 ** i for i in range(10)] * 3, 48, 900, 0, chunks=[500, 400], seed=5)     # default selection: batch > 32 -> many
    run("b96", [2 ** i for i in range(10)] * 3, 96, 900, 0, chunks=[500, 1, 399], seed=5)
    run("b75", [2 ** i for i in range(10)] * 3, 75, 600, 0, seed=6)
if "prime" in stages:
    dil = [1, 2, 4, 8, 16, 32]
    for B in (11, 40):
        d, tensors, blob = make_case(O, dil, scale=0.1)
        m = make_model(B, dil, tensors); m.set_option("xcd_many", 1)
        rf = O.receptive_field(d); rng = np.random.RandomState(9)
        seedwave = rng.uniform(-1, 1, (B, rf)).astype(np.float32)
        mel = rng.uniform(-4, 4, (B, 2, 80)).astype(np.float32); gc = (np.arange(B) % 2).astype(np.int32)
        st = O.State(d, B); zeros = np.zeros((B, 80), np.float32)
        for i in range(rf - 1):
            O.step(d, blob, st, seedwave[:, i], zeros, gc)
        u = mol_uniforms(B, 600, 10)
        want = O.generate_mol(d, blob, st, O.upsample(d, blob, mel), gc, seedwave[:, -1], u)
        try:
            m.prime(seedwave[:, :rf - 1], None, gc)
            got = m.generate(m.create_upsample(mel), gc, seedwave[:, -1], u).cpu().numpy()
            print("prime B", B, "bit-exact" if first_mismatch(got, want) is None else "MISMATCH %s" % (first_mismatch(got, want),))
        except Exception as e:
            print("prime B", B, "ERROR", repr(e)[:300])
if "time" in stages:
    import twvk_amd
    from twvk_amd import weights as W
    dil = [2 ** i for i in range(10)] * 3
    T = 12000
    for B, many in ((8, 0), (8, 1), (32, 0), (32, 1), (48, 1), (64, 1), (72, 1), (80, 1), (88, 1), (96, 1)):
        try:
            d, tensors, blob = make_case(O, dil)
            m = make_model(B, dil, tensors); m.set_option("xcd_many", many)
            rng = np.random.RandomState(1)
            mel = torch.from_numpy(rng.uniform(-4, 4, (B, T // 300, 80)).astype(np.float32)).cuda()
            gc = (np.arange(B) % 2).astype(np.int32); seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
            u = torch.from_numpy(mol_uniforms(B, T, 10)).cuda()
            U = m.create_upsample(mel)
            m.generate(U, gc, seed_in, u); torch.cuda.synchronize()
            m.queue_initializer()
            t0 = time.perf_counter(); out = m.generate(U, gc, seed_in, u); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("time B %2d many %d: %.2f us/step  %.3f M samples/s  finite %s" % (B, many, dt / T * 1e6, B * T / dt / 1e6, bool(torch.isfinite(out).all())))
        except Exception as e:
            print("time B", B, "many", many, "ERROR", repr(e)[:300])
