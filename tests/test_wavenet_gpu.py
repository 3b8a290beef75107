"""-m gpu: the parity tests proper.  Every check calls the HIP path through the C-ABI (ctypes) and compares it
with the CPU oracle BIT FOR BIT (float32 == float32; the arithmetic contract of DESIGN.md makes that the bar)."""
import numpy as np
import pytest

from helpers import first_mismatch, make_case, make_model, mol_uniforms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def test_native_library_is_loaded(torch_cuda):
    import twvk_amd
    L = twvk_amd._lib.lib()
    assert b"gfx950" in L.twv_version()
    # the binary is stamped with the hash of the sources it was built from: what runs here IS this tree (not a stale shipped build)
    assert L.twv_version().decode().endswith("src:" + twvk_amd._lib.source_hash()), (L.twv_version(), twvk_amd._lib.source_hash())
    with open("/proc/self/maps") as f:
        assert "libtwv_amd.so" in f.read(), "the in-tree HIP library must be the thing that runs"


def test_crosslane_primitives(torch_cuda):
    """pins v_permlane32_swap / v_readlane semantics the chain wave relies on"""
    import ctypes as C
    import twvk_amd
    torch = torch_cuda
    out = torch.zeros(256, dtype=torch.float32, device="cuda:0")
    twvk_amd._lib.check(twvk_amd._lib.lib().twv_selftest(C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(64, 4)
    lanes = np.arange(64)
    assert np.array_equal(o[:, 0], (lanes % 32) + 1), o[:, 0]          # new vdst: low half everywhere
    assert np.array_equal(o[:, 1], (lanes % 32) + 33), o[:, 1]         # new src: high half everywhere
    assert np.all(o[:, 2] == 6.0)
    assert np.array_equal(o[:, 3], (lanes ^ 1) + 1)


@pytest.mark.parametrize("name,lo,hi", [("tanh", -12, 12), ("sigmoid", -25, 25), ("exp", -100, 100), ("log1p", 0, 1)])
def test_elementwise_bit_exact(torch_cuda, oracle, name, lo, hi):
    from twvk_amd import ops
    rng = np.random.RandomState(5)
    x = np.concatenate([rng.uniform(lo, hi, 200000), [0.0, -0.0, lo, hi, 1e-30, -1e-30, 1e-42]]).astype(np.float32)
    if name == "log1p":
        x = np.abs(x)
    got = ops.eval_elementwise(name, x).cpu().numpy()
    want = oracle.elementwise(name, x)
    assert first_mismatch(got, want) is None, (name, first_mismatch(got, want))


def test_log_bit_exact(torch_cuda, oracle):
    from twvk_amd import ops
    rng = np.random.RandomState(6)
    x = np.concatenate([np.exp(rng.uniform(-80, 80, 200000)), rng.uniform(1e-5, 1, 100000), [1.0, 0.5, 2.0, 1e-38, 1e-42]]).astype(np.float32)
    got = ops.eval_elementwise("log", x).cpu().numpy()
    want = oracle.elementwise("log", x)
    assert first_mismatch(got, want) is None


@pytest.mark.parametrize("Q", [256, 100, 1000, 16])
def test_categorical_sampler_rows_bit_exact(torch_cuda, oracle, Q):
    """generate.py:219-231 + model.py:243 on the device (twv_sample_categorical, the code both generation kernels draw with) against
    the checker: drawn class ids equal element for element, scaled probabilities bit for bit, three temperatures.  The contract
    itself (AC-5) is held against numpy's literal execution of those lines in tests/test_cpu.py."""
    from twvk_amd import ops
    rng = np.random.RandomState(100 + Q)
    N = 6000
    for temp in (1.0, 0.8, 1.3):
        logits = (rng.randn(N, Q) * rng.uniform(0.5, 4.0, (N, 1))).astype(np.float32)
        u = rng.random_sample(N)
        u[:8] = [0.0, 1.0 - 2.0 ** -53, 0.5, 1e-300, 0.999999, 1e-9, 0.25, 0.75]
        # draws that sit ON a cdf boundary to within an ulp or two (the kernel compares against u * last and only divides inside a
        # 2^-50 band around it: these rows take the division, and must still equal the checker's plain division)
        for i in range(8, 264):
            _, p0 = oracle.sample_categorical(logits[i], temp, 0.5)
            cdf = np.cumsum(p0.astype(np.float64)); cdf /= cdf[-1]
            j = int(rng.randint(0, Q - 1))
            u[i] = min(np.nextafter(cdf[j], [0.0, 1.0, cdf[j]][i % 3]), 1.0 - 2.0 ** -53)
        got, proba = ops.sample_categorical(logits, temp, u, want_proba=True)
        got, proba = got.cpu().numpy(), proba.cpu().numpy()
        want = np.empty(N, np.int32); wp = np.empty((N, Q), np.float32)
        for i in range(N):
            k, p = oracle.sample_categorical(logits[i], temp, u[i])
            want[i] = k; wp[i] = p
        assert np.array_equal(got, want), (temp, first_mismatch(got, want))
        assert first_mismatch(proba, wp) is None, (temp, first_mismatch(proba, wp))
        assert got.min() >= 0 and got.max() < Q


def test_nan_logits_are_an_error_not_a_sample(torch_cuda, oracle):
    """np.random.choice (generate.py:231) raises 'ValueError: probabilities contain NaN' when a logit is NaN or infinite; the device
    sampler marks such a row with class id -1 (twv_sample_categorical), ops.sample_categorical raises like numpy, and inside a
    generation launch the condition sets status code 31, which generate() turns into a TwvError -- on both generation kernels."""
    from twvk_amd import ops
    from twvk_amd._lib import TwvError
    rng = np.random.RandomState(5)
    Q, N = 256, 64
    logits = rng.randn(N, Q).astype(np.float32)
    u = rng.random_sample(N)
    clean = ops.sample_categorical(logits, 1.0, u).cpu().numpy()
    bad = logits.copy(); bad[7, 100] = np.nan; bad[40, 3] = np.inf
    with pytest.raises(ValueError, match="probabilities contain NaN"):
        ops.sample_categorical(bad, 1.0, u)
    got = ops.sample_categorical(bad, 1.0, u, check=False).cpu().numpy()
    assert got[7] == -1 and got[40] == -1
    keep = np.ones(N, bool); keep[[7, 40]] = False
    assert np.array_equal(got[keep], clean[keep])
    # numpy itself, on the reference's own lines (generate.py:219-231), for the same row
    with pytest.raises(ValueError, match="probabilities contain NaN"):
        p = np.exp(bad[7].astype(np.float64) - np.nanmax(bad[7])); p = (p / p.sum()).astype(np.float32)
        sp = np.log(p); sp = np.exp(sp - np.logaddexp.reduce(sp))
        np.random.RandomState(0).choice(np.arange(Q), p=sp)
    dil = [1, 2, 4, 8]
    d, tensors, blob = make_case(oracle, dil, scalar_input=False, Q=256)
    tensors = dict(tensors)
    b2 = tensors["wavenet/conv1d_2/bias"].copy(); b2[17] = np.nan
    tensors["wavenet/conv1d_2/bias"] = b2
    for xcd in (1, 0):
        m = make_model(2, dil, tensors, scalar_input=False, Q=256, xcd=xcd)
        U = rng.uniform(-4, 4, (2, 40, 80)).astype(np.float32)
        with pytest.raises(TwvError, match="NaN"):
            m.generate(U, np.array([0, 1], np.int32), np.array([128, 3], np.int32), rng.random_sample((2, 40)))
        m.queue_initializer()


@pytest.mark.parametrize("name", ["exp64", "log64", "exp64_nonpos"])
def test_elementwise64_bit_exact(torch_cuda, oracle, name):
    """exp64_nonpos: the float64 softmax's straight-line exp (x = logit - max <= 0) against the contract's exp64 over the whole
    non-positive range, the subnormal results and the underflow edge included"""
    from twvk_amd import ops
    rng = np.random.RandomState(7)
    if name == "exp64_nonpos":
        x = np.concatenate([-rng.uniform(0, 760, 60000), -np.exp(rng.uniform(-40, 7, 20000)), np.linspace(-746.0, -700.0, 20001),
                            [0.0, -0.0, -745.13321910194110842, -745.1332191019412, -745.133219101941, -708.39641853226408, -1e-320, -np.inf]])
        got = ops.eval_elementwise(name, x).cpu().numpy()
        want = oracle.elementwise("exp64", x)
        assert first_mismatch(got, want) is None, first_mismatch(got, want)
        return
    x = rng.uniform(-700, 700, 50000) if name == "exp64" else np.exp(rng.uniform(-700, 700, 50000))
    got = ops.eval_elementwise(name, x).cpu().numpy()
    want = oracle.elementwise(name, x)
    assert first_mismatch(got, want) is None


def test_mu_law_codec(torch_cuda, oracle):
    from twvk_amd import ops
    rng = np.random.RandomState(8)
    a = np.concatenate([rng.uniform(-1.2, 1.2, 100000), [0.0, 1.0, -1.0, 0.5]]).astype(np.float32)
    q = ops.mu_law_encode(a, 256).cpu().numpy()
    assert np.array_equal(q, oracle.mu_law_encode(a, 256))
    assert q[-4] == 128 and q[-3] == 255 and q[-2] == 0          # ops.py:22-33 known answers
    allq = np.arange(256, dtype=np.int32)
    assert first_mismatch(ops.mu_law_decode(allq, 256).cpu().numpy(), oracle.mu_law_decode(allq, 256)) is None
    y = rng.uniform(-1, 1, 10000).astype(np.float32)
    assert first_mismatch(ops.mu_law_decode(y, 256, quantization=False).cpu().numpy(), oracle.mu_law_expand(y, 256)) is None


def test_upsample(torch_cuda, oracle):
    dil = [1, 2, 4]
    d, tensors, blob = make_case(oracle, dil, S=64, scale=0.3)
    m = make_model(2, dil, tensors, S=64)
    mel = np.random.RandomState(1).uniform(-4, 4, (2, 7, 80)).astype(np.float32)
    got = m.create_upsample(mel).cpu().numpy()
    want = oracle.upsample(d, blob, mel)
    assert got.shape == (2, 7 * 300, 80)                         # generate.py:152 T_mel * hop_size
    assert first_mismatch(got, want) is None


def _run_mol(oracle, torch, dil, B, Tm, S=512, ifw=32, use_bias=True, G=32, L=80, up=(5, 5, 12), workers=None, scale=0.05,
             T=None, debug_steps=0, seed=0, groups=None):
    d, tensors, blob = make_case(oracle, dil, S=S, ifw=ifw, use_bias=use_bias, G=G, L=L, up=up, seed=seed, scale=scale)
    m = make_model(B, dil, tensors, S=S, ifw=ifw, use_bias=use_bias, G=G, L=L, up=up, workers=workers, groups=groups)
    rng = np.random.RandomState(1)
    hop = int(np.prod(up)) if L else 1
    T = T or Tm * hop
    if L:
        mel = rng.uniform(-4, 4, (B, Tm, L)).astype(np.float32)
        U_o = oracle.upsample(d, blob, mel)[:, :T]
        U_g = m.create_upsample(mel)[:, :T].contiguous()
    else:
        U_o = U_g = None
    gc = (np.arange(B) % 2).astype(np.int32) if G else None
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)            # generate.py:188
    u = mol_uniforms(B, T, d.O // 3)
    st = oracle.State(d, B)
    want = oracle.generate_mol(d, blob, st, U_o, gc, seed_in, u)
    res = m.generate(U_g, gc, seed_in, u, debug_steps=debug_steps)
    return d, blob, m, want, res, (U_o, gc, seed_in, u)


def test_generate_small_with_layer_dumps(torch_cuda, oracle):
    """first line of defence: per-layer z / x and raw outputs of the first steps against the oracle's own dumps"""
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    B, T, dbg = 2, 24, 4
    d, blob, m, want, (got, dump), (U_o, gc, seed_in, u) = _run_mol(oracle, torch_cuda, dil, B, 1, T=T, debug_steps=dbg, scale=0.2)
    # oracle step-by-step with dumps
    st = oracle.State(d, B)
    inp = seed_in.copy()
    dump = dump.cpu().numpy()
    NL = len(dil)
    for t in range(dbg):
        raw, dz, dx = oracle.step(d, blob, st, inp, U_o[:, t], gc, debug=True)
        gz = dump[:, t, :NL * 64].reshape(B, NL, 2, 32)
        assert first_mismatch(gz[:, :, 0], dz) is None, ("z", t, first_mismatch(gz[:, :, 0], dz))
        assert first_mismatch(gz[:, :, 1], dx) is None, ("x", t, first_mismatch(gz[:, :, 1], dx))
        graw = dump[:, t, NL * 64:NL * 64 + d.O]
        assert first_mismatch(graw, raw) is None, ("raw", t, first_mismatch(graw, raw))
        inp = want[:, t]
    assert first_mismatch(got.cpu().numpy(), want) is None


@pytest.mark.parametrize("groups", [1, 2, 4, 8])
def test_generate_c2_shape_short(torch_cuda, oracle, groups):
    """C2 architecture (3x[1..512], R=D=32, S=512, MoL-30, gc+lc), B=2, 2 mel frames = 600 samples; any number of
    workgroups per stream (launch geometry) must give the same bits"""
    dil = [2 ** i for i in range(10)] * 3
    d, blob, m, want, got, _ = _run_mol(oracle, torch_cuda, dil, 2, 2, groups=groups)
    got = got.cpu().numpy()
    assert got.shape == want.shape == (2, 600)
    assert first_mismatch(got, want) is None, first_mismatch(got, want)
    assert np.all(np.abs(got) <= 1.0)


@pytest.mark.parametrize("helpers,S,groups", [(0, 512, 8), (1, 512, 8), (2, 512, 8), (1, 512, 4), (1, 128, 2), (2, 256, 4), (1, 256, 2)])
def test_helper_workgroups_do_not_change_results(torch_cuda, oracle, helpers, S, groups):
    """`helpers`: 0 = every stream workgroup runs conv1d_1 / conv1d_2 itself, 2 = conv1d_1 in helper workgroups, 1 (default) = conv1d_1
    and conv1d_2's chunk partials in helper workgroups -- a launch-geometry choice, the bits must not move; state carries over"""
    dil = [1, 2, 4, 8, 16, 32, 64, 128]
    d, tensors, blob = make_case(oracle, dil, S=S)
    m = make_model(2, dil, tensors, S=S, groups=groups)
    m.set_option("helpers", helpers)
    rng = np.random.RandomState(4)
    mel = rng.uniform(-4, 4, (2, 2, 80)).astype(np.float32)
    U_o = oracle.upsample(d, blob, mel); U_g = m.create_upsample(mel)
    gc = np.array([1, 0], np.int32)
    seed_in = (2 * rng.rand(2) - 1).astype(np.float32)
    u = mol_uniforms(2, 600, d.O // 3)
    want = oracle.generate_mol(d, blob, oracle.State(d, 2), U_o, gc, seed_in, u)
    a = m.generate(U_g[:, :250].contiguous(), gc, seed_in, u[:, :250]).cpu().numpy()          # two launches: the state carries over
    b = m.generate(U_g[:, 250:].contiguous(), gc, a[:, -1], u[:, 250:]).cpu().numpy()
    got = np.concatenate([a, b], axis=1)
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


def test_generate_past_longest_delay_line(torch_cuda, oracle):
    """long enough that every delay line (d=512) wraps more than twice"""
    dil = [1, 4, 16, 64, 256, 512]
    d, blob, m, want, got, _ = _run_mol(oracle, torch_cuda, dil, 1, 5, S=128, scale=0.15)
    assert first_mismatch(got.cpu().numpy(), want) is None


@pytest.mark.parametrize("kw", [dict(use_bias=False), dict(G=0), dict(L=0), dict(ifw=8), dict(S=64), dict(S=1024), dict(S=1024, groups=16), dict(S=128, groups=2), dict(groups=1),
                                dict(up=(3, 4))])
def test_generate_variants(torch_cuda, oracle, kw):
    dil = [1, 2, 4, 8, 16, 1, 2]
    d, blob, m, want, got, _ = _run_mol(oracle, torch_cuda, dil, 3, 6, T=70 if kw.get("L", 80) else 70, scale=0.1, **kw)
    assert first_mismatch(got.cpu().numpy(), want) is None, kw


def test_unchecked_calls_end_with_a_status_read(torch_cuda, oracle):
    """generate(check=False) returns without a host sync and without reading the status word; status() is the read that ends such a
    sequence: 0 after clean launches, TwvError after one that saw non-finite logits (the one-hot model with a NaN weight)"""
    from twvk_amd._lib import TwvError
    dil = [1, 2, 4, 8] * 2
    B, T = 2, 24
    d, tensors, blob = make_case(oracle, dil, S=512, scale=0.1)
    m = make_model(B, dil, tensors, S=512)
    rng = np.random.RandomState(5)
    U = rng.uniform(-1, 1, (B, T, 80)).astype(np.float32)
    gc = np.array([1, 0], np.int32)
    seed_in = rng.uniform(-1, 1, B).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d, blob, oracle.State(d, B), U, gc, seed_in, u)
    a = m.generate(U[:, :10], gc, seed_in, u[:, :10], check=False)
    b = m.generate(U[:, 10:], gc, a[:, -1].cpu().numpy(), u[:, 10:], check=False)
    assert m.status() == 0
    assert first_mismatch(np.concatenate([a.cpu().numpy(), b.cpu().numpy()], axis=1), want) is None
    d2, t2, _ = make_case(oracle, [1, 2, 4, 8], scalar_input=False, Q=256)
    t2 = dict(t2)
    b2 = t2["wavenet/conv1d_2/bias"].copy(); b2[17] = np.nan
    t2["wavenet/conv1d_2/bias"] = b2
    m2 = make_model(2, [1, 2, 4, 8], t2, scalar_input=False, Q=256)
    out = m2.generate(rng.uniform(-4, 4, (2, 40, 80)).astype(np.float32), np.array([0, 1], np.int32), np.array([128, 3], np.int32),
                      rng.random_sample((2, 40)), check=False)
    assert out.shape == (2, 40)
    with pytest.raises(TwvError, match="NaN"):
        m2.status()


@pytest.mark.parametrize("S", [128, 512])
def test_state_carries_over_between_calls(torch_cuda, oracle, S):
    """generate(T1) then generate(T2) == generate(T1+T2); n_steps=1 calls == one sess.run each (generate.py:211).
    S = 128: the generic kernel; S = 512: the XCD kernel fed with materialised upsampled rows"""
    dil = [1, 2, 4, 8] * 2
    B, T = 2, 40
    d, tensors, blob = make_case(oracle, dil, S=S, scale=0.1)
    m = make_model(B, dil, tensors, S=S)
    rng = np.random.RandomState(3)
    U = rng.uniform(-1, 1, (B, T, 80)).astype(np.float32)
    gc = np.array([1, 0], np.int32)
    seed_in = rng.uniform(-1, 1, B).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d, blob, oracle.State(d, B), U, gc, seed_in, u)
    a = m.generate(U[:, :15], gc, seed_in, u[:, :15]).cpu().numpy()
    b = m.generate(U[:, 15:16], gc, a[:, -1], u[:, 15:16]).cpu().numpy()          # a single step
    c = m.generate(U[:, 16:], gc, b[:, -1], u[:, 16:]).cpu().numpy()
    got = np.concatenate([a, b, c], axis=1)
    assert first_mismatch(got, want) is None
    # queue_initializer really resets (generate.py:163)
    m.queue_initializer()
    again = m.generate(U, gc, seed_in, u).cpu().numpy()
    assert first_mismatch(again, want) is None
    # predict_proba_incremental == one step
    m.queue_initializer()
    one = m.predict_proba_incremental(seed_in, U[:, 0], gc, uniforms=u[:, 0]).cpu().numpy()
    assert first_mismatch(one[:, 0], want[:, 0]) is None


def test_committed_restatement_fixture(torch_cuda):
    """HIP path against the committed restatement_* fixture (no oracle library needed at run time)"""
    import os
    from twvk_amd import weights as W
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "restatement_wavenet_mol_small.npz"))
    dil = [int(v) for v in g["dilations"]]
    specs = W.tensor_specs(len(dil), 32, 32, int(g["S"]), 256, 30, True, 32, True, 32, 2, 80, (5, 5, 12))
    rng = np.random.RandomState(int(g["weight_seed"]))
    tensors = {n: (rng.randn(*shp) * float(g["scale"])).astype(np.float32) for n, shp in specs}
    m = make_model(2, dil, tensors, S=int(g["S"]))
    U = m.create_upsample(g["mel"])
    assert first_mismatch(U[:, :8].cpu().numpy(), g["upsampled_head"]) is None
    T = g["uniforms"].shape[1]
    out = m.generate(U[:, :T].contiguous(), g["gc_ids"], g["first_input"], g["uniforms"]).cpu().numpy()
    assert first_mismatch(out, g["samples"]) is None


def test_committed_restatement_fixture_mulaw(torch_cuda):
    """one-hot mu-law model against the committed fixture (no oracle library at run time), both temperatures"""
    import os
    from twvk_amd import weights as W
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "restatement_wavenet_mulaw_small.npz"))
    dil = [int(v) for v in g["dilations"]]
    specs = W.tensor_specs(len(dil), S=int(g["S"]), Q=int(g["Q"]), scalar_input=False)
    tensors = W.random_tensors(specs, seed=int(g["weight_seed"]), scale=float(g["scale"]))
    for temp, key in ((1.0, "samples_t10"), (0.8, "samples_t08")):
        m = make_model(2, dil, tensors, scalar_input=False, S=int(g["S"]), Q=int(g["Q"]))
        out = m.generate(g["upsampled"], g["gc_ids"], g["first_input"], g["uniforms"], temperature=temp).cpu().numpy()
        assert np.array_equal(out, g[key]), (key, first_mismatch(out.astype(np.float32), g[key].astype(np.float32)))


def _run_onehot(oracle, dil, B, T, S=512, Q=256, temperature=1.0, groups=None, scale=0.05, L=80, G=32, debug_steps=0):
    d, tensors, blob = make_case(oracle, dil, scalar_input=False, S=S, Q=Q, scale=scale, L=L, G=G)
    m = make_model(B, dil, tensors, scalar_input=False, S=S, Q=Q, L=L, G=G, groups=groups)
    rng = np.random.RandomState(1)
    U = rng.uniform(-4, 4, (B, T, L)).astype(np.float32) if L else None
    gc = (np.arange(B) % 2).astype(np.int32) if G else None
    seed_in = rng.randint(Q, size=B).astype(np.int32)              # generate.py:192
    u = np.random.RandomState(2).random_sample((B, T))             # np.random.choice's single draw per sample
    want = oracle.generate_mulaw(d, blob, oracle.State(d, B), U, gc, seed_in, u, temperature)
    res = m.generate(U, gc, seed_in, u, temperature=temperature, debug_steps=debug_steps)
    return d, blob, want, res, (U, gc, seed_in, u)


@pytest.mark.parametrize("temperature", [1.0, 0.8])
def test_generate_onehot_small_with_dumps(torch_cuda, oracle, temperature):
    dil = [1, 2, 4, 8, 1, 2, 4, 8]
    B, T, dbg = 2, 40, 3
    d, blob, want, (got, dump), (U, gc, seed_in, u) = _run_onehot(oracle, dil, B, T, S=128, temperature=temperature, scale=0.3,
                                                                  debug_steps=dbg)
    dump = dump.cpu().numpy()
    st = oracle.State(d, B)
    inp = seed_in.copy()
    NL = len(dil)
    for t in range(dbg):
        raw, dz, dx = oracle.step(d, blob, st, inp, U[:, t], gc, debug=True)
        gz = dump[:, t, :NL * 64].reshape(B, NL, 2, 32)
        assert first_mismatch(gz[:, :, 0], dz) is None, ("z", t)
        assert first_mismatch(gz[:, :, 1], dx) is None, ("x", t)
        assert first_mismatch(dump[:, t, NL * 64:NL * 64 + 256], raw) is None, ("logits", t)
        inp = want[:, t]
    assert np.array_equal(got.cpu().numpy(), want), first_mismatch(got.cpu().numpy(), want)


def test_predict_proba_incremental_returns_probabilities_for_the_onehot_model(torch_cuda, oracle):
    """model.py:241-243: the one-hot model's predict_proba_incremental returns tf.cast(softmax(float64(logits)), float32), (B, Q);
    the queues advance by one step per call.  Tolerance 1e-6 on probabilities (float output; north_star's float bar is 1e-4)."""
    dil = [1, 2, 4, 8, 1, 2]
    B, Q = 2, 256
    d, tensors, blob = make_case(oracle, dil, scalar_input=False, S=128, Q=Q, scale=0.3)
    m = make_model(B, dil, tensors, scalar_input=False, S=128, Q=Q)
    rng = np.random.RandomState(5)
    U = rng.uniform(-4, 4, (B, 3, 80)).astype(np.float32)
    gc = np.array([1, 0], np.int32)
    st = oracle.State(d, B)
    inp = rng.randint(Q, size=B).astype(np.int32)
    for t in range(3):
        raw, _, _ = oracle.step(d, blob, st, inp, U[:, t], gc, debug=True)
        x = raw.astype(np.float64)
        e = np.exp(x - x.max(axis=1, keepdims=True))
        want = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
        got = m.predict_proba_incremental(inp, U[:, t], gc).cpu().numpy()
        assert got.shape == (B, Q) and got.dtype == np.float32
        assert np.abs(got - want).max() <= 1e-6, (t, np.abs(got - want).max())
        assert abs(float(got.sum(axis=1).max()) - 1.0) < 1e-5
        inp = want.argmax(axis=1).astype(np.int32)


def test_generic_kernel_conditioning_is_bounded(torch_cuda, oracle):
    """the generic kernel's hoisted projection table grows with B*T (11.8 GB at configs[1]): generate() / prime() cut long requests
    into calls of bounded table size with the state carried over -- same samples as one call.  Forced here with a tiny bound."""
    dil = [1, 2, 4, 8, 16]
    B, T = 2, 1500
    d, tensors, blob = make_case(oracle, dil, S=128, scale=0.2)
    rng = np.random.RandomState(2)
    mel = rng.uniform(-4, 4, (B, 5, 80)).astype(np.float32)
    gc = np.array([0, 1], np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel), gc, seed_in, u)
    m = make_model(B, dil, tensors, S=128)
    assert not m.fused_conditioning()
    m.MAX_COND_BYTES = 4 * B * len(dil) * 64 * 450            # room for 450 steps -> one hop (300 steps) per call
    assert m._steps_per_call(T) == 300
    got = m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)
    # priming through the same bound, then generation
    rf = oracle.receptive_field(d)
    seedwave = rng.uniform(-1, 1, (B, 700)).astype(np.float32)
    st = oracle.State(d, B)
    zeros = np.zeros((B, 80), np.float32)
    for i in range(699):
        oracle.step(d, blob, st, seedwave[:, i], zeros, gc)
    want2 = oracle.generate_mol(d, blob, st, oracle.upsample(d, blob, mel)[:, :300], gc, seedwave[:, -1], u[:, :300])
    m.queue_initializer()
    m.prime(seedwave[:, :699], None, gc)
    got2 = m.generate(m.create_upsample(mel)[:, :300].contiguous(), gc, seedwave[:, -1], u[:, :300]).cpu().numpy()
    assert first_mismatch(got2, want2) is None, first_mismatch(got2, want2)
    assert rf < 700


def test_generate_c1_config(torch_cuda, oracle):
    """BASELINE configs[0]: mu-law-256, default 50 layers, 54 mel frames x 300 = 16 200 samples (about 1 s at 16 kHz), B=1"""
    dil = [2 ** i for i in range(10)] * 5
    d, blob, want, got, _ = _run_onehot(oracle, dil, 1, 54 * 300)
    got = got.cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (1, 16200)
    assert np.array_equal(got, want), first_mismatch(got, want)
    assert got.min() >= 0 and got.max() <= 255


def test_generate_c2_mulaw_variant_at_batch_8(torch_cuda, oracle):
    """the mu-law-256 variant of BASELINE configs[1] (30 layers = 3 x [1..512], S = 512, gc + lc, B = 8) -- the model
    north_star's integer-parity bar is stated on.  12 000 steps per stream (0.5 s at 24 kHz): int32 category ids equal to the
    checker's, element for element (model.py:223-227,243 float64 softmax, generate.py:219-231 temperature + legacy choice)"""
    dil = [2 ** i for i in range(10)] * 3
    oracle.set_threads(min(8, oracle.set_threads(1)))
    try:
        d, blob, want, got, _ = _run_onehot(oracle, dil, 8, 12000)
    finally:
        oracle.set_threads(1)
    got = got.cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (8, 12000)
    assert np.array_equal(got, want), first_mismatch(got, want)
    assert len(np.unique(got)) > 200                     # the streams really walk the whole alphabet


@pytest.mark.parametrize("kw", [dict(groups=1), dict(groups=2, Q=64), dict(L=0, G=0, Q=16)])
def test_generate_onehot_variants(torch_cuda, oracle, kw):
    d, blob, want, got, _ = _run_onehot(oracle, [1, 2, 4, 8, 16], 3, 50, S=128, scale=0.3, **kw)
    assert np.array_equal(got.cpu().numpy(), want), kw


def _onehot_xcd_case(oracle, nl, B, T, use_bias=True, G=32, L=80, temperature=1.0, seed=1):
    dil = ([2 ** i for i in range(10)] * 5)[:nl]
    d, tensors, blob = make_case(oracle, dil, scalar_input=False, S=512, Q=256, scale=0.12, use_bias=use_bias, G=G, L=L)
    m = make_model(B, dil, tensors, scalar_input=False, S=512, Q=256, use_bias=use_bias, G=G, L=L)
    rng = np.random.RandomState(seed)
    U = rng.uniform(-4, 4, (B, T, L)).astype(np.float32) if L else None
    gc = (np.arange(B) % 2).astype(np.int32) if G else None
    seed_in = rng.randint(256, size=B).astype(np.int32)
    u = np.random.RandomState(seed + 1).random_sample((B, T))
    return d, blob, m, U, gc, seed_in, u


@pytest.mark.parametrize("nl,B,kw", [(30, 8, {}), (30, 1, {}), (7, 3, dict(use_bias=False)), (12, 9, dict(G=0)), (30, 19, dict(L=0)),
                                     (30, 32, {}), (1, 2, {}), (33, 2, {}), (50, 9, {})])
def test_xcd_onehot_kernel_shapes(torch_cuda, oracle, nl, B, kw):
    """the one-hot mu-law-256 model (scalar_input False; model.py:41-46,223-227,243, generate.py:219-231) on the
    XCD-per-stream kernel: int32 class ids equal to the checker's element for element -- one to four streams per XCD, 1..50 layers
    (second chain workgroup above 30), without bias / gc / lc, temperature 0.8 on one case."""
    T = 160 if B <= 9 else 90
    temp = 0.8 if (nl, B) == (30, 1) else 1.0
    d, blob, m, U, gc, seed_in, u = _onehot_xcd_case(oracle, nl, B, T, temperature=temp, **kw)
    if kw.get("L", 80):
        assert m.fused_conditioning(), "the one-hot model must take the XCD kernel at S = 512, Q = 256"
    oracle.set_threads(min(8, oracle.set_threads(1)))
    try:
        want = oracle.generate_mulaw(d, blob, oracle.State(d, B), U, gc, seed_in, u, temp)
    finally:
        oracle.set_threads(1)
    got = m.generate(U, gc, seed_in, u, temperature=temp).cpu().numpy()
    assert got.dtype == np.int32
    assert np.array_equal(got, want), first_mismatch(got, want)


def test_xcd_onehot_kernel_layer_dumps_and_chunked_calls(torch_cuda, oracle):
    """per-layer z / x and the 256 logits of the first steps bit for bit (debug build of the one-hot XCD kernel); then the same
    utterance in calls of 1, 7, 50 and the remaining steps: the state (delay lines, the causal queue's previous class) carries over"""
    nl, B, T, dbg = 30, 3, 120, 3
    d, blob, m, U, gc, seed_in, u = _onehot_xcd_case(oracle, nl, B, T, seed=5)
    want = oracle.generate_mulaw(d, blob, oracle.State(d, B), U, gc, seed_in, u, 1.0)
    got, dump = m.generate(U, gc, seed_in, u, debug_steps=dbg)
    dump = dump.cpu().numpy()
    st = oracle.State(d, B)
    inp = seed_in.copy()
    for t in range(dbg):
        raw, dz, dx = oracle.step(d, blob, st, inp, U[:, t], gc, debug=True)
        gz = dump[:, t, :nl * 64].reshape(B, nl, 2, 32)
        assert first_mismatch(gz[:, :, 0], dz) is None, ("z", t)
        assert first_mismatch(gz[:, :, 1], dx) is None, ("x", t)
        assert first_mismatch(dump[:, t, nl * 64:nl * 64 + 256], raw) is None, ("logits", t)
        inp = want[:, t]
    assert np.array_equal(got.cpu().numpy(), want)
    m.queue_initializer()
    outs, fi, p = [], seed_in, 0
    for n in (1, 7, 50, T - 58):
        o = m.generate(U[:, p:p + n].copy(), gc, fi, u[:, p:p + n].copy()).cpu().numpy()
        outs.append(o); fi = o[:, -1]; p += n
    assert np.array_equal(np.concatenate(outs, axis=1), want)


@pytest.mark.parametrize("nl,B", [(30, 3), (41, 2)])
def test_xcd_onehot_kernel_priming_then_generation(torch_cuda, oracle, nl, B):
    """generate.py:168-180 on the one-hot XCD kernel: teacher-forced class ids (zero lc), then generation from the primed queues"""
    T, n_prime = 60, 700
    d, blob, m, U, gc, seed_in, u = _onehot_xcd_case(oracle, nl, B, T, seed=7)
    rng = np.random.RandomState(17)
    seedwave = rng.randint(256, size=(B, n_prime)).astype(np.int32)
    st = oracle.State(d, B)
    zeros = np.zeros((B, 80), np.float32)
    for i in range(n_prime - 1):
        oracle.step(d, blob, st, seedwave[:, i], zeros, gc)
    want = oracle.generate_mulaw(d, blob, st, U, gc, seedwave[:, -1], u, 1.0)
    m.prime(seedwave[:, :n_prime - 1], None, gc)
    got = m.generate(U, gc, seedwave[:, -1], u).cpu().numpy()
    assert np.array_equal(got, want), first_mismatch(got, want)


@pytest.mark.parametrize("scalar", [True, False])
def test_priming_then_generation(torch_cuda, oracle, scalar):
    """generate.py:168-180: prime with RF-1 seed samples (zero lc, predictions discarded), then generate"""
    dil = [1, 2, 4, 8, 16]
    B, T = 2, 30
    d, tensors, blob = make_case(oracle, dil, scalar_input=scalar, S=128, Q=64, scale=0.25)
    m = make_model(B, dil, tensors, scalar_input=scalar, S=128, Q=64)
    rf = oracle.receptive_field(d)
    rng = np.random.RandomState(9)
    seedwave = rng.uniform(-1, 1, (B, rf)).astype(np.float32) if scalar else rng.randint(64, size=(B, rf)).astype(np.int32)
    U = rng.uniform(-4, 4, (B, T, 80)).astype(np.float32)
    gc = np.array([0, 1], np.int32)
    st = oracle.State(d, B)
    zeros = np.zeros((B, 80), np.float32)
    for i in range(rf - 1):                                   # generate.py:177-180
        oracle.step(d, blob, st, seedwave[:, i], zeros, gc)
    if scalar:
        u = mol_uniforms(B, T, 10)
        want = oracle.generate_mol(d, blob, st, U, gc, seedwave[:, -1], u)
    else:
        u = np.random.RandomState(3).random_sample((B, T))
        want = oracle.generate_mulaw(d, blob, st, U, gc, seedwave[:, -1], u, 1.0)
    m.prime(seedwave[:, :rf - 1], None, gc)
    got = m.generate(U, gc, seedwave[:, -1], u).cpu().numpy()
    assert first_mismatch(got, want) is None


@pytest.mark.parametrize("B", [3, 11])
def test_xcd_kernel_priming_then_generation(torch_cuda, oracle, B):
    """generate.py:168-180 on the XCD kernel (B = 11: XCDs 0-2 carry two streams, the others one): prime with RF-1 seed samples
    (zero lc, predictions discarded; the skip / conv1 workgroups idle), then generate from mel frames"""
    dil = [1, 2, 4, 8, 16, 32]
    T = 600
    d, tensors, blob = make_case(oracle, dil, scale=0.1)
    m = make_model(B, dil, tensors)
    assert m.fused_conditioning()
    rf = oracle.receptive_field(d)
    rng = np.random.RandomState(9)
    seedwave = rng.uniform(-1, 1, (B, rf)).astype(np.float32)
    mel = rng.uniform(-4, 4, (B, 2, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    st = oracle.State(d, B)
    zeros = np.zeros((B, 80), np.float32)
    for i in range(rf - 1):                                   # generate.py:177-180
        oracle.step(d, blob, st, seedwave[:, i], zeros, gc)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d, blob, st, oracle.upsample(d, blob, mel), gc, seedwave[:, -1], u)
    m.prime(seedwave[:, :rf - 1], None, gc)
    got = m.generate(m.create_upsample(mel), gc, seedwave[:, -1], u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("nl,use_bias,G,L,O", [(1, True, 32, 80, 30), (5, True, 32, 80, 30), (9, False, 32, 80, 30), (24, True, 0, 80, 30),
                                             (25, True, 32, 0, 30), (28, False, 0, 0, 3), (30, True, 32, 80, 6), (35, False, 0, 0, 3),
                                             (44, True, 0, 80, 30), (47, True, 32, 0, 9)])
def test_xcd_kernel_shapes(torch_cuda, oracle, nl, use_bias, G, L, O):
    """the XCD kernel away from the bench shape: layer counts that end inside / at the edge of a chain wave, no biases, no global /
    local conditioning (the chain's general instantiation), other mixture sizes; B = 3 leaves five XCDs idle"""
    dil = ([1, 2, 4, 8, 16, 32, 64] * 5)[:nl]
    B, T = 3, 450
    d, tensors, blob = make_case(oracle, dil, use_bias=use_bias, G=G, L=L, out_channels=O, scale=0.1)
    m = make_model(B, dil, tensors, use_bias=use_bias, G=G, L=L, out_channels=O)
    assert m.fused_conditioning() == bool(L)
    rng = np.random.RandomState(nl)
    mel = rng.uniform(-4, 4, (B, 2, 80)).astype(np.float32) if L else None
    gc = (np.arange(B) % 2).astype(np.int32) if G else None
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, O // 3)
    want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel)[:, :T] if L else None, gc, seed_in, u)
    got = m.generate(m.create_upsample(mel) if L else None, gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("nl,B", [(31, 3), (33, 2), (41, 9), (50, 2), (50, 16)])
def test_xcd_kernel_more_than_30_layers(torch_cuda, oracle, nl, B):
    """hparams.py's default stack has 50 layers: a second chain workgroup takes layers 30.., the service / skip waves keep the tiles
    of the early layers in LDS, two lc layers per wave; B = 9 and 16 put two streams on an XCD.  Chunked calls, bit for bit."""
    dil = ([2 ** i for i in range(10)] * 5)[:nl]
    d, tensors, blob = make_case(oracle, dil, seed=nl)
    m = make_model(B, dil, tensors)
    assert m.fused_conditioning(), "up to 50 layers are served by the XCD kernel on an MI355X"
    rng = np.random.RandomState(nl + B)
    T = 900
    mel = rng.uniform(-4, 4, (B, 3, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    oracle.set_threads(min(B, oracle.set_threads(1)))
    try:
        want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel), gc, seed_in, u)
    finally:
        oracle.set_threads(1)
    U = m.create_upsample(mel)
    a = m.generate(U[:, :600].contiguous(), gc, seed_in, u[:, :600]).cpu().numpy()
    b = m.generate(U[:, 600:].contiguous(), gc, a[:, -1], u[:, 600:]).cpu().numpy()
    got = np.concatenate([a, b], axis=1)
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("nl", [31, 50])
def test_xcd_kernel_priming_two_chain_workgroups(torch_cuda, oracle, nl):
    """teacher-forced steps through both chain workgroups (the end of a step travels back to the head through L2)"""
    dil = ([2 ** i for i in range(6)] * 9)[:nl]
    B, T = 2, 300
    d, tensors, blob = make_case(oracle, dil, scale=0.1)
    m = make_model(B, dil, tensors)
    assert m.fused_conditioning()
    rf = oracle.receptive_field(d)
    rng = np.random.RandomState(4)
    seedwave = rng.uniform(-1, 1, (B, rf)).astype(np.float32)
    mel = rng.uniform(-4, 4, (B, 1, 80)).astype(np.float32)
    gc = np.array([1, 0], np.int32)
    st = oracle.State(d, B)
    zeros = np.zeros((B, 80), np.float32)
    for i in range(rf - 1):
        oracle.step(d, blob, st, seedwave[:, i], zeros, gc)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d, blob, st, oracle.upsample(d, blob, mel), gc, seedwave[:, -1], u)
    m.prime(seedwave[:, :rf - 1], None, gc)
    got = m.generate(m.create_upsample(mel), gc, seedwave[:, -1], u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


def test_xcd_kernel_layer_dumps_need_the_30_layer_kernel(torch_cuda, oracle):
    """the 31-50 layer instantiation has no instrumented build: asking for layer dumps fails loudly instead of returning nothing"""
    import twvk_amd
    dil = [1, 2, 4, 8] * 8
    d, tensors, blob = make_case(oracle, dil)
    m = make_model(1, dil, tensors)
    mel = np.zeros((1, 1, 80), np.float32)
    with pytest.raises(twvk_amd._lib.TwvError, match="30-layer"):
        m.generate(m.create_upsample(mel), np.zeros(1, np.int32), np.zeros(1, np.float32), mol_uniforms(1, 8, 10), debug_steps=2)
    m0 = make_model(1, dil, tensors, xcd=0)                       # the generic kernel serves the request
    out, dump = m0.generate(m0.create_upsample(mel)[:, :8].contiguous(), np.zeros(1, np.int32), np.zeros(1, np.float32), mol_uniforms(1, 8, 10), debug_steps=2)
    assert np.isfinite(dump.cpu().numpy()).all()


def test_generate_cli(torch_cuda, tmp_path):
    """generate.py surface: flags, params.json override, output files (generate.py:52-69,109,261)"""
    import json
    from scipy.io import wavfile
    import twvk_amd
    from twvk_amd.generate import main
    ck = tmp_path / "ckpt"; ck.mkdir()
    json.dump({"dilations": [1, 2, 4, 8, 16, 32], "skip_channels": 128}, open(ck / "params.json", "w"))
    mel = np.random.RandomState(0).uniform(-4, 4, (2, 80)).astype(np.float32)
    np.save(tmp_path / "mel.npy", mel)
    paths = main([str(ck), "--mel", str(tmp_path / "mel.npy"), "--gc_cardinality", "2", "--gc_id", "1", "--batch_size", "2",
                  "--logdir", str(tmp_path / "log"), "--seed", "3", "--random_init"])
    assert len(paths) == 2 and paths[0].endswith("test-0.wav")
    rate, data = wavfile.read(paths[1])
    assert rate == 24000 and data.dtype == np.int16 and data.shape == (600,) and np.abs(data).max() == 32767
    with pytest.raises(ValueError):
        main([str(ck), "--mel", str(tmp_path / "mel.npy"), "--random_init"])       # gc_cardinality required (generate.py:72-77)
    # restore defaults for the other tests (hparams is a module singleton, like the reference's)
    d = twvk_amd.default_hparams()
    twvk_amd.hparams.__dict__.update(d.__dict__)


# ---------------------------------------------------------------- edge cases
def test_generate_single_step_and_ragged_lengths(torch_cuda, oracle):
    """T = 1 (one sample), then odd lengths that are not a multiple of anything"""
    dil = [1, 2, 4, 8]
    for T in (1, 3, 37):
        d, blob, m, want, got, _ = _run_mol(oracle, torch_cuda, dil, 2, 1, T=T, S=64, scale=0.2)
        assert got.shape == (2, T)
        assert first_mismatch(got.cpu().numpy(), want) is None, T


def test_generate_large_batch(torch_cuda, oracle):
    """B = 40 streams: groups falls back so that all stream workgroups stay co-resident (B*G <= CUs)"""
    dil = [1, 2, 4]
    d, blob, m, want, got, _ = _run_mol(oracle, torch_cuda, dil, 40, 1, T=12, S=64, scale=0.2)
    assert first_mismatch(got.cpu().numpy(), want) is None


def test_generate_rejects_bad_arguments(torch_cuda, oracle):
    import twvk_amd
    from twvk_amd._lib import TwvError
    from helpers import make_case, make_model, mol_uniforms
    dil = [1, 2]
    d, tensors, blob = make_case(oracle, dil, S=64)
    m = make_model(2, dil, tensors, S=64)
    up = m.create_upsample(np.zeros((2, 1, 80), np.float32))
    with pytest.raises(Exception):
        m.generate(up[:, :10].contiguous(), [0, 1], np.zeros(2, np.float32), mol_uniforms(2, 12, 10))      # lc shorter than T
    with pytest.raises(TwvError):
        make_model(2, dil, tensors, S=96)                                                                  # skip_channels not a multiple of 64
    with pytest.raises(ValueError):
        make_model(2, dil, {k: v for k, v in list(tensors.items())[1:]}, S=64)                             # a checkpoint tensor is missing


def test_wav_to_int16_matches_numpy_save_wav(torch_cuda):
    """utils/audio.py:14-17: wav *= 32767 / max(0.01, max|wav|); astype(int16) -- bit for bit, including the quiet-signal floor"""
    from twvk_amd.ops import wav_to_int16
    rng = np.random.RandomState(3)
    for scale in (0.7, 1.0, 0.003, 0.0):
        wav = (rng.randn(3, 10001) * 0.2 * scale).astype(np.float32)
        got = wav_to_int16(wav).cpu().numpy()
        for i in range(3):
            ref = wav[i].copy()
            ref *= 32767 / max(0.01, np.max(np.abs(ref)))
            assert np.array_equal(got[i], ref.astype(np.int16)), (scale, i)


# ---------------------------------------------------------------------------------------------------------------------
#  parity AT THE BENCHMARKED LAUNCH GEOMETRY (BASELINE configs[1]: B = 8 streams, 30 layers): every workgroup role, every XCD,
#  epoch counters far past the short tests.  The checker runs one stream per host thread (same arithmetic per stream).
# ---------------------------------------------------------------------------------------------------------------------
def _bench_case(oracle, B, T, xcd=None, groups=None):
    dil = [2 ** i for i in range(10)] * 3
    d, tensors, blob = make_case(oracle, dil)
    m = make_model(B, dil, tensors, xcd=xcd, groups=groups)
    rng = np.random.RandomState(1)
    Tm = (T + 299) // 300
    mel = rng.uniform(-4, 4, (B, Tm, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    oracle.set_threads(min(B, oracle.set_threads(1)))
    try:
        want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel)[:, :T], gc, seed_in, u)
    finally:
        oracle.set_threads(1)
    return m, mel, gc, seed_in, u, want


def test_xcd_kernel_at_bench_geometry(torch_cuda, oracle):
    """the XCD-per-stream kernel as bench.py launches it: B = 8 (one stream per XCD, 22 workgroups each), fused conditioning
    (create_upsample + lc projections inside the launch), 24 000 steps = 1 s of audio per stream; bit for bit"""
    B, T = 8, 24000
    m, mel, gc, seed_in, u, want = _bench_case(oracle, B, T)
    assert m.fused_conditioning(), "this configuration must be served by the XCD-per-stream kernel on an MI355X"
    got = m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("B", [16, 32])
def test_xcd_kernel_with_several_streams_per_xcd(torch_cuda, oracle, B):
    """the stream sweep of bench.py: 2 and 4 streams per XCD (own chain + service workgroup each, ONE set of skip / conv1 / lc
    workgroups per XCD serving them in turn), 12 000 steps; bit for bit"""
    T = 12000
    m, mel, gc, seed_in, u, want = _bench_case(oracle, B, T)
    m.set_option("xcd_many", 2)           # the batch <= 32 kernel (since round 5 the library takes the many-streams kernel from batch 21 on)
    assert m.fused_conditioning(), "up to 32 streams are served by the XCD kernel on an MI355X"
    got = m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("B", [48, 64, 96])
def test_xcd_many_streams_kernel_at_bench_geometry(torch_cuda, oracle, B):
    """more than 32 streams: the many-streams kernel (five to twelve streams per XCD: two per chain / service workgroup, skip
    workgroups laid out as layer groups x output halves with the layer-ordered sum as a relay), BASELINE configs[1]'s model,
    12 000 steps per stream (6 000 at B = 96: all 32 CUs of every XCD carry a role) with fused conditioning; every sample of
    every stream bit for bit"""
    T = 12000 if B <= 64 else 6000
    m, mel, gc, seed_in, u, want = _bench_case(oracle, B, T)
    assert m.fused_conditioning(), "up to 64 streams are served by the XCD kernels on an MI355X"
    got = m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("B", [1, 9, 33, 43, 64, 75, 96])
def test_xcd_many_streams_kernel_chunked_calls(torch_cuda, oracle, B):
    """the many-streams kernel (forced with the `xcd_many` option where the batch alone would not select it) against the checker:
    stream counts that leave chain workgroups with one slot, XCDs with different numbers of streams, chunked calls including
    single-step launches (the state -- delay lines, causal queue, last lc frame -- carries over), materialised upsampled rows"""
    dil = [2 ** i for i in range(10)] * 3
    d, tensors, blob = make_case(oracle, dil, seed=5)
    rng = np.random.RandomState(9 + B)
    T = 640
    mel = rng.uniform(-4, 4, (B, 3, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    oracle.set_threads(min(B, oracle.set_threads(1)))
    try:
        want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel)[:, :T], gc, seed_in, u)
    finally:
        oracle.set_threads(1)
    m = make_model(B, dil, tensors, xcd_many=1)
    U = m.create_upsample(mel).tensor()
    outs, fi, p = [], seed_in, 0
    # (launch lengths on both sides of the kernel's lc ring of 8 steps: a launch of at most ring + 1 steps waits for the previous launch's
    # last frame to be consumed before it overwrites the slot)
    for n in ((300, 1, 1, 2, 336) if B not in (33, 96) else (290, 1, 1, 2, 8, 9, 10, 16, 17, 286)):
        o = m.generate(U[:, p:p + n].contiguous(), gc, fi, u[:, p:p + n]).cpu().numpy()
        outs.append(o); fi = o[:, -1]; p += n
    assert p == T
    got = np.concatenate(outs, axis=1)
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("nl,use_bias,G,L,O", [(1, True, 32, 80, 30), (7, True, 32, 80, 30), (9, False, 32, 80, 30), (17, True, 0, 80, 30),
                                             (25, True, 32, 0, 30), (28, False, 0, 0, 3), (30, True, 32, 80, 6)])
def test_xcd_many_streams_kernel_shapes(torch_cuda, oracle, nl, use_bias, G, L, O):
    """the many-streams kernel away from the bench shape: layer counts that end inside a chain wave / inside a skip layer group
    (one group, a short first group), no biases, no global / local conditioning, other mixture sizes; B = 43 gives the XCDs five
    and six streams; with layer dumps for the first steps"""
    dil = ([1, 2, 4, 8, 16, 32, 64] * 5)[:nl]
    B, T, dbg = 43, 450, 2
    d, tensors, blob = make_case(oracle, dil, use_bias=use_bias, G=G, L=L, out_channels=O, scale=0.1)
    m = make_model(B, dil, tensors, use_bias=use_bias, G=G, L=L, out_channels=O)
    assert m.fused_conditioning() == bool(L)
    rng = np.random.RandomState(nl)
    mel = rng.uniform(-4, 4, (B, 2, 80)).astype(np.float32) if L else None
    gc = (np.arange(B) % 2).astype(np.int32) if G else None
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, O // 3)
    U_o = oracle.upsample(d, blob, mel)[:, :T] if L else None
    oracle.set_threads(min(B, oracle.set_threads(1)))
    try:
        want = oracle.generate_mol(d, blob, oracle.State(d, B), U_o, gc, seed_in, u)
    finally:
        oracle.set_threads(1)
    got, dump = m.generate(m.create_upsample(mel) if L else None, gc, seed_in, u, debug_steps=dbg)
    dump = dump.cpu().numpy()
    st = oracle.State(d, B)
    inp = seed_in.copy()
    for t in range(dbg):
        raw, dz, dx = oracle.step(d, blob, st, inp, U_o[:, t] if L else None, gc, debug=True)
        gz = dump[:, t, :nl * 64].reshape(B, nl, 2, 32)
        assert first_mismatch(gz[:, :, 0], dz) is None, ("z", t, first_mismatch(gz[:, :, 0], dz))
        assert first_mismatch(gz[:, :, 1], dx) is None, ("x", t, first_mismatch(gz[:, :, 1], dx))
        assert first_mismatch(dump[:, t, nl * 64:nl * 64 + d.O], raw) is None, ("raw", t)
        inp = want[:, t]
    assert first_mismatch(got.cpu().numpy(), want) is None, first_mismatch(got.cpu().numpy(), want)


@pytest.mark.parametrize("B", [11, 40])
def test_xcd_many_streams_kernel_priming_then_generation(torch_cuda, oracle, B):
    """generate.py:168-180 on the many-streams kernel: RF-1 teacher-forced steps through chain workgroups that carry two slots (the
    skip / conv1 workgroups idle, the end of a slot's step releases its next one), then generation from mel frames"""
    dil = [1, 2, 4, 8, 16, 32]
    T = 600
    d, tensors, blob = make_case(oracle, dil, scale=0.1)
    m = make_model(B, dil, tensors, xcd_many=1)
    rf = oracle.receptive_field(d)
    rng = np.random.RandomState(9)
    seedwave = rng.uniform(-1, 1, (B, rf)).astype(np.float32)
    mel = rng.uniform(-4, 4, (B, 2, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    st = oracle.State(d, B)
    zeros = np.zeros((B, 80), np.float32)
    for i in range(rf - 1):
        oracle.step(d, blob, st, seedwave[:, i], zeros, gc)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d, blob, st, oracle.upsample(d, blob, mel), gc, seedwave[:, -1], u)
    m.prime(seedwave[:, :rf - 1], None, gc)
    got = m.generate(m.create_upsample(mel), gc, seedwave[:, -1], u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


def test_busy_device_is_a_clean_error_not_a_hang(torch_cuda, oracle):
    """a persistent kernel's roles spin on each other, so all of them must be resident.  With another kernel holding half of every
    XCD's CUs (twv_debug_occupy on a second stream) the role workgroups cannot all start: the launch must come back quickly with
    TWV_E_BUSY -- no hang, no samples written, the state untouched -- and the same call succeeds bit for bit once the device is free"""
    import ctypes as C
    import time
    import twvk_amd
    from twvk_amd._lib import TwvError
    torch = torch_cuda
    B, T = 8, 600
    m, mel, gc, seed_in, u, want = _bench_case(oracle, B, T)
    side = torch.cuda.Stream()
    # 128 one-wave workgroups x 100 KB of LDS for 400 ms: one per CU on half of the chip (the generation kernel's workgroups need a whole CU's LDS)
    twvk_amd._lib.check(m._L.twv_debug_occupy(128, 100 * 1024, 400.0, C.c_void_p(side.cuda_stream)))
    time.sleep(0.02)
    t0 = time.perf_counter()
    m.BUSY_RETRIES = 0                                                               # the raw behaviour first: one launch, one refusal
    with pytest.raises(TwvError, match="busy"):
        m.generate(m.create_upsample(mel), gc, seed_in, u)
    assert time.perf_counter() - t0 < 0.35, "the refusal must come from the start-up check (~50 ms), not from a watchdog"
    torch.cuda.synchronize()
    got = m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy()           # no queue_initializer in between: the state was not touched
    assert first_mismatch(got, want) is None, first_mismatch(got, want)
    # generate()'s bounded retry: the occupier leaves after ~150 ms, the third or fourth launch goes through -- same samples
    m.queue_initializer()
    m.BUSY_RETRIES = 6
    twvk_amd._lib.check(m._L.twv_debug_occupy(128, 100 * 1024, 150.0, C.c_void_p(side.cuda_stream)))
    time.sleep(0.02)
    got = m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


def test_kernel_options_resize_the_state_buffer(torch_cuda, oracle):
    """the XCD kernels append their exchange area to the state buffer, so switching the kernel selection after
    load_weights() must re-create the buffer (it used to be kept: out-of-bounds device writes); either order gives the same samples"""
    dil = [1, 2, 4, 8, 16, 32]
    d, tensors, blob = make_case(oracle, dil, scale=0.1)
    B, T = 3, 300
    rng = np.random.RandomState(2)
    mel = rng.uniform(-4, 4, (B, 1, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel), gc, seed_in, u)
    m = make_model(B, dil, tensors, xcd=0)                     # sized for the generic kernel ...
    n0 = m._state.numel()
    m.set_option("xcd", 1)                                     # ... then the XCD kernel is selected
    assert m._state.numel() > n0 and m.fused_conditioning()
    assert first_mismatch(m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy(), want) is None
    m.set_option("xcd_many", 1)
    assert first_mismatch(m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy(), want) is None
    m.set_option("xcd", 0)
    assert m._state.numel() == n0 and not m.fused_conditioning()
    up = m.create_upsample(mel)
    assert first_mismatch(m.generate(up, gc, seed_in, u).cpu().numpy(), want) is None


def test_lazy_upsampled_handle_behaves_like_the_tensor(torch_cuda, oracle):
    """create_upsample under fused conditioning returns a handle; everything but generate() sees the (B, T*hop, lc) tensor"""
    dil = [1, 2, 4, 8]
    d, tensors, blob = make_case(oracle, dil, scale=0.1)
    m = make_model(2, dil, tensors)
    mel = np.random.RandomState(1).uniform(-4, 4, (2, 3, 80)).astype(np.float32)
    up = m.create_upsample(mel)
    want = oracle.upsample(d, blob, mel)
    if not m.fused_conditioning():
        pytest.skip("needs the XCD-per-stream kernel")
    assert up.shape == (2, 900, 80) and len(up) == 2
    assert first_mismatch(np.asarray(up), want) is None
    assert first_mismatch(up.float().cpu().numpy(), want) is None
    assert first_mismatch(torch_cuda.cat([up, up], dim=1)[:, 900:].cpu().numpy(), want) is None
    assert first_mismatch((up * 1.0)[:, :5].cpu().numpy(), want[:, :5]) is None
    # chunked generation: rows_from() keeps the fused path at hop boundaries
    gc = np.array([0, 1], np.int32)
    seed_in = np.zeros(2, np.float32)
    u = mol_uniforms(2, 900, 10)
    ref = oracle.generate_mol(d, blob, oracle.State(d, 2), want, gc, seed_in, u)
    a = m.generate(up, gc, seed_in, u[:, :600]).cpu().numpy()
    b = m.generate(up.rows_from(600), gc, a[:, -1], u[:, 600:]).cpu().numpy()
    assert first_mismatch(np.concatenate([a, b], axis=1), ref) is None


def test_generic_kernel_at_bench_geometry(torch_cuda, oracle):
    """the generic kernel at B = 8: 8 workgroups per stream + helper workgroups (64 + 64 co-resident), 24 000 steps; bit for bit"""
    B, T = 8, 24000
    m, mel, gc, seed_in, u, want = _bench_case(oracle, B, T, xcd=0)
    assert not m.fused_conditioning()
    got = m.generate(m.create_upsample(mel)[:, :T].contiguous(), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


def test_xcd_kernel_full_length_utterances(torch_cuda, oracle):
    """BASELINE configs[1] in full: 8 utterances x 8 s = 192 000 steps each, against the checker (a minute or two of host time)"""
    B, T = 8, 192000
    m, mel, gc, seed_in, u, want = _bench_case(oracle, B, T)
    got = m.generate(m.create_upsample(mel), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)


@pytest.mark.parametrize("B", [1, 3, 8, 9, 19, 32])
def test_xcd_kernel_matches_generic_kernel(torch_cuda, oracle, B):
    """both kernels, same inputs, several stream counts (B < 8 leaves XCDs idle; B = 9 and 19 give the XCDs different numbers of
    streams), chunked calls with state carried over"""
    dil = [2 ** i for i in range(10)] * 3
    d, tensors, blob = make_case(oracle, dil, seed=5)
    rng = np.random.RandomState(9)
    T = 900
    mel = rng.uniform(-4, 4, (B, 3, 80)).astype(np.float32)
    gc = (np.arange(B) % 2).astype(np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    outs = []
    for xcd in (1, 0):
        m = make_model(B, dil, tensors, xcd=xcd, xcd_many=2)      # 2: the batch <= 32 XCD kernel also where the library would take the many-streams one
        U = m.create_upsample(mel)
        a = m.generate(U[:, :500].contiguous(), gc, seed_in, u[:, :500]).cpu().numpy()
        b = m.generate(U[:, 500:].contiguous(), gc, a[:, -1], u[:, 500:]).cpu().numpy()
        outs.append(np.concatenate([a, b], axis=1))
    assert first_mismatch(outs[0], outs[1]) is None, first_mismatch(outs[0], outs[1])
    want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel), gc, seed_in, u)
    assert first_mismatch(outs[0], want) is None, first_mismatch(outs[0], want)


def test_fused_conditioning_equals_materialised_upsample(torch_cuda, oracle):
    """create_upsample inside the launch (mel rows) == the stand-alone upsampling kernel + rows handed over, same kernel otherwise"""
    dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    d, tensors, blob = make_case(oracle, dil, seed=2)
    B, T = 2, 1200
    rng = np.random.RandomState(3)
    mel = rng.uniform(-4, 4, (B, 4, 80)).astype(np.float32)
    gc = np.array([1, 0], np.int32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    m = make_model(B, dil, tensors)
    if not m.fused_conditioning():
        pytest.skip("needs the XCD-per-stream kernel")
    lazy = m.create_upsample(mel)
    a = m.generate(lazy, gc, seed_in, u).cpu().numpy()
    m.queue_initializer()
    b = m.generate(lazy.tensor(), gc, seed_in, u).cpu().numpy()
    assert first_mismatch(a, b) is None
    want = oracle.generate_mol(d, blob, oracle.State(d, B), oracle.upsample(d, blob, mel), gc, seed_in, u)
    assert first_mismatch(a, want) is None


@pytest.mark.parametrize("xcd", [1, 0])
def test_global_condition_passed_as_embedding(torch_cuda, oracle, xcd):
    """_embed_gc's second branch (model.py:199-207): no cardinality, the caller passes the (B, gc_channels) embedding itself.
    Feeding the rows a gc_embedding table would have produced gives the id-path's samples bit for bit; both kernels."""
    dil = [1, 2, 4, 8, 16, 32]
    B, T = 2, 300
    d_id, tensors_id, blob_id = make_case(oracle, dil, gc_card=3, seed=4)
    table = tensors_id["wavenet/gc_embedding"]
    ids = np.array([2, 0], np.int32)
    rng = np.random.RandomState(6)
    mel = rng.uniform(-4, 4, (B, 1, 80)).astype(np.float32)
    seed_in = (2 * rng.rand(B) - 1).astype(np.float32)
    u = mol_uniforms(B, T, 10)
    want = oracle.generate_mol(d_id, blob_id, oracle.State(d_id, B), oracle.upsample(d_id, blob_id, mel), ids, seed_in, u)
    # the same weights without the table
    d_e = oracle.make_dims(dil, gc_card=0)
    tensors_e = {k: v for k, v in tensors_id.items() if k != "wavenet/gc_embedding"}
    blob_e = oracle.blob_from_tensors(d_e, tensors_e)
    emb = table[ids]
    assert "wavenet/gc_embedding" not in dict(oracle.tensor_specs(d_e))
    want_e = oracle.generate_mol(d_e, blob_e, oracle.State(d_e, B), oracle.upsample(d_e, blob_e, mel), emb, seed_in, u)
    assert first_mismatch(want_e, want) is None                      # the checker's two branches agree
    m = make_model(B, dil, tensors_e, gc_card=0, xcd=xcd)
    got = m.generate(m.create_upsample(mel), emb, seed_in, u).cpu().numpy()
    assert first_mismatch(got, want) is None, first_mismatch(got, want)
    with pytest.raises(ValueError):
        m.generate(m.create_upsample(mel), emb[:, :5], seed_in, u)   # wrong embedding width (model.py:203-206)
