"""The config / CLI surface held to the reference's own source text -- build-container only (skipped where /root/reference is absent,
i.e. on the GPU box): hparams.py:6-176,190-192 and the three argparse blocks (generate.py:51-79, synthesizer.py:371-380,
train_vocoder.py:34-58) are `ast`-parsed -- nothing of the reference is imported or executed, no stand-in modules -- and every name,
default, type and `required` flag is compared with this repo's.  The only differences allowed are the ones listed (and explained) here.
"""
import ast
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (build container only)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tacotron-wavenet-vocoder-korean_amd")


def _const(node, env=None):
    """Literal value of a small expression: constants, lists / tuples, unary minus, + - * /, names bound earlier in the same block."""
    env = env or {}
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, (ast.List, ast.Tuple)):
        return [_const(e, env) for e in node.elts]
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        return -_const(node.operand, env)
    if isinstance(node, ast.BinOp):
        a, b = _const(node.left, env), _const(node.right, env)
        if isinstance(node.op, ast.Add): return a + b
        if isinstance(node.op, ast.Sub): return a - b
        if isinstance(node.op, ast.Mult): return a * b
        if isinstance(node.op, ast.Div): return a / b
    if isinstance(node, ast.Name) and node.id in env:
        return env[node.id]
    if isinstance(node, ast.Name):
        return ("name", node.id)                      # a callable (type=...) or an unresolved symbol: compared by its spelling
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "str":
        return str(_const(node.args[0], env))
    raise ValueError("not a literal: " + ast.dump(node))


def _parse(path):
    with open(path, encoding="utf-8") as fh:
        return ast.parse(fh.read(), filename=path)


def _argparse_surface(path):
    """{flag: {type, default, required, action}} of every parser.add_argument(...) call in the file, in source order; module- or
    function-level NAME = literal assignments in front of a call are resolved (the reference writes `BATCH_SIZE = 1` first)."""
    tree = _parse(path)
    out = {}
    env = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            try:
                env[node.targets[0].id] = _const(node.value, env)
            except (ValueError, TypeError):
                pass                                  # not a literal: irrelevant to the flags
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            name = _const(node.args[0])
            spec = {"type": None, "default": None, "required": False, "action": None}
            for kw in node.keywords:
                if kw.arg in spec:
                    spec[kw.arg] = _const(kw.value, env)
            out[name] = spec
    return out


def _hparams_of_reference():
    tree = _parse(os.path.join(REF, "hparams.py"))
    values = None
    derived = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "HParams":
            values = {kw.arg: _const(kw.value) for kw in node.keywords}
    assert values, "tf.contrib.training.HParams(...) call not found in the reference's hparams.py"
    # hparams.py:190-192: the `else` branch of `if hparams.use_lws` (use_lws is False by default)
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and isinstance(node.test, ast.Attribute) and node.test.attr == "use_lws":
            for st in node.orelse:
                if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Attribute):
                    derived[st.targets[0].attr] = st.value
    return values, derived


def _eval_derived(expr, hp):
    """hparams.py:190-192 right-hand sides: arithmetic over hparams.<name> and int(...)"""
    if isinstance(expr, ast.Call) and isinstance(expr.func, ast.Name) and expr.func.id == "int":
        return int(_eval_derived(expr.args[0], hp))
    if isinstance(expr, ast.BinOp):
        a, b = _eval_derived(expr.left, hp), _eval_derived(expr.right, hp)
        return {ast.Add: a + b, ast.Sub: a - b, ast.Mult: a * b, ast.Div: a / b}[type(expr.op)]
    if isinstance(expr, ast.Attribute):
        return hp[expr.attr]
    return _const(expr)


def test_hparams_names_and_defaults_equal_the_reference():
    """hparams.py:6-176: every name the reference defines exists here with the same default (value AND Python type), and nothing else
    is defined; hparams.py:190-192: the derived values follow the reference's expressions."""
    sys.path.insert(0, ROOT)
    import twvk_amd
    ref, derived = _hparams_of_reference()
    mine = twvk_amd.default_hparams().values()
    assert len(ref) >= 90
    missing = sorted(set(ref) - set(mine))
    extra = sorted(set(mine) - set(ref) - set(derived))
    assert not missing, "hparams the reference defines and this repo lacks: %s" % missing
    assert not extra, "hparams this repo defines and the reference does not: %s" % extra
    wrong = {k: (ref[k], mine[k]) for k in ref if ref[k] != mine[k] or type(ref[k]) is not type(mine[k])}
    assert not wrong, "defaults that differ (reference, repo): %s" % wrong
    assert sorted(derived) == ["frame_length_ms", "frame_shift_ms", "num_freq"]
    for name, expr in derived.items():
        want = _eval_derived(expr, ref)
        assert mine[name] == want and type(mine[name]) is type(want), (name, mine[name], want)


# flags this repo adds (documented extensions: seeds for the otherwise unseeded samplers, plumbing without a dataset / checkpoint,
# token ids in place of the out-of-scope text frontend)
EXTENSIONS = {
    "generate.py": {"--seed", "--random_init"},
    "synthesizer.py": {"--tokens", "--seed"},
    "train_vocoder.py": {"--num_steps", "--synthetic"},
}
# reference flag -> field -> (reference value, this repo's value): deliberate, explained differences
DEVIATIONS = {
    # the reference default is a Windows path list ('.\\data\\moon,.\\data\\son', train_vocoder.py:40); same directories, POSIX separators
    ("train_vocoder.py", "--data_dir", "default"): (".\\data\\moon,.\\data\\son", "./data/moon,./data/son"),
    # --text is required in the reference (synthesizer.py:374); here exactly one of --text / --tokens is required (checked after
    # parsing: the Korean text frontend is SURVEY section 2 row 17, out of scope, so token ids are the alternative input)
    ("synthesizer.py", "--text", "required"): (True, False),
}


@pytest.mark.parametrize("fname,n_ref", [("generate.py", 9), ("synthesizer.py", 8), ("train_vocoder.py", 5)])
def test_cli_flags_equal_the_reference(fname, n_ref):
    ref = _argparse_surface(os.path.join(REF, fname))
    mine = _argparse_surface(os.path.join(PKG, fname))
    assert len(ref) == n_ref, sorted(ref)
    assert not (set(ref) - set(mine)), "flags of the reference's %s missing here: %s" % (fname, sorted(set(ref) - set(mine)))
    assert set(mine) - set(ref) == EXTENSIONS[fname], (sorted(set(mine) - set(ref)), sorted(EXTENSIONS[fname]))
    for flag, spec in ref.items():
        for field in ("type", "default", "required", "action"):
            want, got = spec[field], mine[flag][field]
            dev = DEVIATIONS.get((fname, flag, field))
            if dev is not None:
                assert (want, got) == dev, (fname, flag, field, want, got)
                continue
            assert want == got and type(want) is type(got), "%s %s %s: reference %r, here %r" % (fname, flag, field, want, got)


def test_cli_post_parse_rules_of_the_reference():
    """generate.py:72-77: with hparams.gc_channels set, --gc_cardinality and --gc_id are mandatory (ValueError); generate.py:45-49: a
    negative temperature is refused by the argument type.  Both files must raise on the same conditions (compared as source text of the
    `raise` statements; nothing is executed)."""
    def raises_of(path):
        with open(path, encoding="utf-8") as fh:
            return " ".join(ast.unparse(r) for r in ast.walk(ast.parse(fh.read())) if isinstance(r, ast.Raise))
    for path in (os.path.join(REF, "generate.py"), os.path.join(PKG, "generate.py")):
        msgs = raises_of(path)
        assert "gc_cardinality" in msgs and "gc_id" in msgs and "ArgumentTypeError" in msgs, path
