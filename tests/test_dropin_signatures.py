"""Every callable a caller of the reference can reach through the flat-name shims in dropin/ accepts the reference's
positional and keyword arguments: same parameter names, order and defaults (tests/golden/signatures.json, recorded by
tests/golden/make_signatures.py from the reference's sources as names + default values); what this build adds must be
keyword-only or trail the reference's parameters with a default."""
import importlib
import inspect
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIGS = json.load(open(os.path.join(ROOT, "tests", "golden", "signatures.json")))

# reference callables with no counterpart, each for a stated reason
ABSENT = {
    ("diarization_baseline", "using_speaker_diarization_cnceleb"): "loads pyannote/speaker-diarization-3.1 from the hub (out of scope, SURVEY 8c)",
}
KIND = {inspect.Parameter.POSITIONAL_ONLY: "positional", inspect.Parameter.POSITIONAL_OR_KEYWORD: "positional",
        inspect.Parameter.VAR_POSITIONAL: "vararg", inspect.Parameter.KEYWORD_ONLY: "keyword_only", inspect.Parameter.VAR_KEYWORD: "varkw"}


def _dropin(mod):
    path = os.path.join(ROOT, "dropin")
    sys.path.insert(0, path)
    try:
        sys.modules.pop(mod, None)
        return importlib.import_module(mod)
    finally:
        sys.path.remove(path)
        sys.modules.pop(mod, None)


def _resolve(module, qual):
    obj = module
    for part in qual.split("."):
        obj = getattr(obj, part)
    return getattr(obj, "__wrapped__", obj)          # lru_cache wrappers


CASES = [(m, q) for m, fns in sorted(SIGS.items()) for q in sorted(fns)]


@pytest.mark.parametrize("mod,qual", CASES)
def test_reference_call_forms_are_accepted(mod, qual):
    if (mod, qual) in ABSENT:
        pytest.skip(ABSENT[(mod, qual)])
    fn = _resolve(_dropin(mod), qual)
    ours = list(inspect.signature(fn).parameters.values())
    ref = SIGS[mod][qual]
    assert len(ours) >= len(ref), f"{mod}.{qual}: fewer parameters than the reference"
    for i, (name, kind, default) in enumerate(ref):
        p = ours[i]
        assert p.name == name and KIND[p.kind] == kind, f"{mod}.{qual}: parameter {i} is {p.name} ({KIND[p.kind]}), reference has {name} ({kind})"
        if default is None:
            assert p.default is inspect.Parameter.empty, f"{mod}.{qual}: {name} has a default, the reference's is required"
        elif "value" in default:
            assert p.default is not inspect.Parameter.empty and p.default == default["value"] and type(p.default) is type(default["value"]), \
                f"{mod}.{qual}: default of {name} is {p.default!r}, reference {default['value']!r}"
        else:
            assert p.default is not inspect.Parameter.empty, f"{mod}.{qual}: {name} needs a default (reference: {default['expr']})"
    for p in ours[len(ref):]:                        # additions never shift or break a reference call
        assert p.kind in (inspect.Parameter.KEYWORD_ONLY, inspect.Parameter.VAR_KEYWORD) or p.default is not inspect.Parameter.empty, \
            f"{mod}.{qual}: extra parameter {p.name} is required"


def test_fixture_holds_names_and_values_only():
    text = open(os.path.join(ROOT, "tests", "golden", "signatures.json")).read()
    assert "def " not in text and "import " not in text and "return" not in text
