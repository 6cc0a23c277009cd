#!/usr/bin/env python3
"""Record the call signatures of the reference's public callables on the hot path as DATA.

    python -B tests/golden/make_signatures.py        # only works where /root/reference exists

The reference's modules are parsed as text with `ast` (nothing is imported or executed); for every module-level
function, and every method of a module-level class, the parameter list is stored as
[name, kind, default] with kind in {"positional", "vararg", "keyword_only", "varkw"} and default = the literal's
value (null when the parameter has none, {"expr": "..."} when it is not a literal).  tests/test_dropin_signatures.py
holds the drop-in modules against this file: a caller of the reference must be able to pass the same positional
and keyword arguments.  No reference source is written, only names and default values.
"""
import ast
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
MODULES = ["speech_encode", "ecapa_annote", "vad", "anti_stick_diarize", "diarization_baseline"]


def default_value(node):
    if node is None:
        return None
    try:
        return {"value": ast.literal_eval(node)}
    except (ValueError, SyntaxError):
        return {"expr": ast.unparse(node)}


def params(fn: ast.FunctionDef):
    a = fn.args
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    out = [[p.arg, "positional", default_value(d)] for p, d in zip(pos, defaults)]
    if a.vararg:
        out.append([a.vararg.arg, "vararg", None])
    out += [[p.arg, "keyword_only", default_value(d)] for p, d in zip(a.kwonlyargs, a.kw_defaults)]
    if a.kwarg:
        out.append([a.kwarg.arg, "varkw", None])
    return out


def main():
    rec = {}
    for mod in MODULES:
        tree = ast.parse(open(os.path.join(REF, mod + ".py"), encoding="utf-8").read())
        fns = {}
        for node in tree.body:
            if isinstance(node, ast.FunctionDef):
                fns[node.name] = params(node)
            elif isinstance(node, ast.ClassDef):
                for sub in node.body:
                    if isinstance(sub, ast.FunctionDef) and (not sub.name.startswith("_") or sub.name in ("__init__", "__call__")):
                        fns[f"{node.name}.{sub.name}"] = params(sub)
        rec[mod] = fns
    with open(os.path.join(HERE, "signatures.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print({m: len(v) for m, v in rec.items()})


if __name__ == "__main__":
    main()
