#!/usr/bin/env python3
"""Statement-level similarity between this package's Python files and the reference's same-named files.

Both files are parsed, docstrings dropped, and every statement (at any nesting depth, compound statements by their
header line) is re-printed with ``ast.unparse`` so that layout, comments and line width do not matter.  The score of a
file is the share of ITS statements that also occur in the reference file (multiset intersection).  The drop-in API
dictates names, keyword arguments and result keys; it does not dictate bodies - files this build wrote itself must stay
below 30 %.  Runs only where /root/reference exists (the build container).

    python tools/ast_similarity.py            # table
"""
import ast
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/src/audio_metrics"
PAIRS = {
    "audio-metrics_amd/audio_metrics.py": "audio_metrics.py",
    "audio-metrics_amd/embed.py": "embed.py",
    "audio-metrics_amd/data.py": "data.py",
    "audio-metrics_amd/projection.py": "projection.py",
    "audio-metrics_amd/mix_functions.py": "mix_functions.py",
    "audio-metrics_amd/metrics/kd.py": "metrics/kd.py",
    "audio-metrics_amd/metrics/fad.py": "metrics/fad.py",
    "audio-metrics_amd/metrics/prdc.py": "metrics/prdc.py",
    "audio-metrics_amd/metrics/apa.py": "metrics/apa.py",
}


def _strip_docstrings(tree):
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Module)):
            body = node.body
            if body and isinstance(body[0], ast.Expr) and isinstance(getattr(body[0], "value", None), ast.Constant) \
                    and isinstance(body[0].value.value, str):
                node.body = body[1:] or [ast.Pass()]
    return tree


def statements(path):
    with open(path) as f:
        tree = _strip_docstrings(ast.parse(f.read()))
    out = Counter()
    for node in ast.walk(tree):
        if not isinstance(node, ast.stmt) or isinstance(node, ast.Pass):
            continue
        text = ast.unparse(node)
        if hasattr(node, "body") and not isinstance(node, ast.Expr):
            text = text.split("\n", 1)[0]                 # compound statement: its header only (the body is walked too)
        out[text] += 1
    return out


def similarity(ours, theirs):
    a, b = statements(ours), statements(theirs)
    shared = sum((a & b).values())
    return shared, sum(a.values()), sum(b.values())


def table():
    rows = []
    for ours, theirs in PAIRS.items():
        ref = os.path.join(REFERENCE, theirs)
        mine = os.path.join(ROOT, ours)
        if os.path.exists(ref) and os.path.exists(mine):
            shared, n_ours, n_ref = similarity(mine, ref)
            rows.append((ours, shared, n_ours, n_ref, shared / max(n_ours, 1)))
    return rows


if __name__ == "__main__":
    if not os.path.isdir(REFERENCE):
        sys.exit("the reference checkout is not available here")
    for ours, shared, n_ours, n_ref, frac in table():
        print(f"{ours:45s} {shared:4d} / {n_ours:4d} statements shared with the reference's {n_ref:4d}  = {100 * frac:5.1f} %")
