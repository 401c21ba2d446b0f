"""CPU: the documents quote what the committed records hold (two rounds running a figure in DESIGN.md / a docstring
disagreed with the profiles/ file it cited).

Citation form, in DESIGN.md, README.md, INTEGRATION.md and the docstrings / comments of tests/*.py:

    <number> [`profiles/<file>.json` "<key>" "<key>" ...]          e.g.  3.2e-4 [`profiles/r05_parity_margins.json` "c3_full/fp16" "max_abs"]

(`<file>.json.log` = a bench log whose last line starting with `{` is the JSON record.)  The number in front must equal the cited value
rounded to the digits that are written: 9117 cites 9117.3, 3.2e-4 cites 3.24e-4, 0.39 cites 0.3861 — and not 0.38.
"""
import json
import os
import re
from decimal import Decimal

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITE = re.compile(r'([-+]?\d[\d.]*(?:[eE][-+]?\d+)?)\s*(?:ms|GB|TB/s|TFLOP/s|pairs/s|%|us|µs|x|×)?\s*\[`(profiles/[\w.\-]+\.json(?:\.log)?)`((?:\s+"[^"]+")+)\]')


def _documents():
    docs = [os.path.join(ROOT, n) for n in ("DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md")]
    docs += [os.path.join(ROOT, "tests", f) for f in sorted(os.listdir(os.path.join(ROOT, "tests")))
             if f.endswith(".py") and f != os.path.basename(__file__)]
    return [d for d in docs if os.path.exists(d)]


def _load(path):
    full = os.path.join(ROOT, path)
    if path.endswith(".json"):
        with open(full) as f:
            return json.load(f)
    rec = None
    with open(full) as f:
        for line in f:
            if line.startswith("{"):
                rec = line
    return json.loads(rec)


def _matches(quoted: str, value: float) -> bool:
    """`quoted` equals `value` rounded to the precision `quoted` is written with."""
    q = Decimal(quoted)
    exp = q.as_tuple().exponent                     # 3.2e-4 -> -5 ; 9117 -> 0 ; 0.39 -> -2
    step = Decimal(1).scaleb(exp)
    v = Decimal(repr(float(value)))
    return abs(v - q) <= step / 2 + Decimal(repr(abs(float(value)) * 1e-12))


def citations():
    out = []
    for doc in _documents():
        text = open(doc, encoding="utf-8").read()
        for m in CITE.finditer(text):
            keys = re.findall(r'"([^"]+)"', m.group(3))
            out.append((os.path.relpath(doc, ROOT), m.group(1), m.group(2), tuple(keys)))
    return out


def test_matcher_rounds_the_way_the_documents_write_numbers():
    assert _matches("9117", 9117.3) and not _matches("9117", 9118.2)
    assert _matches("3.2e-4", 3.24e-4) and not _matches("3.2e-4", 3.3e-4)
    assert _matches("0.39", 0.3861) and not _matches("0.38", 0.3861)
    assert _matches("87.76", 87.7627) and _matches("0.5", 0.5)


def test_every_cited_figure_equals_the_cited_record():
    cites = citations()
    assert len(cites) >= 20, f"only {len(cites)} checked citations: DESIGN.md is expected to cite its figures in the checked form"
    bad = []
    for doc, quoted, path, keys in cites:
        try:
            v = _load(path)
            for k in keys:
                v = v[int(k)] if isinstance(v, list) else v[k]
            if isinstance(v, bool) or not isinstance(v, (int, float)):
                bad.append(f"{doc}: {quoted} cites {path} {keys}, which is not a number ({v!r})")
            elif not _matches(quoted, v):
                bad.append(f"{doc}: {quoted} cites {path} {' '.join(keys)} = {v!r}")
        except (OSError, KeyError, IndexError, ValueError, TypeError) as e:
            bad.append(f"{doc}: {quoted} cites {path} {keys}: {type(e).__name__} {e}")
    assert not bad, "\n".join(bad)


def test_design_is_the_current_state_only():
    """DESIGN.md = current state (<= 400 lines); the round narratives live under docs/rounds/."""
    n = sum(1 for _ in open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8"))
    assert n <= 400, f"DESIGN.md has {n} lines"
    assert os.path.isdir(os.path.join(ROOT, "docs", "rounds"))
    text = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    for section in ("## 1.", "## 2.", "## 3.", "## 4.", "## 5.", "## 6.", "## 7."):
        assert section in text


@pytest.mark.parametrize("header", ["rerank_mi355.h", "rerank_mi355_diag.h"])
def test_headers_cite_the_reference_interface_they_replace(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    assert "/root/reference/" in text or "src/models/rerank" in text
