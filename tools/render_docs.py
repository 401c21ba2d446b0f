"""Render DESIGN.md / README.md from docs/DESIGN.in.md / docs/README.in.md: every measured figure of a document is an expression

    {{profiles/<file>.json "<key>" "<key>" ...|<python format spec>}}        e.g.  {{profiles/r05_h_c3_bench.json.log "value"|.0f}}

that is replaced by the value found in the committed record, formatted, FOLLOWED by its citation in the form
tests/test_docs_cpu.py checks (`<number> [`profiles/<file>.json` "<key>" ...]`).  `{{=...}}` renders the number alone (dense
tables: the row or the caption carries the citation).  A `.json.log` file is a bench log whose last line starting with `{` is the
record.  Nothing is scaled: the document quotes the records in the records' own units.

    python tools/render_docs.py            # writes DESIGN.md and README.md; fails on a missing file / key
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPR = re.compile(r'\{\{(=?)(profiles/[\w.\-]+\.json(?:\.log)?)((?:\s+"[^"]+")+)\|([^}]+)\}\}')
_cache = {}


def load(path):
    if path not in _cache:
        full = os.path.join(ROOT, path)
        if path.endswith(".json"):
            _cache[path] = json.load(open(full))
        else:
            _cache[path] = json.loads([l for l in open(full) if l.startswith("{")][-1])
    return _cache[path]


def render(text):
    def sub(m):
        bare, path, keys, fmt = m.group(1), m.group(2), re.findall(r'"([^"]+)"', m.group(3)), m.group(4)
        v = load(path)
        for k in keys:
            v = v[int(k)] if isinstance(v, list) else v[k]
        s = format(v, fmt)
        return s if bare else s + " [`" + path + "` " + " ".join('"' + k + '"' for k in keys) + "]"
    return EXPR.sub(sub, text)


if __name__ == "__main__":
    for name in ("DESIGN", "README"):                      # docs/<name>.in.md -> <name>.md
        src = os.path.join(ROOT, "docs", name + ".in.md")
        out = render(open(src, encoding="utf-8").read())
        left = re.findall(r"\{\{[^}]*\}\}", out)
        if left:
            sys.exit(f"{name}: unrendered expressions: {left[:5]}")
        open(os.path.join(ROOT, name + ".md"), "w", encoding="utf-8").write(out)
        print(name + ".md:", out.count("\n") + 1, "lines,", len(EXPR.findall(open(src, encoding='utf-8').read())), "figures from records")
