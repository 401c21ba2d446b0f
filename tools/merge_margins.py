"""Fold the parity margins a `-m gpu` run recorded (tests/helpers.record_margin -> gpurun_out/parity_margins.json) into the
committed record profiles/rNN_parity_margins.json WITHOUT ever losing a key (a one-test re-run was copied over the
35-entry file of the full run).

    python tools/merge_margins.py r04            # gpurun_out/parity_margins.json -> profiles/r04_parity_margins.json
    python tools/merge_margins.py r04 --check profiles/r03_full_parity_margins.json   # also require a superset of that file's keys

Rules: keys only ever get added or refreshed; a refresh from a run that covers FEWER keys than the committed file is allowed (the
new values are newer) but is reported; `--fresh` replaces the file and then REFUSES unless the new run covers every key of the
old one.  Every entry carries the `run` tag record_margin stamped it with (one tag per pytest process), so a reader can see
whether the file is from one run.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("round")
ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
ap.add_argument("--check", default=None, help="another margins file whose keys must all be present afterwards")
ap.add_argument("--fresh", action="store_true", help="replace instead of merge; refused when keys would disappear")
a = ap.parse_args()
dst = os.path.join(ROOT, "profiles", f"{a.round}_parity_margins.json")
new = json.load(open(a.src))
old = json.load(open(dst)) if os.path.exists(dst) else {}
if a.fresh:
    lost = sorted(set(old) - set(new))
    if lost:
        sys.exit(f"refused: --fresh would drop {len(lost)} keys, e.g. {lost[:5]}")
    out = dict(new)
else:
    out = dict(old)
    out.update(new)
missing = []
if a.check:
    want = json.load(open(a.check))
    missing = sorted(set(want) - set(out))
    if missing:
        sys.exit(f"refused: {len(missing)} keys of {a.check} are missing, e.g. {missing[:5]}")
with open(dst, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
runs = sorted({str(v.get("run", "?")) for v in out.values() if isinstance(v, dict)})
print(f"{dst}: {len(out)} keys ({len(set(new) - set(old))} new, {len(set(new) & set(old))} refreshed, {len(set(old) - set(new))} kept from "
      f"the committed file); run tags: {runs}")
