"""Build librerank_mi355.so in-tree with hipcc for gfx950 (no torch involved).

`python reranking-multimodal-retrievers_amd/build.py [--force]` or `build_library()` from Python.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librerank_mi355.so")
SOURCES = ["rr_api.hip", "gemm_bf16.hip", "gemm_fp8.hip", "attention_bf16.hip", "elementwise.hip", "head.hip", "pair_tokenizer.cpp"]
HEADERS = [os.path.join(CSRC, "rr_common.h"), os.path.join(os.path.dirname(HERE), "include", "rerank_mi355.h"),
           os.path.join(CSRC, "unicode_tables.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: no compiler-formed v_pk_*_f32.  Root cause of the corruption it was added for (round 2, ISA and
# measurements in profiles/r02_slp_hazard_isa.txt, tools/slp_hazard_probe.py): with SLP on, the LayerNorm-residual
# epilogue of the 128x128 GEMM compiles to `s_waitcnt vmcnt(2)` DIRECTLY followed by `v_pk_add_f32` on the registers the
# load just returned, and lanes 48-63 of the LOW register of the pair are intermittently read stale (16 rows x 1 column
# of a few tiles per launch).  One instruction between the wait and the first packed consumer cures it: the site
# (gemm_bf16.hip ln_apply) now carries an explicit `s_nop 1` behind the loads, so it is safe even with SLP on (verified:
# 0 wrong elements in 4 launches against 16-48 in each of 4).  The flag stays as the second line of defence: other
# epilogues add freshly loaded fp32 rows too, and hipcc's hazard recogniser does not know this pair.  Cost 0.7 % of the
# bench step.  Hand-written 2-wide vector code (attention row sum) only touches VALU-produced values.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
         *os.environ.get("RR_HIPCC_EXTRA", "").split()]          # RR_HIPCC_EXTRA: A/B experiments only
# per-file extras: the attention softmax has no NaNs by construction; without IEEE-mode canonicalisation its 32-way row
# max is 16 v_max3_f32 instead of 54 instructions
EXTRA = {"attention_bf16.hip": ["-fno-honor-nans", "-mno-amdgpu-ieee"]}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        if force or _stale(obj, [src] + HEADERS):
            if s.endswith(".cpp"):       # host-only C++ (no device code)
                jobs.append([HIPCC, "-O2", "-fPIC", "-std=c++17", "-Wall", "-pthread", "-c", src, "-o", obj])
            else:
                jobs.append([HIPCC, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
