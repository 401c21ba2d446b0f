"""Build a VARIANT of the library next to the product one, for same-box A/B runs: one source recompiled with extra defines, the other
objects taken from reranking-multimodal-retrievers_amd/build/ (run build.py first).  Output: tools/bin/librerank_<name>.so (git-ignored,
shipped to the GPU box by gpurun).

    python tools/build_variant.py <name> <source.hip> -DRR_EPI_DIAG=1 [...]
"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("rr_build", os.path.join(ROOT, "reranking-multimodal-retrievers_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build_library()
objdir = os.path.join(b.HERE, "build")
os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
vobj = os.path.join(ROOT, "tools", "bin", f"{os.path.splitext(src)[0]}_{name}.o")
subprocess.run([b.HIPCC, *b.FLAGS, *b.EXTRA.get(src, []), *extra, "-c", os.path.join(b.CSRC, src), "-o", vobj], check=True)
objs = [vobj if s == src else os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in b.SOURCES]
out = os.path.join(ROOT, "tools", "bin", f"librerank_{name}.so")
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out, *objs], check=True)
print(out)
