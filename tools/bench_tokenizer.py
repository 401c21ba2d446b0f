"""Pair-input assembly throughput (CPU only): the native rr_tok_prepare_pairs against the Python mirror driving the
installed HF BertTokenizer, on synthetic text of the bench's shape (K = 100 candidates of ~300 words per query)."""
import os
import random
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402
from rmr_amd.pair_inputs import NativePairTokenizer, prepare_full_context_inputs  # noqa: E402

rng = random.Random(0)
syll = ["ka", "to", "mi", "ra", "ne", "so", "lu", "vi", "en", "or", "th", "st", "ing", "ed", "er", "al", "pre", "con"]
words = sorted({"".join(rng.choice(syll) for _ in range(rng.randint(1, 3))) for _ in range(4000)})
vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words + \
        ["##" + s for s in syll] + list(".,?!'-")
nq, K = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 100
text = lambda n: " ".join(rng.choice(words) + rng.choice(["", "", "", ",", "."]) for _ in range(n))
q = [text(12) + "?" for _ in range(nq)]
c = [text(rng.randint(150, 400)) for _ in range(nq * K)]
d = tempfile.mkdtemp()
open(os.path.join(d, "vocab.txt"), "w").write("\n".join(vocab) + "\n")
from transformers import BertTokenizer  # noqa: E402
hf = BertTokenizer(os.path.join(d, "vocab.txt"), do_lower_case=True)
for threads in (1, 4, 8, 16):
    nt = NativePairTokenizer(vocab, n_threads=threads)
    nt.prepare_full_context_inputs(q, c, 32, 476, 512, K)
    t0 = time.perf_counter()
    for _ in range(3):
        out = nt.prepare_full_context_inputs(q, c, 32, 476, 512, K)
    dt = (time.perf_counter() - t0) / 3
    print(f"native, {threads:2d} threads: {nq * K / dt:9.0f} pairs/s ({dt * 1e3:.1f} ms for {nq * K} pairs)", flush=True)
t0 = time.perf_counter()
ref = prepare_full_context_inputs(q, c, hf, 32, 476, 512, K)
dt = time.perf_counter() - t0
print(f"python mirror on HF BertTokenizer ({type(hf).__name__}, Rust-backed): {nq * K / dt:9.0f} pairs/s ({dt * 1e3:.1f} ms)")
print("identical ids:", all((ref[k] == out[k]).all().item() for k in ref))
