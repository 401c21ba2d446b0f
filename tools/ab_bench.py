"""A/B a process-wide tuning switch on the whole c3 forward, interleaved rounds in ONE process (rule 24).

    python tools/ab_bench.py ln_lite 0 1
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd
from rmr_amd import _lib
from rmr_amd.synthetic import image_features, pair_batch

key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
lib = _lib.load()
arch = rmr_amd.make_arch(dict(cross_encoder_num_hidden_layers=1, cross_encoder_max_position_embeddings=750, loss_fn="BCE"), compute_dtype=os.environ.get("RR_DTYPE", "fp16"))
eng = rmr_amd.RerankEngine(arch)
eng.load_state_dict(rmr_amd.synthetic_state_dict(arch, 0, True))
Bq, K, S = 8, 100, 512
ids, am, tt = [t.cuda() for t in pair_batch(arch["vocab_size"], Bq, K, S, regime=os.environ.get("RR_REGIME", "full"))]   # RR_REGIME=realistic: doc length ~ U[64, S], tail padded
cls, pat = [t.cuda() for t in image_features(Bq, 49, 768)]
res = {v: [] for v in vals}
ref = None
for r in range(6):
    for v in vals:
        if key == "gemm_variant":
            assert lib.rr_set_gemm_variant(v) == 0
        elif key == "gemm_stagger":          # 50..55: tile-order group of 2..64 row panels in the persistent GEMM, 56: row-major, 59: without the serpentine K walk, 0: heuristic
            assert lib.rr_set_gemm_stagger(v) == 0
        else:
            assert lib.rr_set_tuning(key.encode(), v) == 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            out = eng.forward_ids(ids, am, tt, Bq, K, cls, pat, want_order=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        if r:
            res[v].append(dt)
        if ref is None:
            ref = out["logits"].clone()
        else:
            print(f"  {key}={v}: max |dlogit| vs first setting {(out['logits'] - ref).abs().max().item():.2e}") if r == 0 else None
for v, t in res.items():
    print(f"{key}={v}: min {min(t)*1e3:.2f} ms  med {sorted(t)[len(t)//2]*1e3:.2f} ms  -> {Bq*K/min(t):.0f} pairs/s")
