"""40 forwards of each main configuration (c3 / configs[4] fp8, padded and packed): the logits must hash to ONE value (run-to-run
determinism: no atomics, no order-dependent reductions on the path), stay finite, and leave the fp16 range guard quiet.

    python tools/soak_determinism.py
"""
import sys, hashlib
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rmr_amd
from rmr_amd.synthetic import image_features, pair_batch
def run(wl, packed):
    shape = dict(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16, ce_intermediate=4096) if wl.startswith("c5") else {}
    arch = rmr_amd.make_arch(dict(cross_encoder_num_hidden_layers=1, cross_encoder_max_position_embeddings=750, loss_fn="BCE"),
                             has_vision=int(wl == "c3"), compute_dtype="fp16", **shape)
    if wl == "c5fp8": arch["fp8"] = 1
    eng = rmr_amd.RerankEngine(arch); eng.load_state_dict(rmr_amd.synthetic_state_dict(arch, 0, True))
    Bq, K, S = (8, 100, 512) if wl == "c3" else (4, 200, 512)
    ids, am, tt = [t.cuda() for t in pair_batch(arch["vocab_size"], Bq, K, S, regime="realistic" if packed else "full")]
    cls, pat = [t.cuda() for t in image_features(Bq, 49, 768)] if wl == "c3" else (None, None)
    hs = set()
    for i in range(40):
        out = eng.forward_ids_packed(ids, am, tt, Bq, K, cls, pat) if packed else eng.forward_ids(ids, am, tt, Bq, K, cls, pat)
        hs.add(hashlib.sha256(out["logits"].cpu().numpy().tobytes()).hexdigest()[:12])
    print(wl, "packed" if packed else "padded", "40 forwards, distinct logit hashes:", len(hs), "finite:", bool(torch.isfinite(out["logits"]).all()), "range flag:", eng.activation_range_exceeded(), flush=True)
for wl, p in (("c3", False), ("c3", True), ("c5fp8", False), ("c5fp8", True)):
    run(wl, p)
