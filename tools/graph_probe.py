"""Single-query latency (Bq = 1, K = 100, S = 512, vision): eager launches vs one HIP-graph replay of the same calls."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rmr_amd  # noqa: E402
from rmr_amd import synthetic  # noqa: E402

arch = rmr_amd.make_arch(dict(cross_encoder_num_hidden_layers=1, cross_encoder_max_position_embeddings=750, loss_fn="BCE"))
eng = rmr_amd.RerankEngine(arch)
eng.load_state_dict(rmr_amd.synthetic_state_dict(arch, seed=0))
for Bq in (1, 2):
    K, S = 100, 512
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 30000, (Bq * K, S), generator=g).cuda()
    am = torch.ones_like(ids)
    tt = torch.zeros_like(ids)
    cls = torch.randn(Bq, 768, generator=g).cuda()
    pat = torch.randn(Bq, 49, 768, generator=g).cuda()

    def run():
        return eng.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_order=True, want_scores=True)

    for _ in range(3):
        r = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r = run()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    ref = r["logits"].clone()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            out = run()
    torch.cuda.synchronize()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 20
    print(f"Bq={Bq}: eager {eager*1e3:.2f} ms, graph replay {graph*1e3:.2f} ms, logits equal {torch.equal(out['logits'], ref)}", flush=True)
