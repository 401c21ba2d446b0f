"""CLIP ViT-B/32 vision tower (rr_encode_image) throughput: images/s at several batch sizes (synthetic pixels, HF-init
weights).  The tower runs once per query, i.e. 1/K of the pair count — DESIGN.md quotes these numbers."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import rmr_amd  # noqa: E402

arch = rmr_amd.make_arch(None, vit_layers=12, layers=1, ce_layers=1)
sd = rmr_amd.synthetic_state_dict(arch, seed=0)
eng = rmr_amd.RerankEngine(arch)
eng.load_state_dict(sd)
res = []
for B in (1, 8, 64, 256):
    px = torch.randn(B, 3, 224, 224, device="cuda")
    for _ in range(3):
        eng.encode_image(px)
    torch.cuda.synchronize()
    it = 20
    t0 = time.perf_counter()
    for _ in range(it):
        eng.encode_image(px)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    T, H, I, L = 50, 768, 3072, 12
    flops = B * (2 * 49 * 3072 * H + L * (8 * T * H * H + 4 * T * T * H + 4 * T * H * I))
    res.append(dict(images=B, ms=round(dt * 1e3, 3), images_per_s=round(B / dt, 1), tflops=round(flops / dt / 1e12, 1)))
print(json.dumps(res))
