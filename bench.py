#!/usr/bin/env python3
"""Benchmark of the rerank hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): reranked query x candidate pairs / second at K=100, seq_len=512.
Workload (BASELINE.json configs[2], "c3"): FLMR multimodal query (32 prefix + 49 ViT-patch tokens + 512
text tokens), monoPreFLMR-B shaped full-context cross-encoder: 12-layer bert-base text encoder ->
768->128 -> mask -> L2 norm -> 128->768 -> 1-layer cross encoder over 593 tokens -> CLS heads ->
pointwise sigmoid/BCE head + top-K order.  Synthetic random tokens / random ViT features, seeded
random-init weights (no datasets or checkpoints exist offline).

A step = one pass of the hot path over one batch of `--queries-per-gpu x n_gpus` queries x 100 candidates,
inputs resident in HBM.  Multi-GPU: the pair list is split into contiguous per-rank slices (weak scaling:
per-GPU pairs fixed), logits are exchanged with one RCCL all_gather_into_tensor, the head runs on every
rank.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md "Chip-level parameters")


def flops_per_pair(a, S, vision):
    """SURVEY.md §8d algorithmic FLOPs per pair (padded positions count; elementwise excluded)."""
    def fl(T, H, I):
        return 8.0 * T * H * H + 4.0 * T * T * H + 4.0 * T * H * I
    H, I, D = a["hidden"], a["intermediate"], a["li_dim"]
    P = (a["prefix_len"] + a["n_patches"]) if vision else 0
    f = a["layers"] * fl(S, H, I) + 2.0 * S * H * D + 2.0 * (S + P) * D * a["ce_hidden"] \
        + a["ce_layers"] * fl(S + P, a["ce_hidden"], a["ce_intermediate"])
    if vision:
        Vh, npat, ca = a["vision_hidden"], a["n_patches"], a["cross_attn_len"]
        mid = D * a["prefix_len"] // 2
        f += 2.0 * (Vh * mid + mid * D * a["prefix_len"]) + 2.0 * npat * Vh * H + 2.0 * npat * H * D
        f += a["map_layers"] * (8.0 * npat * H * H + 4.0 * npat * npat * H + 4.0 * npat * H * H
                                + 4.0 * ca * H * H + 4.0 * npat * ca * H + 4.0 * npat * H * I)
    return f


def pmc_traffic_gb(kernel_class):
    """HBM-side bytes per launch of a kernel class from the newest committed PMC pass (profiles/*_hbm_traffic.json,
    produced by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this bench)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))["per_kernel_class"][kernel_class]
        return d["hbm_bytes_per_launch"] / 1e9
    except Exception:
        return None


def pmc_mfma_busy():
    """Fraction of SIMD cycles with the MFMA pipe busy over the step's big GEMM launches, from the newest committed SQ
    counter pass (profiles/*_sq_counters.json, tools/sq_counters.py: SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x launch
    duration x the shader clock measured under this load), duration-weighted; and that clock in GHz."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        num = den = 0.0
        for k, v in d["per_kernel"].items():
            if k.startswith("gemm_kernel") and "mfma_busy" in v:
                w = v["launches"] * v["avg_duration_ms"]
                num += w * v["mfma_busy"]
                den += w
        return (num / den if den else None), d.get("shader_clock_ghz_under_load")
    except Exception:
        return None, None


def host_cores():
    """CPU threads this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(arch, sd, K, S, vision, target_pairs):
    """The oracle (fp32 torch restatement of the reference forward) timed on this box's host cores, on a
    bounded sample of the same workload.  Reported beside the GPU number; never the thing shipped."""
    from oracle import rerank_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = O.OracleConfig(**{k: arch[k] for k in (
        "vocab_size", "hidden", "layers", "heads", "intermediate", "max_pos", "type_vocab", "ln_eps", "li_dim",
        "ce_hidden", "ce_layers", "ce_heads", "ce_intermediate", "ce_max_pos", "vision_hidden", "prefix_len",
        "n_patches", "map_layers", "cross_attn_len")})
    cfg.loss_fn = arch["loss_fn"]
    n = max(2, target_pairs)
    ids, am, tt = O.make_pair_batch(cfg, 1, n, S, seed=2022, regime="full")
    img = O.make_image_feats(cfg, 1) if vision else (None, None)
    with torch.no_grad():
        O.full_context_forward(cfg, sd, ids[:2], am[:2], tt[:2], 1, 2, img[0], img[1])      # warm the allocator
        t0 = time.perf_counter()
        O.full_context_forward(cfg, sd, ids, am, tt, 1, n, img[0], img[1])
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} pairs of the same workload (K={n}, S={S}, vision={vision}), fp32 torch oracle, "
                      f"{cores} threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries-per-gpu", type=int, default=8)
    ap.add_argument("--K", type=int, default=100)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--text-only", action="store_true")
    ap.add_argument("--encoder", default="bert-base", choices=["bert-base", "bert-large"],
                    help="bert-large = BASELINE configs[4] shape (24 layers, hidden 1024, 16 heads, FFN 4096; cross encoder of "
                         "the same width, text-only); the headline metric is quoted on bert-base")
    ap.add_argument("--regime", default="full", choices=["full", "realistic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=64)
    ap.add_argument("--no-profile", action="store_true", help="do not record per-launch HIP events")
    ap.add_argument("--compute-dtype", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit MFMA operand type of the timed run (north_star: bf16)")
    ap.add_argument("--no-alt-dtype", action="store_true", help="skip the extra fp16-operand timing (N=1 only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ    # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=dev)

    import rmr_amd
    from rmr_amd.sharding import sharded_forward
    from rmr_amd.synthetic import image_features, pair_batch

    large = args.encoder == "bert-large"
    vision = not args.text_only and not large
    shape = dict(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16, ce_intermediate=4096) if large else {}
    arch = rmr_amd.make_arch(dict(cross_encoder_num_hidden_layers=1, cross_encoder_max_position_embeddings=750,
                                  loss_fn="BCE", pos_weight=None), has_vision=int(vision),
                             compute_dtype=args.compute_dtype, **shape)
    sd = rmr_amd.synthetic_state_dict(arch, seed=0, hf_init=True)
    eng = rmr_amd.RerankEngine(arch, dev)
    eng.load_state_dict(sd)

    K, S = args.K, args.seq_len
    Bq = args.queries_per_gpu * world                       # global queries per step (weak scaling)
    N = Bq * K
    ids, am, tt = pair_batch(arch["vocab_size"], Bq, K, S, seed=2022, regime=args.regime)
    ids, am, tt = ids.to(dev), am.to(dev), tt.to(dev)
    cls = pat = None
    if vision:
        cls, pat = image_features(Bq, arch["n_patches"], arch["vision_hidden"])
        cls, pat = cls.to(dev), pat.to(dev)

    def step():
        if distributed:      # also with one rank: the same slice -> all-gather -> head path the N-GPU runs take
            return sharded_forward(eng, ids, am, tt, Bq, K, cls, pat, None, want_scores=True)
        return eng.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    eng.set_profiling(not args.no_profile)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    prof = eng.get_profile(reset=True) if not args.no_profile else None
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out["logits"]).all()

    if rank == 0:
        pairs_per_s = N * args.steps / dt
        fpp = flops_per_pair(arch, S, vision)
        res = {
            "metric": "reranked query x candidate pairs/sec at K=100, seq_len=512",
            "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.compute_dtype, "data": "synthetic",
            "config": {"workload": ("c5-shape: bert-large cross-encoder rerank (24 layers, hidden 1024, text-only, 16-bit MFMA), "
                                    f"Lc=1, K={K}, seq_len={S}, vision_tokens=0") if large else
                       ("c3: FLMR multimodal query cross-encoder rerank" if vision
                        else "c3-text: text-only cross-encoder rerank")
                       + f" (monoPreFLMR-B shape, Lc=1), K={K}, seq_len={S}, vision_tokens={81 if vision else 0}",
                       "queries_per_step": Bq, "pairs_per_step": N, "token_regime": args.regime,
                       "parallelism": f"pairs sharded over {world} GPU(s), 1 RCCL all-gather of logits/step",
                       "weights": "seeded random init (HF init), fp32 master -> bf16 MFMA operands"},
            "gflop_per_pair": fpp / 1e9,
            "whole_path_tflops_per_gpu": pairs_per_s * fpp / 1e12 / world,
            "whole_path_frac_of_bf16_peak": pairs_per_s * fpp / 1e12 / world / PEAK_BF16_TFLOPS,
        }
        if prof is not None:
            g = prof["gemm"]
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
            res["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel_hp (16-bit MFMA GEMM, persistent half-tile LDS ring, fused epilogues; all GEMM launches of the step)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_BF16_TFLOPS,
                               # the committed PMC passes were taken on the headline workload only
                               "traffic": None if large else pmc_traffic_gb("gemm"),
                               "mfma_busy_pmc": None if large else pmc_mfma_busy()[0],
                               "shader_clock_ghz_pmc": None if large else pmc_mfma_busy()[1],
                               "traffic_unit": "GB per launch beyond L2 (FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)",
                               "algorithmic_gb_per_launch": g["bytes"] / max(1, g["launches"]) / 1e9,
                               "launches": g["launches"], "avg_launch_ms": g["ms"] / max(1, g["launches"]),
                               "avg_launch_gflop": g["flops"] / max(1, g["launches"]) / 1e9,
                               "note": "per-launch HIP events on the work stream inside the timed region, rank 0"}
            tot = sum(v["ms"] for v in prof.values())
            res["kernel_time_share"] = {k: (v["ms"] / tot if tot else 0.0) for k, v in prof.items()}
            a = prof["attention"]
            res["attention_tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
        if world == 1 and not args.no_alt_dtype and args.compute_dtype == "bf16":
            # same kernels with fp16 MFMA operands (the mode that meets 1e-3 against the fp32 reference logits)
            del eng
            torch.cuda.empty_cache()
            eng2 = rmr_amd.RerankEngine(dict(arch, compute_dtype="fp16"), dev)
            eng2.load_state_dict(sd)
            for _ in range(max(1, args.warmup)):
                eng2.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng2.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)
            torch.cuda.synchronize(dev)
            res["fp16_operand_mode"] = {"value": N * args.steps / (time.perf_counter() - t1), "unit": "pairs/s",
                                        "note": "compute_dtype=fp16: logits within 1e-3 of the fp32 HF goldens "
                                                "(tests/test_gpu_forward.py)"}
            del eng2
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(arch, sd, K, S, vision, args.cpu_pairs)
            res["gpu_over_cpu"] = pairs_per_s / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
