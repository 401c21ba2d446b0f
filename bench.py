#!/usr/bin/env python3
"""Benchmark of the rerank hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: starts its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                (the driver's form: ranks come from the environment)

Metric (BASELINE.json): reranked query x candidate pairs / second at K=100, seq_len=512.
Workloads (`--workload`, BASELINE.json `configs`):
  c3 (default, the headline; configs[2]): FLMR multimodal query (32 prefix + 49 ViT-patch tokens + 512 text tokens),
     monoPreFLMR-B shaped full-context cross-encoder: 12-layer bert-base text encoder -> 768->128 -> mask -> L2 norm ->
     128->768 -> 1-layer cross encoder over 593 tokens -> CLS heads -> pointwise sigmoid/BCE head + top-K order.
  c4 (configs[3]): the same encoder with the LISTWISE head (loss_fn negative_sampling: logits viewed [Bq, K], softmax over
     K, cross-entropy against candidate 0), 64 queries per step at 8 GPUs (8 per GPU, weak scaling).
  c5 (configs[4]): bert-large (24 layers, hidden 1024, FFN 4096) text-only cross encoder, K = 200.
  L : monoPreFLMR-L geometry (ViT-L/14 features: 1024-d, 256 patches -> P = 288 vision tokens, T = 800).
Synthetic random tokens / random ViT features, seeded random-init weights (no datasets or checkpoints exist offline).

A step = one pass of the hot path over one batch, inputs resident in HBM.  `--scaling weak` (default): the batch is
`--queries-per-gpu x n_gpus` queries x K candidates; `--scaling strong`: `--queries-per-gpu` queries in total (default 1:
one query's K pairs split over the ranks).  Multi-GPU: the pair list is split into contiguous per-rank slices, logits are
exchanged with one RCCL all_gather_into_tensor, the head runs on every rank.  Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then K steps between (barrier + device synchronize) fences, max over ranks -> `value`.  Inside
the timed region every step is also bracketed by HIP events on the work stream (-> median / p10 / p90 of the per-step
device time) and the library brackets every kernel launch with HIP events (-> per-class device time, the roofline of
the GEMM); `--no-profile` switches the per-launch events off.  At N = 1 the end-to-end window the reference times
(Reranker_base_executor.py:898-939: H2D ids -> forward -> D2H logits/order for ONE query) and the CPU baseline follow,
outside the timed region.

`python bench.py --gpus N` without a launcher environment starts `python -m torch.distributed.run --nproc-per-node N`
on itself as a CHILD process before anything touches the GPU and exits with the child's code.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this driver (set before HIP initialises)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md "Chip-level parameters")
PEAK_FP8_TFLOPS = 5000.0      # dense e4m3 on the block-scaled matrix core


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5", "L"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--queries-per-gpu", type=int, default=None,
                    help="weak: queries per GPU per step (default 8; c5: 4); strong: queries per step in total (default 1)")
    ap.add_argument("--K", type=int, default=None, help="candidates per query (default 100; c5: 200)")
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--text-only", action="store_true")
    ap.add_argument("--encoder", default=None, choices=["bert-base", "bert-large"], help="deprecated alias: bert-large = --workload c5")
    ap.add_argument("--regime", default="full", choices=["full", "realistic"])
    ap.add_argument("--fp8", action="store_true", help="c5: e4m3 operands for the QKV and FFN-up GEMMs (DESIGN.md §fp8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=100, help="pairs of the CPU-baseline sample (one query of K = 100)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-launch HIP events")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end per-query window (N = 1)")
    ap.add_argument("--compute-dtype", default="fp16", choices=["bf16", "fp16"],
                    help="16-bit MFMA operand type of the timed run.  Default fp16: the mode whose logits are within 1e-3 of the "
                         "fp32 reference forward on every golden (same MFMA rate as bf16, 3 more mantissa bits; accumulation, "
                         "residual stream, LayerNorm, softmax stay fp32).  bf16 (what the reference's autocast uses) drifts "
                         "2e-3..1e-2 from fp32 in ANY implementation, the reference's own included (tests/golden/autocast.npz)")
    ap.add_argument("--no-alt-dtype", action="store_true", help="skip the extra timing of the other operand type (N=1 only)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks time-share cuda:0 and exchange over gloo: exercises the spawn / shard / gather / head "
                         "path with the real engine on a one-GPU box; the line is marked, it is NOT a scaling measurement")
    ap.add_argument("--bucketed", action="store_true",
                    help="length-bucketed execution (RerankEngine.forward_ids_bucketed): with --regime realistic the pairs run at the "
                         "row length of their bucket instead of the padded seq_len; pairs/s still counts padded pairs (the contract)")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: capture the step into a HIP graph (the library is capturable after rr_reserve) and time replays; "
                         "profiling events are off in this mode (the line carries no live roofline)")
    ap.add_argument("--packed", action="store_true",
                    help="packed execution (RerankEngine.forward_ids_packed / rr_forward_packed): the pairs grouped by length in steps "
                         "of --granule rows, every GEMM of a layer ONE launch over the rows that exist; pairs/s still counts padded pairs")
    ap.add_argument("--granule", type=int, default=16, help="row-length step of --packed")
    ap.add_argument("--segment-cost-rows", type=int, default=0,
                    help="--packed: merge neighbouring lengths where a segment's fixed launches cost more than the rows the merge pads "
                         "(rows one segment is worth; 0 = one segment per distinct length)")
    ap.add_argument("--weights-gain", type=float, default=1.0,
                    help="std multiplier of the Linear matrices of the synthetic weights (1 = HF init: near-uniform attention; 2.5 = the "
                         "peaked-attention regime of tests/golden c3_sep).  The line then also reports how many attention workgroups "
                         "the fixed-reference schedule had to recompute online (`attention_redo`)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=INT",
                    help="process-wide A/B switch of the library (rr_set_tuning), e.g. --tuning ln_fold=0; diagnostic")
    a = ap.parse_args(argv)
    if a.encoder == "bert-large":
        a.workload = "c5"
    if a.K is None:
        a.K = 200 if a.workload == "c5" else 100
    if a.queries_per_gpu is None:
        a.queries_per_gpu = 1 if a.scaling == "strong" else (4 if a.workload == "c5" else (2 if a.workload == "L" else 8))
    return a


def spawn_ranks(args):
    """`bench.py --gpus N` outside a launcher: start the N ranks as a child `torch.distributed.run` (never re-exec a
    process that has touched the GPU; this process has not — torch is not even imported yet)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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


def archived_pmc():
    """Counter passes cannot run inside this process (rocprofv3 owns the counters), so the PMC figures printed beside the
    live numbers come from the newest COMMITTED passes of this very command (profiles/*_hbm_traffic.json from
    tools/pmc_traffic.py = separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs, FETCH_SIZE doubled per the
    gfx950 correction; profiles/*_sq_counters.json from tools/sq_counters.py).  They are labelled with their source file
    and were taken on the headline workload (c3) only."""
    import glob
    out = {"note": "archived rocprofv3 --pmc passes of `python bench.py` (c3) on an earlier MI355X box; not measured in this run"}
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")) if "_c5" not in os.path.basename(f))
    if files:
        try:
            d = json.load(open(files[-1]))["per_kernel_class"]["gemm"]
            out["traffic_gb_per_gemm_launch"] = d["hbm_bytes_per_launch"] / 1e9
            out["traffic_source"] = os.path.relpath(files[-1], ROOT)
        except Exception:
            pass
    # the vendor library's GEMM at the path's shapes, measured beside this kernel's plain-epilogue form (tools/bench_vendor_gemm.py):
    # what a library tile loop reaches at K = 768 / 3 072 on this chip — the yardstick for roofline.frac
    files_v = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_vendor_gemm_yardstick.log")))
    if files_v:
        try:
            import re
            y = {}
            for line in open(files_v[-1]):
                mo = re.match(r"(\w+)\s+M=\d+ N=\d+ K=\d+ (\w+): ours: min [\d.]+ ms\s+([\d.]+) TF .*vendor: min [\d.]+ ms\s+([\d.]+) TF", line)
                if mo and mo.group(2) == "fp16":
                    y[mo.group(1)] = {"this_kernel_plain_epilogue_tflops": float(mo.group(3)), "vendor_gemm_no_epilogue_tflops": float(mo.group(4))}
            if y:
                out["vendor_gemm_yardstick_fp16"] = y
                out["vendor_gemm_source"] = os.path.relpath(files_v[-1], ROOT)
        except Exception:
            pass
    # what the matrix core SUSTAINS on this chip under its power cap (tools/mfma_shape_probe.hip: every CU, two waves per SIMD, random
    # fp16 operands; with the GEMM main loop's 12 ds_read_b128 per 32 MFMAs beside them): the physical ceiling of an LDS-fed tile loop,
    # below the 2.5 PFLOP/s nominal peak that roofline.frac is priced against
    files_m = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mfma_shape_probe.log")))
    if files_m:
        try:
            import re
            best = {}
            for line in open(files_m[-1]):
                mo = re.match(r"(\S+ f16)(, registers only| \+ 12 ds_read_b128)\s+rep \d+:\s+[\d.]+ ms\s+([\d.]+) TFLOP/s", line)
                if mo:
                    k = mo.group(1).split()[0] + ("_registers_only" if "registers" in mo.group(2) else "_with_lds_fragment_reads")
                    best[k] = max(best.get(k, 0.0), float(mo.group(3)))
            if best:
                out["sustained_mfma_probe_tflops"] = best
                out["sustained_mfma_probe_source"] = os.path.relpath(files_m[-1], ROOT)
        except Exception:
            pass
    # the framework's own fused attention (torch SDPA, whichever backend the image's PyTorch-ROCm offers) at the path's shape beside
    # this repo's attention launch, one process, same operands (tools/bench_vendor_attention.py): yardstick only, never on the product path
    files_a = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_vendor_attention.json")))
    if files_a:
        try:
            d = json.load(open(files_a[-1]))
            y = {}
            for dt in ("fp16", "bf16"):
                rs = [r for r in d["results"] if r["dtype"] == dt and "ms_min" in r]
                ours = [r for r in rs if r["impl"].startswith("ours")]
                biased = [r for r in rs if r["impl"].startswith("sdpa") and r.get("mask", "").startswith("key bias") and "math" not in r["impl"]]
                plain = [r for r in rs if r["impl"].startswith("sdpa") and r.get("mask") == "no mask" and "math" not in r["impl"]]
                if ours and biased and plain:
                    b, p = min(biased, key=lambda r: r["ms_min"]), min(plain, key=lambda r: r["ms_min"])
                    y[dt] = {"this_kernel_with_key_bias_ms": ours[0]["ms_min"], "sdpa_best_with_key_bias_ms": b["ms_min"],
                             "sdpa_best_with_key_bias_backend": b["impl"], "sdpa_best_without_mask_ms": p["ms_min"],
                             "sdpa_best_without_mask_backend": p["impl"]}
            if y:
                out["vendor_attention_yardstick"] = dict(y, shape=d.get("shape"), torch=d.get("torch"),
                                                         note="stand-alone launches back to back (higher clock than inside the step)")
                out["vendor_attention_source"] = os.path.relpath(files_a[-1], ROOT)
        except Exception:
            pass
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json")) if "_c5" not in os.path.basename(f))
    if files:
        try:
            d = json.load(open(files[-1]))
            num = den = 0.0
            for k, v in d["per_kernel"].items():
                if k.startswith("gemm_kernel") and "mfma_busy" in v:
                    w = v["launches"] * v["avg_duration_ms"]
                    num += w * v["mfma_busy"]
                    den += w
            if den:
                out["gemm_mfma_busy"] = num / den
            out["shader_clock_ghz_under_load"] = d.get("shader_clock_ghz_under_load")
            out["sq_source"] = os.path.relpath(files[-1], ROOT)
        except Exception:
            pass
    return out


def archived_fp8_ranking(first_layer, qkv, down):
    """`ranking` block of the --fp8 line: what the e4m3 configuration of THIS run does to the fp32 reference's top-5
    on the committed bert-large ranking fixtures, next to what the reference's own bf16-autocast arithmetic does on the same lists.
    Archived from the device study (tests/tools/fp8_subset_study.py -> profiles/*_fp8_subset_study.json, seeded random-init weights; the
    GPU tests assert the same verdicts live: tests/test_gpu_fp8.py); not measured inside this process (the fixtures' generator is test
    infrastructure).  ranks_with_margin := on every list where the rule binds (the autocast reference keeps the fp32 top-5 with
    max |d| <= gap / 4) the top-5 set is kept and the centred drift is <= half the designed rank-5/6 gap."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fp8_subset_study.json")))
    if not files:
        return {"note": "no archived study under profiles/"}
    d = json.load(open(files[-1]))
    tag = f"first{first_layer}_qkv{qkv}_down{down}"
    lists, binding_ok, binding = {}, True, 0
    for k, v in d.items():
        fixture, q, t = k.split("/")
        if t != tag:
            continue
        ref = d.get(f"{fixture}/{q}/reference_autocast", {})
        lists[f"{fixture}/{q}"] = {"rule_binds": v["rule_binds"], "top5_set_kept": v["top5_set_kept"], "max_abs": v["max_abs"],
                                   "centred": v["centred"], "rho": v["rho"], "gap_5_6": v["gap_5_6"],
                                   "reference_bf16_autocast": {"max_abs": ref.get("max_abs"), "centred": ref.get("centred"),
                                                               "top5_set_kept": ref.get("top5_set_kept")}}
        if v["rule_binds"]:
            binding += 1
            binding_ok = binding_ok and v["top5_set_kept"] and v["centred"] <= 0.5 * v["gap_5_6"]
    out = {"configuration": {"fp8_first_layer": first_layer, "fp8_qkv": qkv, "fp8_ffn_down": down}, "source": os.path.relpath(files[-1], ROOT),
           "note": "archived device study on the committed ranking fixtures (seeded random-init bert-large, Linear matrices widened x1.5 - x2.5); "
                   "not measured in this run"}
    if not lists:
        out["verdict"] = "this configuration is not in the archived study"
        return out
    out["lists"] = lists
    out["binding_lists"] = binding
    out["ranks_with_margin"] = bool(binding and binding_ok)
    worst = max(lists.values(), key=lambda x: x["centred"] / x["gap_5_6"])
    out["verdict"] = ("keeps the fp32 top-5 with margin on every list the reference's own bf16 arithmetic ranks" if out["ranks_with_margin"] else
                      f"DOES NOT RANK on the fixtures: worst centred drift {worst['centred']:.3f} against a rank-5/6 gap of {worst['gap_5_6']:.3f} "
                      f"(rank correlation {worst['rho']:.2f}; the reference's own bf16-autocast drift there {worst['reference_bf16_autocast']['centred']:.3f})")
    return out


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


def cpu_baseline(arch, sd, S, vision, target_pairs):
    """The oracle (fp32 torch restatement of the reference forward) timed on this box's host cores, on a
    bounded sample of the same workload: ONE query of `target_pairs` candidates (K = 100 by default, the metric's own
    list length).  Reported beside the GPU number; never the thing shipped."""
    import torch
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
            "sample": f"one query of K={n} candidates of the same workload (S={S}, vision={vision}), fp32 torch oracle, "
                      f"{cores} threads, {dt:.1f} s"}


def pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    i = q * (len(xs) - 1)
    lo, hi = int(i), min(int(i) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (i - lo)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.rehearse_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ    # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")          # RCCL refuses two ranks on one device; the gather is staged through the host
        else:
            dist.init_process_group("nccl", device_id=dev)

    import rmr_amd
    from rmr_amd.sharding import sharded_forward
    from rmr_amd.synthetic import image_features, pair_batch

    wl = args.workload
    large = wl == "c5"
    vision = not args.text_only and not large
    shape = {}
    if large:
        shape = dict(hidden=1024, layers=24, heads=16, intermediate=4096, ce_hidden=1024, ce_heads=16, ce_intermediate=4096)
    if wl == "L":           # monoPreFLMR-L_pointwise.jsonnet:5-6,117: ViT-L/14 features, 900-long cross-encoder position table
        shape = dict(vision_hidden=1024, n_patches=256)
    loss_fn = "negative_sampling" if wl == "c4" else "BCE"
    arch = rmr_amd.make_arch(dict(cross_encoder_num_hidden_layers=1,
                                  cross_encoder_max_position_embeddings=900 if wl == "L" else 750,
                                  loss_fn=loss_fn, pos_weight=None), has_vision=int(vision),
                             compute_dtype=args.compute_dtype, **shape)
    if args.fp8:
        arch["fp8"] = 1
    for kv in args.tuning:
        k, v = kv.split("=")
        assert rmr_amd._lib.load().rr_set_tuning(k.encode(), int(v)) == 0, kv
    sd = rmr_amd.synthetic_state_dict(arch, seed=0, hf_init=True, gain=args.weights_gain)
    eng = rmr_amd.RerankEngine(arch, dev)
    eng.load_state_dict(sd)

    K, S = args.K, args.seq_len
    Bq = args.queries_per_gpu * (world if args.scaling == "weak" else 1)   # global queries per step
    N = Bq * K
    ids, am, tt = pair_batch(arch["vocab_size"], Bq, K, S, seed=2022, regime=args.regime)
    ids, am, tt = ids.to(dev), am.to(dev), tt.to(dev)
    cls = pat = None
    if vision:
        cls, pat = image_features(Bq, arch["n_patches"], arch["vision_hidden"])
        cls, pat = cls.to(dev), pat.to(dev)
    eng.reserve(-(-N // world) + 1, Bq, S, packed=bool(args.packed or args.bucketed))   # no allocation / synchronisation inside the steps

    host_lengths = None
    if args.packed:
        host_lengths = (((ids != 0) | (am != 0)) * torch.arange(1, S + 1, device=dev)).amax(1).clamp_(min=1).cpu().numpy()

    def step():
        if distributed:      # also with one rank: the same slice -> all-gather -> head path the N-GPU runs take
            # status words of the exchange are looked at when the next step begins / after the timed loop: no host read per step
            return sharded_forward(eng, ids, am, tt, Bq, K, cls, pat, None, want_scores=True, defer_status=True)
        if args.bucketed:
            return eng.forward_ids_bucketed(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)
        if args.packed:        # the pair lengths are host data in the real pipeline (the tokenizer produced them)
            return eng.forward_ids_packed(ids, am, tt, Bq, K, cls, pat, None, granule=args.granule, want_scores=True, want_order=True,
                                          lengths=host_lengths, segment_cost_rows=args.segment_cost_rows)
        return eng.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)

    def fence():
        if distributed:
            from rmr_amd.sharding import check_deferred_status
            check_deferred_status()                    # raises on every rank if any rank's encoder failed in an earlier step
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    if args.graph:
        assert not distributed and not args.bucketed and not args.packed, "--graph: plain single-process step only"
        args.no_profile = True
        gst = torch.cuda.Stream()
        with torch.cuda.stream(gst):
            eng.reserve(N + 1, Bq, S)                  # the redo-flag buffer is per stream
            step()                                     # function attributes etc. on the capture stream
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=gst):
                gout = step()
        eager_step = step

        def step():                                    # noqa: F811 — replay on the current stream
            graph.replay()
            return gout
        ref = eager_step()
        step()
        torch.cuda.synchronize(dev)
        assert torch.equal(ref["logits"], gout["logits"]), "graph replay differs from the eager step"
    redo = None
    if args.weights_gain != 1.0:      # one untimed step with the redo counters on: how often the fixed-reference attention falls back
        cnt = torch.zeros(2, dtype=torch.int64, device=dev)
        rmr_amd._lib.load().rr_set_attn_redo_stats(cnt.data_ptr())
        step()
        fence()
        rmr_amd._lib.load().rr_set_attn_redo_stats(0)
        redo = {"workgroups_recomputed_online": int(cnt[0]), "workgroups": int(cnt[1]),
                "fraction": float(cnt[0]) / max(1, int(cnt[1])), "weights_gain": args.weights_gain,
                "note": "fixed-reference attention: a workgroup whose row sum leaves the operand type's range (or that sees no valid "
                        "key) is recomputed with the online softmax by the second launch; counted over one step"}
    eng.set_profiling(not args.no_profile)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        out = step()
        ev[i][1].record()
    fence()
    dt = time.perf_counter() - t0
    eng.set_profiling(False)
    prof = eng.get_profile(reset=True) if not args.no_profile else None
    step_ms = [a.elapsed_time(b) for a, b in ev]
    if distributed:
        t = torch.tensor([dt], device="cpu" if args.rehearse_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out["logits"]).all()
    nranks_seen = dist.get_world_size() if distributed else 1

    if rank == 0:
        pairs_per_s = N * args.steps / dt
        fpp = flops_per_pair(arch, S, vision)
        P = (arch["prefix_len"] + arch["n_patches"]) if vision else 0
        names = {"c3": "c3: FLMR multimodal query cross-encoder rerank (monoPreFLMR-B shape, Lc=1), pointwise BCE head",
                 "c4": "c4: listwise rerank head (negative_sampling: softmax over K) on the c3 encoder",
                 "c5": "c5: bert-large cross-encoder rerank (24 layers, hidden 1024, FFN 4096, text-only)"
                       + (", e4m3 QKV/FFN-up GEMMs in the layers config.fp8_layers names" if args.fp8 else ", 16-bit MFMA"),
                 "L": "L: monoPreFLMR-L geometry (ViT-L/14 features, 288 vision tokens, T=800), pointwise BCE head"}
        res = {
            "metric": "reranked query x candidate pairs/sec at K=100, seq_len=512",
            "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "fp8+" + args.compute_dtype if args.fp8 else args.compute_dtype, "data": "synthetic",
            "config": {"workload": (names[wl] if not args.text_only else "c3-text: text-only cross-encoder rerank (Lc=1)")
                       + f", K={K}, seq_len={S}, vision_tokens={P}",
                       "queries_per_step": Bq, "pairs_per_step": N, "token_regime": args.regime,
                       **({"execution": f"length-bucketed: {int(out['bucket_rows'])} of {N * S} padded rows computed per step"}
                          if args.bucketed else
                          {"execution": f"packed rows (granule {args.granule}, segment_cost_rows {args.segment_cost_rows}, pair lengths known on the host): "
                                        f"{int(out['packed_rows'])} of {N * S} padded rows computed per step in {int(out.get('packed_segments', 0))} segments"}
                          if args.packed else {}),
                       "parallelism": (f"REHEARSAL: {world} ranks time-sharing ONE GPU, logits exchanged over gloo through the host; "
                                       "not a scaling measurement") if args.rehearse_one_gpu else
                                      (f"pairs sharded over {world} GPU(s) ({nranks_seen} ranks in the process group), "
                                       "1 RCCL all-gather of logits/step") if distributed else
                                      "1 GPU, no process group, no collective (the N > 1 legs shard the pairs over ranks with one "
                                      "RCCL all-gather of logits per step)",
                       "cross_encoder_last_layer": "all layers run; in the cross-encoder's (single, last) layer K/V are projected for every row and the "
                                                   "query / attention-output / LayerNorm / FFN work is done for the CLS row of each pair only - "
                                                   "the only row the classifiers read (utils.py:105-108); logits equal the all-rows computation up "
                                                   "to rounding; all_cross_encoder_rows_mode = the same step with every row computed",
                       **({"launch": "the step is ONE captured HIP graph, replayed"} if args.graph else {}),
                       "weights": "seeded random init (HF init), fp32 master -> 16-bit MFMA operands"
                                  + (f"; Linear matrices widened x{args.weights_gain} (peaked attention)" if args.weights_gain != 1.0 else "")},
            "step_ms_device": {"median": pct(step_ms, 0.5), "p10": pct(step_ms, 0.1), "p90": pct(step_ms, 0.9),
                               "n": len(step_ms), "note": "HIP events around each timed step on the work stream, rank 0"},
            "parity": {"bf16": "|logit - fp32 stock-HF| <= max(1e-3, the bf16-autocast reference's own drift) on every golden "
                               "(tests/test_gpu_forward.py, tests/golden/autocast.npz)",
                       "fp16": "|logit - fp32 stock-HF| <= 1e-3 on every golden (compute_dtype=fp16; throughput below)",
                       "mode_meeting_1e-3_vs_fp32": "fp16", "this_line": args.compute_dtype},
            "gflop_per_pair": fpp / 1e9,
            "gflop_per_pair_note": "SURVEY.md 8d: the REFERENCE's algorithmic FLOPs (every row of every layer); the library computes the "
                                   "cross-encoder's last layer behind its K/V projection for the CLS rows only, so whole_path_* below are "
                                   "reference-equivalent rates, not executed FLOPs (roofline.achieved counts executed FLOPs per launch)",
            "whole_path_tflops_per_gpu": pairs_per_s * fpp / 1e12 / world,
            "whole_path_frac_of_bf16_peak": pairs_per_s * fpp / 1e12 / world / PEAK_BF16_TFLOPS,
        }
        if redo is not None:
            res["attention_redo"] = redo
        if args.fp8:
            fl, fq, fd = (eng.get_option(k) for k in ("fp8_first_layer", "fp8_qkv", "fp8_ffn_down"))
            res["config"]["fp8_layers"] = (f"e4m3 QKV / FFN-up operands in text-encoder layers {fl}..{arch['layers'] - 1} of {arch['layers']} "
                                           f"(handle options fp8_first_layer={fl}, fp8_qkv={fq}, fp8_ffn_down={fd}); the shipped default is the last "
                                           "layer, the largest subset that ranks with a robust margin; --tuning fp8_first_layer=0 = the whole stack")
            res["ranking"] = archived_fp8_ranking(fl, fq, fd)
        if prof is not None:
            g = prof["gemm"]
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
            apm = archived_pmc() if wl in ("c3", "c4") else {}
            res["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel_hp (16-bit MFMA GEMM, persistent half-tile LDS ring, fused epilogues; all GEMM launches of the step)",
                               "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_BF16_TFLOPS,
                               "traffic": apm.get("traffic_gb_per_gemm_launch"),
                               "traffic_unit": "GB per launch beyond L2 (FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)",
                               "traffic_source": apm.get("traffic_source"),
                               "algorithmic_gb_per_launch": g["bytes"] / max(1, g["launches"]) / 1e9,
                               "launches": g["launches"], "avg_launch_ms": g["ms"] / max(1, g["launches"]),
                               "avg_launch_gflop": g["flops"] / max(1, g["launches"]) / 1e9,
                               "note": "live: per-launch HIP events on the work stream INSIDE the timed region, rank 0 "
                                       "(they cost ~1 % of the step; --no-profile removes them)"}
            sus = (apm.get("sustained_mfma_probe_tflops") or {}).get("16x16x32_with_lds_fragment_reads")
            if sus:
                res["roofline"]["frac_of_sustained_mfma_probe"] = ach / sus
                res["roofline"]["sustained_mfma_probe_note"] = (f"{sus:.0f} TFLOP/s = what v_mfma_f32_16x16x32_f16 sustains on this chip under its power cap "
                                                                "with the main loop's LDS fragment reads beside it and nothing else (archived probe, "
                                                                "archived_pmc.sustained_mfma_probe_source); frac stays priced against the 2.5 PFLOP/s nominal peak")
            g8 = prof.get("gemm_fp8")
            if g8 and g8["launches"]:
                a8 = g8["flops"] / (g8["ms"] * 1e-3) / 1e12
                res["roofline_fp8"] = {"bound": "mfma", "kernel": "gemm_kernel_hp8 (e4m3 GEMM on v_mfma_scale_f32_32x32x64_f8f6f4, persistent "
                                       "half-tile LDS ring, per-row x per-channel scales; the QKV and FFN-up launches of the step)",
                                       "achieved": a8, "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s", "frac": a8 / PEAK_FP8_TFLOPS,
                                       "launches": g8["launches"], "avg_launch_ms": g8["ms"] / g8["launches"],
                                       "algorithmic_gb_per_launch": g8["bytes"] / g8["launches"] / 1e9, "traffic": None}
            res["archived_pmc"] = apm
            tot = sum(v["ms"] for v in prof.values())
            res["kernel_time_share"] = {k: (v["ms"] / tot if tot else 0.0) for k, v in prof.items()}
            res["kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in prof.items()}
            res["kernel_launches_per_step"] = {k: v["launches"] / args.steps for k, v in prof.items()}
            a = prof["attention"]
            res["attention_tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
            # FLOPs of the launches that RAN (GEMMs + attention of this rank), over the wall time of the timed steps: the
            # executed rate beside the reference-equivalent whole_path_* (which counts the rows of the cross-encoder layer
            # the library does not compute)
            exe = sum(v["flops"] for v in prof.values())
            res["executed_tflops_per_gpu"] = exe / dt / 1e12
            res["executed_frac_of_bf16_peak"] = exe / dt / 1e12 / PEAK_BF16_TFLOPS
            res["executed_gflop_per_pair"] = exe / max(1, (N // world) * args.steps) / 1e9
        if world == 1 and not args.no_e2e:
            # the window the reference prints as "Rerank time" (Reranker_base_executor.py:898-939), for ONE query of K
            # candidates: host ids -> device, forward, logits/order back to the host (the sort runs on the device)
            h_ids, h_am, h_tt = (x[:K].cpu().pin_memory() for x in (ids, am, tt))
            h_img = (cls[:1].cpu().pin_memory(), pat[:1].cpu().pin_memory()) if vision else (None, None)
            e2e = []
            for it in range(12):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                d = [x.to(dev, non_blocking=True) for x in (h_ids, h_am, h_tt)]
                di = [x.to(dev, non_blocking=True) for x in h_img] if vision else [None, None]
                r = eng.forward_ids(d[0], d[1], d[2], 1, K, di[0], di[1], None, want_order=True)
                lg, od = r["logits"].cpu(), r["order"].cpu()        # D2H; synchronises
                ranked = od[0].tolist()
                if it >= 2:
                    e2e.append(1e3 * (time.perf_counter() - t1))
            assert len(ranked) == K and len(lg) == K
            res["end_to_end_ms_per_query"] = {"median": pct(e2e, 0.5), "p10": pct(e2e, 0.1), "p90": pct(e2e, 0.9), "n": len(e2e),
                                              "window": "pinned host ids/masks/features -> HBM, rr_forward (K pairs, device-side "
                                                        "stable top-K), logits + order -> host; the reference's 'Rerank time' "
                                                        "window (1.40 s per query published for monoPreFLMR-B on its GPU)"}
        if world == 1 and not args.no_alt_dtype and not args.bucketed and not args.packed:
            # The library computes the cross-encoder's last layer behind its K / V projection for the CLS rows only (the classifiers
            # read hidden state [:, 0]; same logits up to rounding).  The same step with every row of that layer computed, as the
            # reference does, is reported beside the headline (rr_set_option "ce_cls_only" 0).
            eng.set_option("ce_cls_only", 0)                       # a handle option: nothing process-wide changes
            try:
                for _ in range(max(1, args.warmup)):
                    eng.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    eng.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)
                torch.cuda.synchronize(dev)
                res["all_cross_encoder_rows_mode"] = {"value": N * args.steps / (time.perf_counter() - t1), "unit": "pairs/s",
                                                      "note": "the last cross-encoder layer computed for all rows (handle option ce_cls_only = 0)"}
            finally:
                eng.set_option("ce_cls_only", -1)
        if world == 1 and not args.no_alt_dtype and not args.fp8:
            # the same kernels with the other 16-bit operand type
            alt = "bf16" if args.compute_dtype == "fp16" else "fp16"
            del eng
            torch.cuda.empty_cache()
            eng2 = rmr_amd.RerankEngine(dict(arch, compute_dtype=alt), dev)
            eng2.load_state_dict(sd)
            for _ in range(max(1, args.warmup)):
                eng2.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng2.forward_ids(ids, am, tt, Bq, K, cls, pat, None, want_scores=True, want_order=True)
            torch.cuda.synchronize(dev)
            res[alt + "_operand_mode"] = {"value": N * args.steps / (time.perf_counter() - t1), "unit": "pairs/s",
                                          "note": res["parity"][alt]}
            del eng2
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(arch, sd, S, vision, args.cpu_pairs)
            res["gpu_over_cpu"] = pairs_per_s / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
