"""Multi-GPU execution of one rerank batch: candidate pairs are independent, so the N = Bq*K pairs are
split into contiguous per-rank slices (weights replicated), every rank runs the encoder on its slice, the
fp32 logits are exchanged with ONE `all_gather_into_tensor` (RCCL over xGMI when the backend is "nccl" on
ROCm; gloo in the CPU tests) and the scoring head + top-K run redundantly on every rank.

The reference has no such exchange (DDP shards queries, every rank writes its own JSON —
/root/reference/src/executors/Reranker_base_executor.py:1118-1121); this is the new collective of SURVEY §8e.

Buffers: the send slice, the gathered block and (ragged case only) the compacted logits are allocated once per
(device, world size, slice length, heads) and reused by every later batch of that shape — no allocation and no
`torch.cat` in the steady state.  Both heads of the 2H_BCE variant travel in the same collective.  The collective is
enqueued by `torch.distributed` behind the work stream's kernels (ProcessGroupNCCL waits on the current stream's
event and the current stream waits on the collective), so the encoder of the next batch may already be enqueued.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [begin, end) of rank `rank`; sizes differ by at most one, earlier ranks larger."""
    q, r = divmod(n_pairs, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


class _GatherPlan:
    """Preallocated buffers of one (device, dtype, world, n_pairs, n_heads) exchange."""

    def __init__(self, device, dtype, world: int, n_pairs: int, heads: int):
        self.world, self.n, self.heads = world, n_pairs, heads
        self.per = -(-n_pairs // world)                                  # ceil(N / W): the padded slice length
        self.host = None
        self.send = torch.zeros(heads * self.per, dtype=dtype, device=device)
        self.recv = torch.empty(world * heads * self.per, dtype=dtype, device=device)
        self.ragged = n_pairs % world != 0
        if self.ragged:      # compaction map: full[h][i] = recv[rank(i), h, i - begin(rank)] — one index_select, no cat
            idx = torch.empty(heads, n_pairs, dtype=torch.int64)
            for r in range(world):
                b, e = shard_range(n_pairs, r, world)
                for h in range(heads):
                    idx[h, b:e] = torch.arange(e - b) + (r * heads + h) * self.per
            self.index = idx.reshape(-1).to(device)
            self.full = torch.empty(heads * n_pairs, dtype=dtype, device=device)


_PLANS: "OrderedDict[tuple, _GatherPlan]" = None      # small LRU: a serving loop sees a handful of batch sizes (the full one, the last one)
_MAX_PLANS = 8


def _plan(device, dtype, world, n_pairs, heads, group=None) -> _GatherPlan:
    from collections import OrderedDict
    global _PLANS
    if _PLANS is None:
        _PLANS = OrderedDict()
    key = (str(device), dtype, world, n_pairs, heads, id(group) if group is not None else 0)
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = _GatherPlan(device, dtype, world, n_pairs, heads)
        while len(_PLANS) > _MAX_PLANS:
            _PLANS.popitem(last=False)
    else:
        _PLANS.move_to_end(key)
    return p


def gather_logits(local: torch.Tensor, n_pairs: int, group=None, local2: Optional[torch.Tensor] = None):
    """All-gather the ragged per-rank logit slices (and, for the two-head variant, the second head's) into the full
    [n_pairs] vector(s) with ONE collective.  Returns (full, full2 | None); the tensors are views of buffers that the
    next call with the same shape overwrites."""
    world = dist.get_world_size(group)
    heads = 1 if local2 is None else 2
    p = _plan(local.device, local.dtype, world, n_pairs, heads, group)
    n_loc = local.numel()
    p.send[:n_loc].copy_(local.reshape(-1))
    if local2 is not None:
        p.send[p.per: p.per + n_loc].copy_(local2.reshape(-1))
    if p.send.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on one GPU (bench.py --rehearse-one-gpu): gloo has no device all-gather, stage through the host
        if p.host is None:
            p.host = (torch.empty_like(p.send, device="cpu").pin_memory(), torch.empty_like(p.recv, device="cpu").pin_memory())
        p.host[0].copy_(p.send)
        dist.all_gather_into_tensor(p.host[1], p.host[0], group=group)
        p.recv.copy_(p.host[1])
    else:
        dist.all_gather_into_tensor(p.recv, p.send, group=group)
    if not p.ragged:
        if heads == 1:
            return p.recv, None                                       # rank-major == pair order: the block IS the vector
        blk = p.recv.view(world, 2, p.per)
        return blk[:, 0].reshape(-1), blk[:, 1].reshape(-1)            # two strided copies of N floats
    torch.index_select(p.recv, 0, p.index, out=p.full)
    return p.full[:n_pairs], (p.full[n_pairs:] if heads == 2 else None)


class ShardedReranker:
    """Wraps a per-rank `encode(pair_begin, pair_end) -> (logits_slice, logits2_slice | None)` and a
    `head(logits, logits2) -> dict` into the sharded forward.  `RerankEngine` provides both on the GPU; the gloo CPU
    tests plug in a CPU stand-in engine to cover the slicing / gather / head plumbing."""

    def __init__(self, encode: Callable, head: Callable, group=None):
        self.encode, self.head, self.group = encode, head, group

    def __call__(self, n_pairs: int):
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        b, e = shard_range(n_pairs, rank, world)
        l1, l2 = self.encode(b, e)
        full1, full2 = gather_logits(l1, n_pairs, self.group, l2)
        return self.head(full1, full2)


def sharded_forward(engine, input_ids, attention_mask, token_type_ids, Bq: int, K: int, image_cls=None,
                    image_patches=None, labels: Optional[torch.Tensor] = None, group=None, want_scores=False):
    """One rerank batch over all ranks of `group` with `engine` (a RerankEngine) on each rank.  Every rank holds the
    whole id tensors (they are 3 x 8 B per token — 1.2 MB per query of 100 x 512 — against ~10 TFLOP of encoder work) and
    encodes only its pair slice; image-derived per-query work is done for the queries the slice touches."""
    N = Bq * K
    two_heads = engine.arch["loss_fn"] == "2H_BCE"

    def encode(b, e):
        if e <= b:                       # more ranks than pairs: this rank contributes an empty slice
            z = torch.empty(0, dtype=torch.float32, device=input_ids.device)
            return z, (z if two_heads else None)
        r = engine.forward_ids(input_ids, attention_mask, token_type_ids, Bq, K, image_cls, image_patches, None,
                               pair_range=(b, e), want_loss=False)
        return r["logits"][b:e], (r["logits2"][b:e] if two_heads else None)

    def head(l1, l2):
        out = engine.head(l1, l2, labels, Bq, K, want_scores=want_scores, want_order=True)
        # the gathered vectors are views of the exchange buffers, which the next call of the same shape overwrites: hand the
        # caller its own N floats (an eval loop keeps the outputs of many batches)
        out["logits"] = l1.clone()
        if l2 is not None:
            out["logits2"] = l2.clone()
        return out

    return ShardedReranker(encode, head, group)(N)
