"""Multi-GPU execution of one rerank batch: candidate pairs are independent, so the N = Bq*K pairs are
split into contiguous per-rank slices (weights replicated), every rank runs the encoder on its slice, the
fp32 logits are exchanged with ONE `all_gather_into_tensor` (RCCL over xGMI when the backend is "nccl" on
ROCm; gloo in the CPU tests) and the scoring head + top-K run redundantly on every rank.

The reference has no such exchange (DDP shards queries, every rank writes its own JSON —
/root/reference/src/executors/Reranker_base_executor.py:1118-1121); this is the new collective of SURVEY §8e.

Buffers: the send slice, the gathered block and the compacted logits (+ status words) are allocated once per
(device, world size, slice length, heads) and reused by every later batch of that shape — no allocation and no
`torch.cat` in the steady state.  Both heads of the 2H_BCE variant travel in the same collective.  The collective is
enqueued by `torch.distributed` behind the work stream's kernels (ProcessGroupNCCL waits on the current stream's
event and the current stream waits on the collective), so the encoder of the next batch may already be enqueued.

Errors are COLLECTIVE: an exception inside one rank's encoder — e.g. the sticky RR_ERR_RANGE of a handle whose
pair slice left the fp16 range, which depends on that rank's own pairs — must not leave the other ranks waiting in the
all-gather.  Every rank's block therefore ends in one status word (0 ok / 1 range error / 2 any other error); a failing rank
still takes part in the exchange (zero logits, its status word set) and every rank raises after it: the failing rank its own
exception, the others `ShardPeerError` (an OverflowError when the peer's was one) naming the ranks.  Reading the status words
is one host read of `world` floats per batch; a loop that must not synchronise per batch passes `defer_status=True`: the words
are folded into a device-side running maximum (no allocation, no host read) and NOBODY raises — a rank whose encoder failed keeps
taking part with zero logits and its status word set — until `check_deferred_status()`, which such a loop calls before it uses its
results (and which the next non-deferred call runs first): there every rank reads the same maximum and raises together.
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


STATUS_OK, STATUS_RANGE, STATUS_ERROR = 0, 1, 2


class ShardPeerError(RuntimeError):
    """Another rank's encoder failed in this batch (its own process carries the original exception)."""


class ShardPeerRangeError(ShardPeerError, OverflowError):
    """Another rank's handle reported RR_ERR_RANGE (OverflowError there): the batch's logits are unreliable on every rank."""


class _GatherPlan:
    """Preallocated buffers of one (device, dtype, world, n_pairs, n_heads) exchange.  A rank's block is
    [head 0: per | head 1: per | status word]."""

    def __init__(self, device, dtype, world: int, n_pairs: int, heads: int):
        self.world, self.n, self.heads = world, n_pairs, heads
        self.per = -(-n_pairs // world)                                  # ceil(N / W): the padded slice length
        self.blk = heads * self.per + 1                                  # + the status word
        self.host = None
        self.send = torch.zeros(self.blk, dtype=dtype, device=device)
        self.sent_status = STATUS_OK                                     # what send[-1] holds (rewritten only when it changes: no launch in the steady state)
        self.recv = torch.empty(world * self.blk, dtype=dtype, device=device)
        # compaction map: full[h][i] = recv[rank(i), h, i - begin(rank)], then the `world` status words — one index_select, no cat
        idx = torch.empty(heads, n_pairs, dtype=torch.int64)
        for r in range(world):
            b, e = shard_range(n_pairs, r, world)
            for h in range(heads):
                idx[h, b:e] = torch.arange(e - b) + r * self.blk + h * self.per
        status = torch.arange(world, dtype=torch.int64) * self.blk + (self.blk - 1)
        self.index = torch.cat([idx.reshape(-1), status]).to(device)
        self.full = torch.empty(heads * n_pairs + world, dtype=dtype, device=device)
        self.status_accum = torch.zeros(world, dtype=dtype, device=device)   # deferred mode: running maximum of the status words
        self.status_host = torch.empty(world, dtype=dtype, device="cpu")
        if torch.device(device).type == "cuda":
            self.status_host = self.status_host.pin_memory()


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


def gather_logits(local: torch.Tensor, n_pairs: int, group=None, local2: Optional[torch.Tensor] = None, status: int = STATUS_OK,
                  want_status: bool = False, defer_status: bool = False):
    """All-gather the ragged per-rank logit slices (and, for the two-head variant, the second head's) into the full
    [n_pairs] vector(s) with ONE collective.  Returns (full, full2 | None) — and, with `want_status`, the `world` status
    words every rank sent (a device tensor of the logits' dtype) as a third element; the tensors are views of buffers that the
    next call with the same shape overwrites.  `defer_status`: fold the status words into the plan's running maximum for
    check_deferred_status() instead."""
    world = dist.get_world_size(group)
    heads = 1 if local2 is None else 2
    p = _plan(local.device, local.dtype, world, n_pairs, heads, group)
    n_loc = local.numel()
    p.send[:n_loc].copy_(local.reshape(-1))
    if local2 is not None:
        p.send[p.per: p.per + n_loc].copy_(local2.reshape(-1))
    if status != p.sent_status:
        p.send[p.blk - 1] = float(status)
        p.sent_status = status
    if p.send.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal on one GPU (bench.py --rehearse-one-gpu): gloo has no device all-gather, stage through the host
        if p.host is None:
            p.host = (torch.empty_like(p.send, device="cpu").pin_memory(), torch.empty_like(p.recv, device="cpu").pin_memory())
        p.host[0].copy_(p.send)
        dist.all_gather_into_tensor(p.host[1], p.host[0], group=group)
        p.recv.copy_(p.host[1])
    else:
        dist.all_gather_into_tensor(p.recv, p.send, group=group)
    torch.index_select(p.recv, 0, p.index, out=p.full)
    full1, full2 = p.full[:n_pairs], (p.full[n_pairs:2 * n_pairs] if heads == 2 else None)
    if defer_status:
        torch.maximum(p.status_accum, p.full[heads * n_pairs:], out=p.status_accum)
        _PENDING[id(p)] = (p, dist.get_rank(group))
    return (full1, full2, p.full[heads * n_pairs:]) if want_status else (full1, full2)


_PENDING: Dict[int, tuple] = {}      # plans whose running status maximum has not been looked at: id -> (plan, rank)
_OWN_DEFERRED: list = []             # this rank's own encoder exceptions of deferred batches (the first one is re-raised)


def check_deferred_status():
    """Look at what the `defer_status` batches since the last check reported: one host read per plan in use.  Every rank holds the
    same maxima, so every rank raises here together (the rank whose encoder failed its own first exception, its peers
    ShardPeerError) or none does.  The results of the batches since the failure are to be discarded by the caller."""
    own = _OWN_DEFERRED[0] if _OWN_DEFERRED else None
    failed = None
    for p, rank in list(_PENDING.values()):
        p.status_host.copy_(p.status_accum)               # synchronises with the stream the gathers ran on
        words = p.status_host.tolist()
        p.status_accum.zero_()
        if failed is None and any(int(w) != STATUS_OK for w in words):
            failed = (words, rank)
    _PENDING.clear()
    _OWN_DEFERRED.clear()
    if failed is not None:
        _raise_collectively(failed[0], failed[1], own)
    elif own is not None:
        raise own


def _raise_collectively(status_words, rank: int, own: Optional[BaseException]):
    """Every rank sees the same status words, so every rank raises (or none does)."""
    bad = [(r, int(s)) for r, s in enumerate(status_words) if int(s) != STATUS_OK]
    if not bad:
        return
    if own is not None:
        raise own
    ranks = [r for r, _ in bad]
    if all(s == STATUS_RANGE for _, s in bad):
        raise ShardPeerRangeError(f"rank(s) {ranks} reported an activation-range error (RR_ERR_RANGE) in this batch; "
                                  f"this rank ({rank}) discards the gathered logits")
    raise ShardPeerError(f"rank(s) {ranks} failed in their encoder in this batch (status {dict(bad)}); this rank ({rank}) "
                         f"discards the gathered logits")


class ShardedReranker:
    """Wraps a per-rank `encode(pair_begin, pair_end) -> (logits_slice, logits2_slice | None)` and a
    `head(logits, logits2) -> dict` into the sharded forward.  `RerankEngine` provides both on the GPU; the gloo CPU
    tests plug in a CPU stand-in engine to cover the slicing / gather / head plumbing."""

    def __init__(self, encode: Callable, head: Callable, group=None, two_heads: Optional[bool] = None, empty: Optional[Callable] = None,
                 defer_status: bool = False):
        """`empty(n) -> tensor` makes the zero slice a failing rank sends (device / dtype of the logits); `two_heads` tells
        whether that rank must send a second head (both are only needed when `encode` can raise)."""
        self.encode, self.head, self.group, self.two_heads, self.empty = encode, head, group, two_heads, empty
        self.defer_status = defer_status

    def __call__(self, n_pairs: int):
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        if not self.defer_status and (_PENDING or _OWN_DEFERRED):
            check_deferred_status()         # earlier deferred batches are settled before a batch that reports immediately
        b, e = shard_range(n_pairs, rank, world)
        own, status = None, STATUS_OK
        try:
            l1, l2 = self.encode(b, e)
        except Exception as ex:      # noqa: BLE001 — the exchange must still happen on this rank (see the module docstring)
            if self.empty is None:
                raise
            own, status = ex, (STATUS_RANGE if isinstance(ex, OverflowError) else STATUS_ERROR)
            l1 = self.empty(e - b)
            l2 = self.empty(e - b) if self.two_heads else None
        full1, full2, st = gather_logits(l1, n_pairs, self.group, l2, status=status, want_status=True, defer_status=self.defer_status)
        if self.defer_status:
            if own is not None:
                _OWN_DEFERRED.append(own)                 # raised by check_deferred_status(), together with the peers
        else:
            _raise_collectively(st.tolist(), rank, own)   # one host read of `world` floats per batch
        return self.head(full1, full2)


def sharded_forward(engine, input_ids, attention_mask, token_type_ids, Bq: int, K: int, image_cls=None,
                    image_patches=None, labels: Optional[torch.Tensor] = None, group=None, want_scores=False,
                    defer_status: bool = False):
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

    def empty(n):
        return torch.zeros(n, dtype=torch.float32, device=input_ids.device)

    return ShardedReranker(encode, head, group, two_heads=two_heads, empty=empty, defer_status=defer_status)(N)
