"""Multi-GPU execution of one rerank batch: candidate pairs are independent, so the N = Bq*K pairs are
split into contiguous per-rank slices (weights replicated), every rank runs the encoder on its slice, the
fp32 logits are exchanged with ONE `all_gather_into_tensor` (RCCL over xGMI when the backend is "nccl" on
ROCm; gloo in the CPU tests) and the scoring head + top-K run redundantly on every rank.

The reference has no such exchange (DDP shards queries, every rank writes its own JSON —
/root/reference/src/executors/Reranker_base_executor.py:1118-1121); this is the new collective of SURVEY §8e.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [begin, end) of rank `rank`; sizes differ by at most one, earlier ranks larger."""
    q, r = divmod(n_pairs, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


def gather_logits(local: torch.Tensor, n_pairs: int, group=None) -> torch.Tensor:
    """All-gather ragged per-rank logit slices into the full [n_pairs] vector with one collective:
    slices are padded to ceil(N/W) so a single all_gather_into_tensor suffices."""
    world = dist.get_world_size(group)
    per = -(-n_pairs // world)
    buf = torch.zeros(per, dtype=local.dtype, device=local.device)
    buf[: local.numel()] = local
    out = torch.empty(per * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = []
    for r in range(world):
        b, e = shard_range(n_pairs, r, world)
        parts.append(out[r * per: r * per + (e - b)])
    return torch.cat(parts)


class ShardedReranker:
    """Wraps a per-rank `encode(pair_begin, pair_end) -> (logits_slice, logits2_slice)` and a
    `head(logits, logits2) -> dict` into the sharded forward.  `RerankEngine` provides both on the GPU; the
    gloo CPU tests plug in stand-ins to cover the slicing / gather logic."""

    def __init__(self, encode: Callable, head: Callable, group=None):
        self.encode, self.head, self.group = encode, head, group

    def __call__(self, n_pairs: int):
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        b, e = shard_range(n_pairs, rank, world)
        l1, l2 = self.encode(b, e)
        full1 = gather_logits(l1, n_pairs, self.group)
        full2 = gather_logits(l2, n_pairs, self.group) if l2 is not None else None
        return self.head(full1, full2)


def sharded_forward(engine, input_ids, attention_mask, token_type_ids, Bq: int, K: int, image_cls=None,
                    image_patches=None, labels: Optional[torch.Tensor] = None, group=None, want_scores=False):
    """One rerank batch over all ranks of `group` with `engine` (a RerankEngine) on each rank."""
    N = Bq * K
    two_heads = engine.arch["loss_fn"] == "2H_BCE"

    def encode(b, e):
        r = engine.forward_ids(input_ids, attention_mask, token_type_ids, Bq, K, image_cls, image_patches, None,
                               pair_range=(b, e), want_loss=False)
        return r["logits"][b:e], (r["logits2"][b:e] if two_heads else None)

    def head(l1, l2):
        out = engine.head(l1, l2, labels, Bq, K, want_scores=want_scores, want_order=True)
        out["logits"] = l1
        return out

    return ShardedReranker(encode, head, group)(N)
