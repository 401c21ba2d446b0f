"""Synthetic benchmark inputs of the shape the reference's executor feeds the reranker:
tokenised (query, candidate) pairs `[CLS] q.. [SEP] ctx.. [SEP] pad*` with pair-encoding token types, and
CLIP-shaped image features per query.  Seeded numpy PCG64 (seed 2022, echoing hpc_meta_config.libsonnet:8)."""
from __future__ import annotations

import numpy as np
import torch


def pair_batch(vocab_size: int, n_queries: int, K: int, S: int, seed: int = 2022, regime: str = "full",
               q_len: int = 32):
    """int64 (input_ids, attention_mask, token_type_ids) [n_queries*K, S], query-major.
    regime "full": every position is a real token (throughput headline); "realistic": total length
    ~ U[64, S] with zero padding behind it (exercises the masks)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    N = n_queries * K
    ids = np.zeros((N, S), dtype=np.int64)
    tt = np.zeros((N, S), dtype=np.int64)
    lo = min(1000, vocab_size // 2)
    ql = min(q_len, max(1, S // 4))
    for n in range(N):
        L = S if regime == "full" else int(rng.integers(min(64, S // 2), S + 1))
        L = max(L, ql + 4)
        row = rng.integers(lo, vocab_size, size=L)
        row[0], row[ql + 1], row[L - 1] = 101, 102, 102
        ids[n, :L] = row
        tt[n, ql + 2: L] = 1
    am = (ids != 0).astype(np.int64)
    return torch.from_numpy(ids), torch.from_numpy(am), torch.from_numpy(tt)


def image_features(n_queries: int, n_patches: int, vision_hidden: int, seed: int = 2022):
    g = torch.Generator().manual_seed(seed + 7)
    return (torch.randn(n_queries, vision_hidden, generator=g),
            torch.randn(n_queries, n_patches, vision_hidden, generator=g))
