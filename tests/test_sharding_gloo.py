"""CPU, world_size 2 over gloo: pair sharding + the single all-gather of logits + redundant head reproduce the
single-process result (the N>1 path of bench.py / rmr_amd.sharding with stand-in compute)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import O


def test_shard_range_partitions_exactly():
    from rmr_amd import shard_range
    for n in (1, 7, 100, 101, 800):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in rs]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, K, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rmr_amd.sharding import ShardedReranker, shard_range
        g = torch.Generator().manual_seed(0)
        table = torch.randn(n_pairs, generator=g)             # "logit of pair i" every rank could compute
        seen = []

        def encode(b, e):
            seen.append((b, e))
            return table[b:e].clone(), (table[b:e] * 2).clone()

        def head(l1, l2):
            Bq = n_pairs // K
            order = [O.rank_descending_stable(r) for r in l1.view(Bq, K).tolist()]
            return dict(logits=l1, logits2=l2, order=order)

        out = ShardedReranker(encode, head)(n_pairs)
        assert seen == [shard_range(n_pairs, rank, world)]
        q.put((rank, out["logits"].tolist(), out["logits2"].tolist(), out["order"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs,K", [(10, 5), (9, 3), (200, 100)])
def test_two_rank_gather_matches_single_process(n_pairs, K):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, K, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    table = torch.randn(n_pairs, generator=g)
    want_order = [O.rank_descending_stable(r) for r in table.view(n_pairs // K, K).tolist()]
    for rank, l1, l2, order in res:
        assert l1 == table.tolist() and l2 == (table * 2).tolist()
        assert order == want_order
