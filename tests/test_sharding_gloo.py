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


class _CpuEngine:
    """CPU stand-in with RerankEngine's interface (forward_ids with pair_range / head): logit of pair i is a fixed
    function of its token ids, so any slicing or ordering mistake in sharded_forward shows up in the numbers."""

    def __init__(self, loss_fn):
        self.arch = {"loss_fn": loss_fn}
        self.calls = []

    def forward_ids(self, ids, am, tt, Bq, K, cls, pat, labels, pair_range=None, want_loss=True, **kw):
        b, e = pair_range
        self.calls.append((b, e))
        N = ids.shape[0]
        l1 = torch.full((N,), float("nan"))
        l2 = torch.full((N,), float("nan"))
        f = (ids[b:e].double() * torch.arange(1, ids.shape[1] + 1)).sum(1)
        l1[b:e] = torch.sin(f).float()
        l2[b:e] = torch.cos(f).float()
        return dict(logits=l1, logits2=l2)

    def head(self, l1, l2, labels, Bq, K, want_scores=False, want_order=True):
        assert torch.isfinite(l1).all()                      # every slice arrived
        ranked = l2 if self.arch["loss_fn"] == "2H_BCE" else l1
        order = torch.tensor([O.rank_descending_stable(r) for r in ranked.view(Bq, K).tolist()], dtype=torch.int32)
        return dict(order=order, scores=torch.sigmoid(ranked) if want_scores else None,
                    loss=ranked.double().sum().float())


def _engine_worker(rank, world, port, Bq, K, loss_fn, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rmr_amd.sharding import shard_range, sharded_forward
        ids = torch.randint(1, 1000, (Bq * K, 16), generator=torch.Generator().manual_seed(3))
        eng = _CpuEngine(loss_fn)
        outs = []
        for _ in range(2):                                   # second call reuses the preallocated gather buffers
            out = sharded_forward(eng, ids, ids, ids, Bq, K, want_scores=True)
            outs.append((out["logits"].clone(), None if out.get("logits2") is None else out["logits2"].clone(),
                         out["order"].clone()))
        b, e = shard_range(Bq * K, rank, world)
        assert eng.calls == ([(b, e)] * 2 if e > b else [])   # an empty slice never reaches the engine
        assert all(torch.equal(a, b) for a, b in zip(outs[0][:1], outs[1][:1])) and torch.equal(outs[0][2], outs[1][2])
        q.put((rank, outs[0][0].tolist(), None if outs[0][1] is None else outs[0][1].tolist(), outs[0][2].tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("Bq,K,loss_fn", [(2, 5, "BCE"), (3, 3, "2H_BCE"), (1, 101, "negative_sampling"), (1, 1, "BCE")])
def test_sharded_forward_with_an_engine_matches_single_rank(Bq, K, loss_fn):
    """world 2 over gloo through rmr_amd.sharding.sharded_forward itself: ragged N % world != 0, the two-head variant's
    second logit vector in the same collective, one pair for two ranks (an empty slice)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_engine_worker, args=(r, world, port, Bq, K, loss_fn, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = torch.randint(1, 1000, (Bq * K, 16), generator=torch.Generator().manual_seed(3))
    ref = _CpuEngine(loss_fn).forward_ids(ids, ids, ids, Bq, K, None, None, None, pair_range=(0, Bq * K))
    want = ref["logits2"] if loss_fn == "2H_BCE" else ref["logits"]
    want_order = [O.rank_descending_stable(r) for r in want.view(Bq, K).tolist()]
    for rank, l1, l2, order in res:
        assert l1 == ref["logits"].tolist()
        if loss_fn == "2H_BCE":
            assert l2 == ref["logits2"].tolist()
        assert order == want_order


# ---- world size 8: the width BASELINE configs[3] names (/root/reference/submit_test_jobs.py:74 launches one process per GPU) ----
def _run_world(target, world, args, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=timeout) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


@pytest.mark.parametrize("Bq,K,loss_fn", [(1, 100, "BCE"), (64, 100, "negative_sampling"), (2, 100, "2H_BCE")])
def test_eight_ranks_through_sharded_forward(Bq, K, loss_fn):
    """world 8 over gloo through sharded_forward: one query of K = 100 (the 13/12-pair ragged split of configs[2] strong-scaled,
    SURVEY 8e), configs[3]'s 64 queries x 100 = 6 400 pairs (800 per rank, listwise head) and the two-head variant."""
    world = 8
    res = _run_world(_engine_worker, world, (Bq, K, loss_fn))
    ids = torch.randint(1, 1000, (Bq * K, 16), generator=torch.Generator().manual_seed(3))
    ref = _CpuEngine(loss_fn).forward_ids(ids, ids, ids, Bq, K, None, None, None, pair_range=(0, Bq * K))
    want = ref["logits2"] if loss_fn == "2H_BCE" else ref["logits"]
    want_order = [O.rank_descending_stable(r) for r in want.view(Bq, K).tolist()]
    assert [r for r, *_ in res] == list(range(world))
    from rmr_amd import shard_range
    sizes = [e - b for b, e in (shard_range(Bq * K, r, world) for r in range(world))]
    assert sum(sizes) == Bq * K and (sizes == [13, 13, 13, 13, 12, 12, 12, 12] if Bq * K == 100 else max(sizes) - min(sizes) <= 1)
    for rank, l1, l2, order in res:
        assert l1 == ref["logits"].tolist()
        if loss_fn == "2H_BCE":
            assert l2 == ref["logits2"].tolist()
        assert order == want_order


class _FailingEngine(_CpuEngine):
    """Stand-in whose forward raises on ONE rank, as RerankEngine does for a handle with the sticky RR_ERR_RANGE
    (rmr_amd/_lib.py maps it to OverflowError) — a rank-local event, because it depends on the rank's own pair slice."""

    def __init__(self, loss_fn, fail, exc):
        super().__init__(loss_fn)
        self.fail, self.exc = fail, exc

    def forward_ids(self, *a, **kw):
        if self.fail:
            raise self.exc("RR_ERR_RANGE stand-in" if self.exc is OverflowError else "encoder failed")
        return super().forward_ids(*a, **kw)


def _failing_worker(rank, world, port, bad_rank, exc_name, defer, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rmr_amd.sharding import ShardPeerError, check_deferred_status, sharded_forward
        exc = {"OverflowError": OverflowError, "ValueError": ValueError}[exc_name]
        Bq, K = 2, 7
        ids = torch.randint(1, 1000, (Bq * K, 16), generator=torch.Generator().manual_seed(3))
        ok = _CpuEngine("BCE")
        out = sharded_forward(ok, ids, ids, ids, Bq, K)                              # a healthy batch first
        first = out["logits"].tolist()
        eng = _FailingEngine("BCE", rank == bad_rank, exc)
        raised = None
        try:
            if defer:
                # a hot loop: three batches enqueued back to back, nobody raises and nobody leaves the loop (the failing rank keeps
                # taking part with zero logits), then ONE check on which every rank raises together
                for _ in range(3):
                    sharded_forward(eng, ids, ids, ids, Bq, K, defer_status=True)
                check_deferred_status()
            else:
                sharded_forward(eng, ids, ids, ids, Bq, K)
        except Exception as ex:      # noqa: BLE001
            raised = (type(ex).__name__, isinstance(ex, OverflowError), isinstance(ex, ShardPeerError), str(ex))
        # nobody is stuck in a collective: the group still works afterwards, and so does the next healthy batch
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        again = sharded_forward(ok, ids, ids, ids, Bq, K)["logits"].tolist()
        q.put((rank, raised, float(t), first == again))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exc_name,defer", [("OverflowError", False), ("ValueError", False), ("OverflowError", True)])
def test_an_error_on_one_rank_is_raised_on_every_rank(exc_name, defer):
    """the sticky range error is rank-local; the exchange carries one status word per rank, the failing rank still
    takes part, and every rank raises after the gather (the failing one its own exception, its peers ShardPeerError — an
    OverflowError too when the peer's was) instead of waiting in a collective one rank never enters."""
    world, bad = 3, 1
    res = _run_world(_failing_worker, world, (bad, exc_name, defer), timeout=120)
    for rank, raised, total, same in res:
        assert raised is not None, f"rank {rank} did not raise"
        name, is_overflow, is_peer, msg = raised
        if rank == bad:
            assert name == exc_name and not is_peer
        else:
            assert is_peer and "[1]" in msg and is_overflow == (exc_name == "OverflowError")
        assert total == 3.0 and same
