"""The N > 1 path (htm_hashjoin_amd/sharded.py) under torch.distributed with the gloo backend on
CPU, world_size 2 and 4. The sharding / exchange logic is the product's; the per-rank compute engine
is replaced by an oracle-backed one (tests only) because there is no GPU here. The all-reduced
totals must equal oracle.sharded_reference() on the concatenated input, bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle


class OracleShardEngine:
    """CPU stand-in for HipShardEngine: same interface, numpy + the CPU oracle (TEST ONLY)."""

    def __init__(self):
        self._res = None

    def empty_keys(self, n):
        return torch.empty(int(n), dtype=torch.int32)

    @staticmethod
    def _dest(k, n_shards, mode):
        """hj_shard_histogram_dev: ((key - b) >> d) & (nShards - 1), d = mode & 0xFF, b = 1 if HJ_SHARD_ONE_BASED"""
        k32 = k.astype(np.uint32) - np.uint32(1 if mode & 0x100 else 0)
        return ((k32 >> np.uint32(mode & 0xFF)) & np.uint32(n_shards - 1)).astype(np.int64)

    def histogram(self, t, n_shards, mode=0):
        k = t.numpy().view(np.uint64)
        return torch.from_numpy(np.bincount(self._dest(k, n_shards, mode), minlength=n_shards)).to(torch.int64)

    def scatter(self, t, n_shards, counts, mode=0):
        """grouped by destination, input order kept inside each (what hj_shard_scatter_dev guarantees), keys only"""
        k = t.numpy().view(np.uint64)
        order = np.argsort(self._dest(k, n_shards, mode), kind="stable")
        return torch.from_numpy(k[order].astype(np.uint32).view(np.int32).copy())

    def max_key(self, t):
        return int(t.max().item()) if t.numel() else 0

    def reserve(self, table_size, max_r, max_s):
        pass

    def build(self, r_keys, home_shift, table_size):
        # position in the receive buffer = insertion order
        self._built = (r_keys.numpy().view(np.uint32).astype(np.uint64), home_shift, table_size)

    _check = None

    def set_check(self, n_shards, mode=0, shard_id=0):
        self._check = (n_shards, mode, shard_id) if n_shards else None

    def _count_foreign(self, k):
        if self._check is None:
            return 0
        n_shards, mode, shard_id = self._check
        return int((self._dest(k, n_shards, mode) != shard_id).sum())

    def build_tuples(self, r_tuples):
        k = r_tuples.numpy().view(np.uint64).copy()
        self._foreign = self._count_foreign(k)
        self._built = (k, 0, 2 * r_tuples.numel())

    def probe_tuples(self, s_tuples):
        keys, home_shift, table_size = self._built
        k = s_tuples.numpy().view(np.uint64).copy()
        self._foreign += self._count_foreign(k)
        self._res = oracle.build_probe_seq_ts(keys, k, table_size, home_shift)

    def foreign(self):
        return self._foreign

    def probe(self, s_keys):
        keys, home_shift, table_size = self._built
        s = s_keys.numpy().view(np.uint32).astype(np.uint64)
        self._res = oracle.build_probe_seq_ts(keys, s, table_size, home_shift)

    def finish(self):
        r = dict(self._res)
        r.update(buildVariant=0, buildDeferred=0, build_us=0.0, probe_us=0.0, clear_us=0.0, buildPhaseA_us=0.0)
        return r

    def sync(self):
        pass


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, dist_name, window, n_local, out, split="low", second=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import htm_hashjoin_amd as hj
    from htm_hashjoin_amd.sharded import ShardedJoin
    n = n_local * world
    R = hj.generate_data(dist_name, n, n, window)
    S = hj.generate_data("sorted", n)
    r_local = torch.from_numpy(R[rank * n_local:(rank + 1) * n_local].view(np.int64).copy())
    s_local = torch.from_numpy(S[rank * n_local:(rank + 1) * n_local].view(np.int64).copy())
    job = ShardedJoin(OracleShardEngine(), torch, dist, rank, world, split=split)
    job.max_msg_tuples = 700            # force the exchange into several messages per peer
    job.step(r_local, s_local, 2 * n_local)
    if second:                                       # a second step: the optimistic in-place attempt, or its fallback
        in_place_first = "in place" in job.last["split"]
        if second == "moved" and rank == 0:
            r_local[0] = n                           # the largest key belongs to the last rank: one tuple has to travel
        job.step(r_local, s_local, 2 * n_local)
        assert in_place_first
        assert ("checked inside build and probe" in job.last["split"]) == (second == "same"), job.last
    res = job.result()
    if rank == 0:
        d = {k: res[k] for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum")}
        d["mode"] = job.mode
        d["sent"] = res["exchange"]["sent_r"] + res["exchange"]["sent_s"]
        out.put(d)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("dist_name,window", [("uniform", 16), ("local_shuffle", 1024), ("random", 16)])
def test_sharded_join_matches_sharded_reference(world, dist_name, window):
    n_local = 1 << 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT[0], dist_name, window, n_local, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = n_local * world
    R = oracle.generate_data(dist_name, n, n, window)
    S = oracle.generate_data("sorted", n)
    want = oracle.sharded_reference(R, S, world)
    assert got.pop("mode") == 0 and got.pop("sent") > 0
    assert got == want
    if dist_name == "local_shuffle":
        assert got["conflicts"] == 0 and got["totalMatches"] == n


@pytest.mark.parametrize("split,dist_name,window,expect_high", [
    ("high", "uniform", 16, True), ("auto", "uniform", 16, True), ("auto", "local_shuffle", 1024, True),
    ("auto", "shuffle", 16, False), ("high", "shuffle", 16, True)])
def test_range_split(split, dist_name, window, expect_high):
    """Destination by the HIGH key bits (a range split): ranks that hold contiguous pieces of a near-sorted relation
    keep almost everything; "auto" takes it only then (a shuffled relation falls back to the low bits). Either way the
    totals equal the sharded reference for the digit that was used."""
    world, n_local = 4, 1 << 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT[0], dist_name, window, n_local, q, split)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = n_local * world
    R = oracle.generate_data(dist_name, n, n, window)
    S = oracle.generate_data("sorted", n)
    mode, sent = got.pop("mode"), got.pop("sent")
    if expect_high:
        assert mode == (12 | 0x100)                      # keys 1..2^14 over 4 ranks: digit of (key - 1) at bit 12
        want = oracle.sharded_reference(R, S, world, digit_shift=12, one_based=True)
        if dist_name != "shuffle":
            assert sent < n // 8, sent                   # near-sorted pieces: next to nothing crosses the wire
    else:
        assert mode == 0
        want = oracle.sharded_reference(R, S, world)
    assert got == want


@pytest.mark.parametrize("second", ["same", "moved"])
def test_optimistic_in_place_step_and_its_fallback(second):
    """After a step that moved nothing the next one joins in place straight away, the shard check riding on build and
    probe; if a tuple has meanwhile appeared that belongs elsewhere, every rank sees the non-zero count and the step is
    redone with split and exchange. Totals equal the sharded reference of the data of the second step either way."""
    world, n_local = 2, 1 << 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT[0], "sorted", 16, n_local, q, "auto", second)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = n_local * world
    R = oracle.generate_data("sorted", n); S = oracle.generate_data("sorted", n)
    if second == "moved":
        R[0] = n
    assert got.pop("mode") == (12 | 0x100)
    assert (got.pop("sent") > 0) == (second == "moved")
    assert got == oracle.sharded_reference(R, S, world, digit_shift=12, one_based=True)


def test_range_split_nothing_moves_joins_in_place():
    """`sorted` pieces under the range split: every tuple is already home -- the ranks join their pieces in place
    (no split, no exchange) and the totals still equal the sharded reference."""
    world, n_local = 2, 1 << 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT[0], "sorted", 16, n_local, q, "auto")) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = n_local * world
    R = oracle.generate_data("sorted", n); S = oracle.generate_data("sorted", n)
    assert got.pop("mode") == (12 | 0x100) and got.pop("sent") == 0
    assert got == oracle.sharded_reference(R, S, world, digit_shift=12, one_based=True)
    assert got["totalMatches"] == n


PORT = [0]


@pytest.fixture(autouse=True)
def _port():
    PORT[0] = _free_port()
    yield


def test_single_rank_is_the_plain_operator():
    """world == 1: no exchange, no home shift; totals equal the single-table oracle."""
    from htm_hashjoin_amd.sharded import ShardedJoin
    n = 1 << 12
    R = oracle.generate_data("uniform", n, n, 16)
    S = oracle.generate_data("sorted", n)
    job = ShardedJoin(OracleShardEngine(), torch, None, 0, 1)
    job.step(torch.from_numpy(R.view(np.int64).copy()), torch.from_numpy(S.view(np.int64).copy()), 2 * n)
    got = job.result()
    want = oracle.build_probe_seq(R, S)
    for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"):
        assert got[k] == want[k]
