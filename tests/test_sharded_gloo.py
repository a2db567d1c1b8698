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


from oracle_shard_engine import OracleShardEngine  # noqa: E402  (tests/ is on sys.path under pytest's rootdir layout)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, dist_name, window, n_local, out, split="low", second=None, exchange="p2p", ragged=0,
            max_msg=700):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import htm_hashjoin_amd as hj
    from htm_hashjoin_amd.sharded import ShardedJoin
    n = n_local * world
    R = hj.generate_data(dist_name, n, n, window)
    S = hj.generate_data("sorted", n)
    # ragged: rank 0's piece is `ragged` tuples longer, the last rank's that much shorter (pieces stay contiguous)
    b = [r * n_local + (ragged if 0 < r < world else 0) for r in range(world + 1)]
    r_local = torch.from_numpy(R[b[rank]:b[rank + 1]].view(np.int64).copy())
    s_local = torch.from_numpy(S[b[rank]:b[rank + 1]].view(np.int64).copy())
    job = ShardedJoin(OracleShardEngine(), torch, dist, rank, world, split=split, exchange=exchange)
    job.max_msg_tuples = max_msg        # default: force the exchange into several messages per peer
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
        d["form"] = job.last_exchange_form
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
    assert got.pop("mode") == 0 and got.pop("sent") > 0 and got.pop("form") == "batch_isend_irecv"
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
    got.pop("form")
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
    got.pop("form")
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
    assert got.pop("mode") == (12 | 0x100) and got.pop("sent") == 0 and got.pop("form") is None
    assert got == oracle.sharded_reference(R, S, world, digit_shift=12, one_based=True)
    assert got["totalMatches"] == n


def _run(world, *args, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, PORT[0]) + args + (q,), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,msg_limit_hit", [(2, False), (4, False), (2, True)])
def test_exchange_as_one_all_to_all_single(world, msg_limit_hit):
    """exchange="a2a": ONE all_to_all_single per relation (the collective BASELINE config 4 names) moves the same keys
    to the same places as the pairwise batch; a step with a per-peer message over the limit falls back to the batch on
    every rank alike (the limit travels with the counts, no extra collective)."""
    n_local = 1 << 12
    got = _run(world, "uniform", 16, n_local, exchange="a2a", max_msg=700 if msg_limit_hit else 1 << 27)
    n = n_local * world
    R = oracle.generate_data("uniform", n, n, 16); S = oracle.generate_data("sorted", n)
    assert got.pop("form") == ("batch_isend_irecv" if msg_limit_hit else "all_to_all_single")
    assert got.pop("mode") == 0 and got.pop("sent") > 0
    assert got == oracle.sharded_reference(R, S, world)


def test_ragged_pieces_make_the_same_collective_calls():
    """Pieces of unequal length under the range split (round-1 ADVICE): whether a rank's piece fits its table is a
    rank-local fact, so it must travel INSIDE the reduced value, never decide whether the rank calls the collective.
    Rank 0 holds 512 tuples more than its share, the last rank 512 fewer: no hang, and the totals equal the sharded
    reference (tuples outside a rank's key range are exchanged)."""
    world, n_local = 2, 1 << 12
    for split in ("auto", "high"):
        got = _run(world, "sorted", 16, n_local, split=split, ragged=512)
        n = n_local * world
        R = oracle.generate_data("sorted", n); S = oracle.generate_data("sorted", n)
        got.pop("form")
        assert got.pop("mode") == (12 | 0x100) and got.pop("sent") == 2 * 512
        assert got == oracle.sharded_reference(R, S, world, digit_shift=12, one_based=True)


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


def _bench_worker(rank, world, port, out):
    """bench.py's own N > 1 leg (sharded.bench_sharded) with bench.py's own default flags, under gloo, the compute
    engine swapped for the checker-backed one."""
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import htm_hashjoin_amd as hj
    from htm_hashjoin_amd import sharded
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.argv = ["bench.py", "--gpus", str(world), "--log2n", "12", "--steps", "2", "--warmup", "1"]
    import bench
    line = sharded.bench_sharded(bench.parse(), torch, dist, hj, rank, world, 0, engine=OracleShardEngine(), device="cpu")
    if rank == 0:
        out.put(line)
    dist.destroy_process_group()


def test_default_bench_path_really_exchanges_and_reports_both_splits():
    """BASELINE config 4 is "radix join with RCCL all-to-all partition exchange": the default N > 1 bench line must be
    measured with tuples crossing ranks (low-bit split), carry the range split beside it as other_split, and report
    exchange and local time separately (round-1 VERDICT: the default moved nothing)."""
    import json
    world, n = 2, 1 << 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, PORT[0], q)) for r in range(world)]
    for p in procs:
        p.start()
    line = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    json.dumps(line)                                              # one JSON line's worth
    ex = line["exchange"]
    assert ex["split"] == "low key bits" and ex["sent_r"] + ex["sent_s"] > 0
    # about half of each relation leaves every rank at G = 2 (S = 1..n exactly; R's keys are random draws)
    assert abs(ex["sent_r"] - n // 2) < n // 16 and ex["sent_s"] == n // 2
    # --exchange auto (the default): the pre-flight self-check ran both forms and the faster correct one carried the steps
    assert line["exchange_chosen"] in ("a2a", "p2p") and all(sz[f]["ok"] for sz in line["a2a_selfcheck"].values() for f in ("a2a", "p2p"))
    assert ex["form"] == {"a2a": "all_to_all_single", "p2p": "batch_isend_irecv"}[line["exchange_chosen"]]
    assert ex["bytes_sent_per_rank_per_step"] == 4 * (ex["sent_r"] + ex["sent_s"])
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["config"]["per_gpu_rSize"] == n
    assert line["exchange_ms"] > 0 and line["local_ms"] > 0
    assert set(line["phase_ms"]) == {"histogram", "counts", "split", "exchange_r", "exchange_s", "build", "probe"}
    alt = line["other_split"]
    assert "range split" in alt["split"] and alt["sent_r"] == 0 and alt["sent_s"] == 0
    # totals of the headline = the sharded reference over the concatenated pieces (rank g draws from keys (g*n, (g+1)*n])
    from htm_hashjoin_amd.sharded import rank_key_range, squeeze_into_range
    Rs, Ss = [], []
    for g in range(world):
        lo, width = rank_key_range(world, n, g)
        Rs.append(squeeze_into_range(oracle.generate_data("uniform", n, n, 16), n, lo, width, np))
        Ss.append(squeeze_into_range(np.arange(1, n + 1, dtype=np.uint64), n, lo, width, np))
    want = oracle.sharded_reference(np.concatenate(Rs), np.concatenate(Ss), world)
    for k in ("conflicts", "totalMatches", "inputSum"):
        assert line["result"][k] == want[k], k
