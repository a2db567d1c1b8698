"""Parity tests proper: the HIP engine, called through the C ABI, against the CPU
oracle on identical inputs. Bit-exact (integer work). Run with -m gpu on an MI355X."""
import json
import os

import numpy as np
import pytest

try:                                     # torch has to bring HIP up before the library does, or it finds no device later
    import torch
    _TORCH_GPU = bool(torch.cuda.is_available()) and torch.zeros(1, device="cuda:0").numel() == 1
except Exception:                        # noqa: BLE001 -- no torch / no GPU: only the in-process multi-rank test needs it
    _TORCH_GPU = False

import htm_hashjoin_amd as hj
from htm_hashjoin_amd import _lib
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hj.HashJoinContext(0)
    yield c
    c.close()


def check_oa(got, want, algo="atomic"):
    for k in ("conflicts", "totalMatches", "inputSum", "tableSumHalf", "tableSumFull", "conflictSum"):
        assert got[k] == want[k], (k, got[k], want[k])
    assert got["outputSum"] == (want["outputSumNocc"] if algo == "nocc" else want["outputSumAtomic"])


DISTS = [("uniform", 16), ("uniform", 2), ("random", 16), ("sorted", 16), ("shuffle", 16),
         ("local_shuffle", 16), ("local_shuffle", 1024)]


def _variant_that_runs(variant, table_size, compact_held=None):
    """hj_params.buildVariant -> the kernel that actually runs: 2 needs a table of one 8192-slot window, 3 and 4 of one
    1024-slot ring; below that they fall back (4 -> 3 -> 2 -> 1). 4 (the compact rings) reports 4 when the compact
    table held and 3 when the classic rings had to redo it (compact_held = hj_result.compactFallback == 0)."""
    if variant == 4 and (table_size < 1024 or not compact_held):
        variant = 3
    if variant == 3 and table_size < 1024:
        variant = 2
    if variant == 2 and table_size < 8192:
        variant = 1
    return variant


# inputs on which the compact ring build must hold (every tuple within reach of its wavefront's ring, walks across a seam
# seen by the next wavefront's shadow granule) / must hand over to the classic rings (no locality)
COMPACT_HOLDS = {("uniform", 16), ("uniform", 2), ("sorted", 16), ("local_shuffle", 16)}
COMPACT_FAILS = {("shuffle", 16), ("random", 16), ("local_shuffle", 4096), ("local_shuffle", 65536)}


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
@pytest.mark.parametrize("dist,window", DISTS + [("local_shuffle", 128), ("local_shuffle", 4096), ("local_shuffle", 65536)])
@pytest.mark.parametrize("n", [1 << 10, 1 << 16, 1 << 20])
def test_build_probe_matches_sequential_oracle(ctx, dist, window, n, variant):
    """variant 1 = global atomicMin kernel, 2 = block ownership + workgroup LDS window, 3 = wavefront-private LDS
    rings over static slot ranges (2 and 3 + the deferred phase), 4 = the same rings writing the compact 4-byte table
    (no deferred phase; the classic rings redo the table where the input needs one): all must give the table a single
    thread builds in input order, slot for slot -- with locality (where 2, 3 and 4 are meant to run) and without (where
    nearly every tuple takes the deferred road, and 4 hands over)."""
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.relS_for(dist, R)
    want = oracle.build_probe_seq(R, S, 4, want_table=True)
    got = ctx.run("atomic", R, S, buildVariant=variant)
    check_oa(got, want)
    assert got["buildVariant"] == _variant_that_runs(variant, 2 * n, got["compactFallback"] == 0)
    if variant == 4 and 2 * n >= 1024 and n >= 1 << 16:
        if (dist, window) in COMPACT_HOLDS:
            assert got["buildVariant"] == 4 and got["compactFallback"] == 0 and got["buildDeferred"] == 0
        if (dist, window) in COMPACT_FAILS:
            assert got["buildVariant"] == 3 and got["compactFallback"] != 0
    if variant != 4:
        assert got["compactFallback"] == 0
    if got["buildVariant"] >= 2 and n >= 1 << 16 and dist in ("sorted", "uniform"):
        assert got["buildDeferred"] < n // 16         # locality: the LDS window / rings take almost everything
    # the whole table, slot for slot, equals the table a single thread builds in input order
    assert np.array_equal(ctx.export_table(2 * n), want["table"])


def test_golden_appendix_b_through_c_abi(ctx, golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "survey_appendix.json")))["appendix_b"]
    for row in rows:
        R = hj.generate_data(row["dist"], row["rSize"], row["rSize"], row["window"])
        S = R.copy() if row["dist"] == "random" else hj.generate_data("sorted", row["rSize"])
        for algo, key in (("nocc", "outputSumNocc"), ("atomic", "outputSumAtomic")):
            got = ctx.run(algo, R, S)
            assert (got["conflicts"], got["totalMatches"], got["inputSum"]) == (
                row["conflicts"], row["totalMatches"], row["inputSum"]), (row, got)
            if key in row:
                assert got["outputSum"] == row[key], (row, algo, got["outputSum"])


@pytest.mark.parametrize("probe_length", [1, 2, 3, 5, 8])
def test_other_probe_lengths(ctx, probe_length):
    n = 1 << 14
    R = oracle.generate_data("uniform", n, n, 16)
    S = oracle.generate_data("sorted", n)
    want = oracle.build_probe_seq(R, S, probe_length, want_table=True)
    for variant in (1, 2, 3, 4):
        got = ctx.run("atomic", R, S, probeLength=probe_length, buildVariant=variant)
        check_oa(got, want)
        assert np.array_equal(ctx.export_table(2 * n), want["table"])
        if variant == 4:
            assert got["buildVariant"] == 4 and got["compactFallback"] == 0   # near-sorted input: the compact table holds at any budget


def test_auto_variant_follows_locality(ctx):
    """buildVariant 0 samples R (the pre-round of HTMHashBuild.hpp:100-154): near-sorted input takes
    the LDS-window kernel, a random permutation the global-atomic one; results equal either way."""
    n = 1 << 20
    S = oracle.generate_data("sorted", n)
    for dist, window, expect in (("uniform", 16, (3,)), ("sorted", 16, (4,)), ("local_shuffle", 16, (4,)), ("local_shuffle", 64, (3, 4)),
                                 ("local_shuffle", 1024, (2,)), ("shuffle", 16, (1,)), ("random", 16, (1,))):
        R = oracle.generate_data(dist, n, n, window)
        Sx = oracle.relS_for(dist, R)
        got = ctx.run("atomic", R, Sx)
        # rings: the compact ones (4) on (nearly) unique keys where they hold, the classic ones (3) where the sample shows
        # many duplicate keys (`uniform`: their retry rounds bound the build either way) or the device handed over
        assert got["buildVariant"] in expect, (dist, got["buildVariant"])
        if dist == "uniform":
            assert got["compactFallback"] == 0          # not tried
        check_oa(got, oracle.build_probe_seq(R, Sx, 4))


def test_a_context_follows_its_relation_from_step_to_step():
    """hj_build_dev with buildVariant 0 on a reserved context enqueues only the kernels of the variant the PREVIOUS build's
    sample preferred (no global-atomic safety net behind the LDS builds: they are correct on any input). A relation that
    changes its locality class between two builds therefore takes one step on whatever was enqueued -- the rings with
    everything deferred, or global atomics on sorted keys -- and the next step follows the sample. Every step's table is
    the sequential one, slot for slot; buildVariant shows the road taken."""
    n = 1 << 18
    S = np.arange(1, n + 1, dtype=np.uint64)
    rel = {k: oracle.generate_data(k[0], n, n, k[1]) for k in (("sorted", 16), ("shuffle", 16), ("local_shuffle", 1024), ("uniform", 16))}
    want = {k: oracle.build_probe_seq(R, S, 4, want_table=True) for k, R in rel.items()}
    steps = [(("sorted", 16), 4), (("sorted", 16), 4),                       # first build: everything enqueued
             (("shuffle", 16), 3), (("shuffle", 16), 1),                     # locality lost: the classic rings defer it all, then global atomics
             (("sorted", 16), 1), (("sorted", 16), 4),                       # and back: one step on global atomics
             (("local_shuffle", 1024), 3), (("local_shuffle", 1024), 2),     # looser: the rings (behind the compact ones) once, then the window
             (("uniform", 16), 2), (("uniform", 16), 3),                     # duplicate-heavy and tight: the window once, then the classic rings
             (("sorted", 16), 3), (("sorted", 16), 4)]
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); dS = c.dev_alloc(n * 8)
        c.copy_h2d(dS, S)
        c.reserve("atomic", n, n)
        for key, variant in steps:
            c.copy_h2d(dR, rel[key])
            c.build(dR, n); c.probe(dS, n); c.checksums()
            got = c.fetch()
            assert got["buildVariant"] == variant, (key, got["buildVariant"], variant)
            check_oa(got, want[key])
            assert np.array_equal(c.export_table(2 * n), want[key]["table"]), key
        c.dev_free(dR); c.dev_free(dS)


def test_algo_auto_switches_between_table_and_radix_join(ctx):
    """HJ_ALGO_AUTO (the reference's adaptive idea, HTMHashBuild.hpp:98-154): inputs with locality take the
    no-partition path and give the open-addressing result bit for bit, inputs without it take the radix join and give
    PRJ's; on unique keys both agree with each other (totalMatches = |R|)."""
    n = 1 << 20
    for dist, window, expect in (("uniform", 16, "atomic"), ("local_shuffle", 1024, "atomic"), ("sorted", 16, "atomic"),
                                 ("shuffle", 16, "prj"), ("local_shuffle", 1 << 19, "prj")):
        R = oracle.generate_data(dist, n, n, window)
        S = oracle.relS_for(dist, R)
        got = ctx.run("auto", R, S)
        assert got["algoUsed"] == expect, (dist, window, got["algoUsed"])
        if expect == "atomic":
            assert got["buildVariant"] == (2 if window == 1024 else 3 if dist == "uniform" else 4)
            check_oa(got, oracle.build_probe_seq(R, S, 4))
        else:
            want = oracle.prj_join(R, S, got["radixBits"])
            assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"])
            assert got["totalMatches"] == n
    # a relation size the table join does not take (not a power of two): auto is the radix join
    R = oracle.generate_data("sorted", n)[: n - 12345]
    S = oracle.generate_data("sorted", n)
    got = ctx.run("auto", R, S)
    assert got["algoUsed"] == "prj" and got["totalMatches"] == R.size
    # split API: hj_reserve("auto") + hj_join_dev on device pointers, R only
    R = oracle.generate_data("shuffle", n)
    with hj.HashJoinContext(0) as c2:
        dR = c2.dev_alloc(n * 8)
        c2.copy_h2d(dR, R)
        c2.reserve("auto", n, 0)
        c2.join(dR, n, 0, 0)
        got = c2.fetch()
        assert got["algoUsed"] == "prj" and got["totalMatches"] == 0
        assert got["prjChecksum"] == oracle.prj_join(R, None, got["radixBits"])["checksum"]


def test_unaligned_device_pointers(ctx):
    """R and S handed over at an odd tuple offset (8-byte, not 16-byte aligned), through the
    split device-pointer API (hj_reserve / hj_build_dev / hj_probe_dev / hj_fetch_result)."""
    n = 1 << 16
    R = oracle.generate_data("uniform", n, n, 16)
    S = oracle.generate_data("sorted", n)
    want = oracle.build_probe_seq(R, S, 4, want_table=True)
    for variant in (1, 2, 3, 4):
        with hj.HashJoinContext(0) as c2:
            dR = c2.dev_alloc((n + 2) * 8)
            dS = c2.dev_alloc((n + 2) * 8)
            c2.copy_h2d(dR + 8, R)
            c2.copy_h2d(dS + 8, S)
            c2.reserve("atomic", n, n, buildVariant=variant)
            c2.build(dR + 8, n)
            c2.probe(dS + 8, n)
            c2.checksums()
            got = c2.fetch()
            check_oa(got, want)
            assert got["buildVariant"] == variant
            assert np.array_equal(c2.export_table(2 * n), want["table"])
            c2.dev_free(dR)
            c2.dev_free(dS)


def test_heavy_duplicates_and_tiny_sizes(ctx):
    rng = np.random.default_rng(7)
    for n, hi in ((1, 2), (2, 3), (4, 3), (64, 5), (1 << 12, 17), (1 << 15, 1 << 10)):
        R = rng.integers(1, hi, size=n, dtype=np.uint64)
        S = rng.integers(1, hi + 3, size=3 * n + 1, dtype=np.uint64)   # |S| != |R|, odd length
        want = oracle.build_probe_seq(R, S, 4, want_table=True)
        for variant in (1, 2, 3, 4):
            got = ctx.run("atomic", R, S, buildVariant=variant)
            check_oa(got, want)
            assert np.array_equal(ctx.export_table(2 * n), want["table"])


def test_wraparound_at_table_end(ctx):
    n = 1 << 13
    # keys whose home slots are the last slots of the 2n-slot table: probing must wrap (:52)
    R = np.full(n, 2 * n - 1, dtype=np.uint64)
    R[::3] = 2 * n - 2
    S = np.array([2 * n - 1, 2 * n - 2, 1, 2, 4 * n - 1], dtype=np.uint64)
    want = oracle.build_probe_seq(R, S, 4, want_table=True)
    for variant in (1, 2, 3, 4):
        got = ctx.run("atomic", R, S, buildVariant=variant)
        check_oa(got, want)
        assert np.array_equal(ctx.export_table(2 * n), want["table"])
        if variant == 4:
            assert got["buildVariant"] == 3 and got["compactFallback"] & 1   # walks wrap around the table end: the classic rings redo it


def test_build_only_and_empty_probe(ctx):
    n = 1 << 12
    R = oracle.generate_data("uniform", n, n, 16)
    want = oracle.build_probe_seq(R, None, 4)
    got = ctx.run("atomic", R, None)
    check_oa(got, want)
    assert got["totalMatches"] == 0 and got["sSize"] == 0


def test_rejects_non_datagen_tuples(ctx):
    R = np.arange(1, 1025, dtype=np.uint64)
    bad = R.copy(); bad[17] |= np.uint64(1) << np.uint64(40)          # payload bits set
    with pytest.raises(hj.HashJoinError) as e:
        ctx.run("atomic", bad, R)
    assert e.value.status == _lib.HJ_ERR_KEY_RANGE
    zero = R.copy(); zero[5] = 0                                       # 0 is the empty marker (:48)
    with pytest.raises(hj.HashJoinError):
        ctx.run("atomic", zero, R)
    with pytest.raises(hj.HashJoinError) as e:                         # rSize must be a power of two
        ctx.run("atomic", R[:1000], R)
    assert e.value.status == _lib.HJ_ERR_INVALID


def test_operator_mirrors(ctx):
    n = 1 << 16
    R = hj.generate_data("local_shuffle", n, n, 1024)
    S = hj.generate_data("sorted", n)
    tri = n * (n + 1) // 2
    j = hj.NoCCHashBuild(R, n, S, n, 2, 64, 4)
    assert (j["conflicts"], j["totalMatches"], j["inputSum"], j["outputSum"]) == (0, n, tri, tri - n)
    j = hj.AtomicHashBuild(R, n, S, n, 2, 64, 4)
    assert (j["conflicts"], j["totalMatches"], j["inputSum"], j["outputSum"]) == (0, n, tri, tri)
    j = hj.HTMHashBuild(R, n, S, n, 16, 2, 64, 4)
    assert list(j)[:12] == ["algo", "rSize", "transactionSize", "probeLength", "hashBuildTimeInMicroseconds", "firstRoundTime",
                            "firstRoundFailureFraction", "conflictCount", "failedTransactions", "failedTransactionPercentage",
                            "totalFailedPercentage", "totalMatches"]                 # HTMHashBuild.hpp:417-452
    assert (j["transactionSize"], j["conflictCount"], j["totalMatches"], j["inputSum"], j["outputSum"]) == (16, 0, n, tri, tri)
    j = hj.PRO(R, S)
    assert j["matches"] == n


# ---- the bucketised table of --algo htm ---------------------------------------------------------
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("dist,window", [("uniform", 16), ("random", 16), ("sorted", 16), ("shuffle", 16),
                                         ("local_shuffle", 1024), ("local_shuffle", 65536)])
@pytest.mark.parametrize("n", [1 << 10, 1 << 16, 1 << 20])
def test_htm_bucket_table_matches_sequential_oracle(ctx, dist, window, n, variant):
    """HJ_ALGO_HTM against the sequential restatement of HTMHashBuild.hpp (oracle.htm_build_probe_seq): every counter,
    the primary buckets tuple for tuple (a bucket holds its three lowest-indexed tuples in input order), and every
    overflow chain in walk order (head = newest overflow bucket). buildVariant 3 = the LDS rings, 2 = the workgroup
    window (looser locality), 1 = global atomics, 0 = whichever the locality pre-round picks: same table."""
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.relS_for(dist, R)
    want = oracle.htm_build_probe_seq(R, S, want_buckets=True)
    got = ctx.run("htm", R, S, buildVariant=variant)
    assert got["algoUsed"] == "htm" and got["htmBuckets"] == want["numBuckets"]
    assert (got["conflicts"], got["conflictSum"], got["totalMatches"], got["inputSum"], got["tableSumFull"],
            got["htmOverflowBuckets"], got["htmOverflowSum"], got["outputSum"]) == (
        want["conflictCount"], want["conflictSum"], want["totalMatches"], want["inputSum"], want["bucketSum"],
        want["overflowBuckets"], want["overflowSum"], want["outputSum"])
    assert got["totalMatches"] == oracle.true_cardinality(R, S)                 # every tuple is stored once: a correct join
    if variant:
        assert got["buildVariant"] == _variant_that_runs(variant, 4 * want["numBuckets"])
    elif n >= 1 << 16:                         # the pre-round with the table's own hash: rings / window / global atomics
        assert got["buildVariant"] == {("uniform", 16): 3, ("sorted", 16): 3, ("local_shuffle", 1024): 2}.get((dist, window), 1)
    if got["buildVariant"] == 3 and dist in ("uniform", "sorted"):
        assert got["compactFallback"] & 0x100 == 0     # near-sorted keys behind the rings: the chains were built in LDS (hj_htm.hip)
    if got["buildVariant"] != 3:
        assert got["compactFallback"] == 0
    buckets, overflows = ctx.export_buckets(want["numBuckets"])
    assert np.array_equal(buckets["tuples"], want["buckets"]["tuples"]) and np.array_equal(buckets["count"], want["buckets"]["count"])
    assert np.array_equal(buckets["nextIndex"] != 0, want["buckets"]["nextIndex"] != 0)
    assert overflows.size == want["overflows"].size
    if n <= 1 << 16:                                                            # chains, logically (physical indices differ by design)
        a, ao = oracle.htm_chains(buckets, overflows)
        b, bo = oracle.htm_chains(want["buckets"], want["overflows"])
        assert np.array_equal(ao, bo) and np.array_equal(a, b)
    else:                                                                       # same multiset of overflow buckets
        key = lambda o: np.sort(o[1:].view(np.uint64).reshape(-1, 4)[:, :3].sum(axis=1))          # noqa: E731
        assert np.array_equal(key(overflows), key(want["overflows"]))


def test_htm_heavy_duplicates_odd_sizes_and_split_api(ctx):
    """Long chains (few keys), sizes that are not powers of two (the bucket hash does not need one), R-only builds, an
    unaligned device pointer, and a probe side with keys the table never saw."""
    rng = np.random.default_rng(5)
    for n, hi in ((1, 2), (7, 3), (1000, 5), (4097, 50), (100_000, 1000), (1 << 15, 1 << 14)):
        R = rng.integers(1, hi, size=n, dtype=np.uint64)
        S = rng.integers(1, hi + 5, size=2 * n + 3, dtype=np.uint64)
        want = oracle.htm_build_probe_seq(R, S, want_buckets=True)
        for variant in (1, 2, 3):
            with hj.HashJoinContext(0) as c:
                dR = c.dev_alloc(n * 8 + 16); dS = c.dev_alloc(S.size * 8 + 16)
                c.copy_h2d(dR + 8, R); c.copy_h2d(dS + 8, S)
                c.reserve("htm", n, S.size, buildVariant=variant)
                c.build(dR + 8, n)
                c.probe(dS + 8, S.size)
                c.checksums()
                got = c.fetch()
                assert (got["conflicts"], got["totalMatches"], got["inputSum"], got["outputSum"], got["htmOverflowBuckets"]) == (
                    want["conflictCount"], want["totalMatches"], want["inputSum"], want["inputSum"], want["overflowBuckets"]), (n, hi, variant)
                buckets, overflows = c.export_buckets(want["numBuckets"])
                a, ao = oracle.htm_chains(buckets, overflows)
                b, bo = oracle.htm_chains(want["buckets"], want["overflows"])
                assert np.array_equal(ao, bo) and np.array_equal(a, b), (n, hi, variant)
                c.dev_free(dR); c.dev_free(dS)


def test_htm_log_pins_at_2p27(ctx):
    """The `htm` lines of the reference's logs (experiments/new_backup/probe_log*, tests/golden/reference_logs.json):
    local_shuffle at 2^27 -- conflictCount 0, totalMatches = rSize, inputSum = outputSum = 9007199321849856."""
    n = 1 << 27
    R = hj.generate_data("local_shuffle", n, n, 1024)
    S = hj.generate_data("sorted", n)
    got = ctx.run("htm", R, S)
    # (a shuffle window of 1024 is more than the 8 KiB rings hold at 4 slots per 3 keys: the pre-round takes the workgroup window)
    assert (got["conflicts"], got["totalMatches"], got["inputSum"], got["outputSum"], got["htmOverflowBuckets"], got["buildVariant"]) == (
        0, n, 9007199321849856, 9007199321849856, 0, 2)
    R = hj.generate_data("uniform", n, n, 16)                                   # duplicates at full size: the order-independent count
    got = ctx.run("htm", R, S)
    assert got["buildVariant"] == 3                                             # the reference's default window: the LDS rings
    per = np.bincount(((R // np.uint64(3)) & np.uint64(got["htmBuckets"] - 1)).astype(np.int64), minlength=got["htmBuckets"])
    assert got["conflicts"] == int(np.maximum(per - 3, 0).sum())
    assert got["htmOverflowBuckets"] == int(((np.maximum(per - 3, 0) + 2) // 3).sum())
    assert got["outputSum"] == got["inputSum"] == 9006807263251667
    assert got["totalMatches"] == int(per.sum()) == n                            # S = 1..N, keys in [1, N]: every R tuple matches once


def test_mc_workloads_true_cardinality_on_both_join_paths(ctx, golden_dir):
    """The workloads of mc/src/main.c:343-408 (pk x fk with |S| != |R|, --non-unique, --skew, --local-shuffle-range) from
    hj_generate_relation, joined on the GPU: the radix join and the bucketised table join (the correct-join mode of the
    table path: dropped tuples live in overflow chains the probe also walks) must both give the TRUE cardinality the
    reference's own NPO printed for the same command line (tests/golden/mc_workloads.json)."""
    rows = json.load(open(os.path.join(golden_dir, "mc_workloads.json")))["rows"]
    for row in rows:
        skew = next((float(f.split("=")[1]) for f in row["flags"] if f.startswith("--skew=")), 0.0)
        win = next((int(f.split("=")[1]) for f in row["flags"] if f.startswith("--local-shuffle-range=")), 0)
        R = hj.generate_relation(row["rKind"], row["rSize"], row["rSize"], win, 0.0, row["rSeed"])
        S = hj.generate_relation(row["sKind"], row["sSize"], row["rSize"], 0, skew, row["sSeed"])
        got = ctx.run("prj", R, S)                              # takes key 0 (mc's nonunique keys are 0 .. maxid-1)
        assert got["totalMatches"] == row["results"], ("prj", row)
        if "--non-unique" in row["flags"]:                      # key 0 is the table paths' empty marker: shift the key domain
            with pytest.raises(hj.HashJoinError) as e:
                ctx.run("htm", R, S)
            assert e.value.status == _lib.HJ_ERR_KEY_RANGE
            R, S = R + np.uint64(1), S + np.uint64(1)
        for variant in (1, 2, 3):
            got = ctx.run("htm", R, S, buildVariant=variant)
            assert got["totalMatches"] == row["results"], ("htm", variant, row)
            assert got["outputSum"] == got["inputSum"] == int(R.sum())


# ---- PRJ ---------------------------------------------------------------------
@pytest.mark.parametrize("dist,window", [("uniform", 16), ("random", 16), ("shuffle", 16), ("local_shuffle", 1024)])
@pytest.mark.parametrize("n,bits", [(1 << 10, 4), (1 << 16, 14), (1 << 20, 14), (1 << 20, 9), (1 << 20, 16)])
def test_prj_matches_oracle(ctx, dist, window, n, bits):
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.relS_for(dist, R)
    want = oracle.prj_join(R, S, bits)
    got = ctx.run("prj", R, S, radixBits=bits)
    assert got["totalMatches"] == want["matches"]
    assert got["prjChecksum"] == want["checksum"]
    assert got["radixBits"] == bits


def test_prj_reference_checksum_pin(ctx, golden_dir):
    """PRO "Results" printed by oracle/_ref/mchashjoins (and the closed form behind
    experiments/new_backup/motivation_log1:8) through the C ABI."""
    rows = json.load(open(os.path.join(golden_dir, "mc_ref.json")))["rows"]
    for row in rows:
        n = row["rSize"]
        R = hj.generate_data("shuffle", n)
        S = hj.generate_data("sorted", n)
        got = ctx.run("prj", R, S, radixBits=14)
        if row["algo"] == "PRO":
            assert got["prjChecksum"] == row["results"], (n, got["prjChecksum"])
        else:
            assert got["totalMatches"] == row["results"]


def test_prj_ragged_skewed_and_r_only(ctx):
    rng = np.random.default_rng(3)
    n = 1 << 16
    R = oracle.generate_data("shuffle", n)
    S = oracle.generate_zipf(3 * n + 7, n, 0.9, 1)                    # |S| != |R|, skewed, odd
    assert ctx.run("prj", R, S, radixBits=12)["totalMatches"] == S.size == oracle.true_cardinality(R, S)
    # all tuples in ONE partition, more than one LDS block of R: the multi-block path
    R1 = (rng.integers(1, 1 << 14, size=60000, dtype=np.uint64) << np.uint64(8)) | np.uint64(5)
    S1 = (rng.integers(1, 1 << 14, size=50001, dtype=np.uint64) << np.uint64(8)) | np.uint64(5)
    want = oracle.prj_join(R1, S1, 8)
    got = ctx.run("prj", R1, S1, radixBits=8)
    assert got["totalMatches"] == want["matches"] == oracle.true_cardinality(R1, S1)
    assert got["prjChecksum"] == want["checksum"]
    # radixBits 16 = the direct-addressed counter join: heavy duplicates inside one partition (counters well above 1),
    # a partition of exactly 65535 R tuples (the counters' limit) and one of 65536 (falls back to the hash table)
    for m in (65535, 65536, 40000):
        R2 = (rng.integers(0, 300, size=m, dtype=np.uint64) << np.uint64(16)) | np.uint64(0x1234)
        S2 = (rng.integers(0, 400, size=m + 11, dtype=np.uint64) << np.uint64(16)) | np.uint64(0x1234)
        R2 = np.concatenate([R2, oracle.generate_data("shuffle", 1 << 16)])          # plus ordinary partitions
        want = oracle.prj_join(R2, S2, 16)
        got = ctx.run("prj", R2, S2, radixBits=16)
        assert got["totalMatches"] == want["matches"] == oracle.true_cardinality(R2, S2), m
        assert got["prjChecksum"] == want["checksum"], m
    # R-side only (what the fork's PRO actually runs): checksum, no matches
    got = ctx.run("prj", R, None, radixBits=14)
    assert got["totalMatches"] == 0 and got["prjChecksum"] == oracle.prj_join(R, None, 14)["checksum"]


@pytest.mark.parametrize("dist,window,nS", [("uniform", 16, None), ("shuffle", 16, 20_000_003), ("local_shuffle", 1024, None),
                                            ("sorted", 16, (1 << 24) + 4099)])
@pytest.mark.parametrize("n,bits", [(1 << 24, 14), (1 << 24, 16), (19_999_999, 15)])
def test_prj_histogram_free_passes_match_oracle(ctx, dist, window, nS, n, bits):
    """prjMode 2: the histogram-free passes (fragments + counts instead of dense partitions) at sizes the oracle joins in
    seconds; same matches and the same PRO checksum as the exact passes, and the path really was the one taken."""
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.relS_for(dist, R) if nS is None else hj.generate_data("uniform", nS, n, 16)
    want = oracle.prj_join(R, S, bits)
    got = ctx.run("prj", R, S, radixBits=bits, prjMode=2)
    # DataGen's `uniform` is (rand() & (distinct - 1)) + 1 (DataGen.hpp:41): with a `distinct` that is no power of two the
    # mask has holes, whole residue classes of the low bits never occur, and the fragments of the others overflow
    masked = (n & (n - 1)) != 0 and (dist == "uniform" or nS is not None)
    assert got["prjPath"] == (2 if masked else 1), got
    assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"])
    exact = ctx.run("prj", R, S, radixBits=bits, prjMode=1)
    assert exact["prjPath"] == 0
    assert (exact["totalMatches"], exact["prjChecksum"]) == (want["matches"], want["checksum"])


@pytest.mark.parametrize("skewed", ["R", "S", "S-pass2"])
def test_prj_histogram_free_passes_fall_back_on_skew(ctx, skewed):
    """A fragment that would overflow (low key bits far from uniform) sets the fallback word on the device; the exact
    passes enqueued behind redo the join. Skew in R's pass 1, in S's pass 1 (after R went through), in a pass 2."""
    n, bits = 1 << 24, 14
    rng = np.random.default_rng(5)
    R = oracle.generate_data("shuffle", n)
    S = hj.generate_data("uniform", n, n, 16)
    if skewed == "R":
        R = (rng.integers(1, 1 << 17, size=n, dtype=np.uint64) << np.uint64(7)) | np.uint64(3)       # one pass-1 bin
    elif skewed == "S":
        S = (rng.integers(1, 1 << 17, size=n, dtype=np.uint64) << np.uint64(7)) | np.uint64(3)
    else:
        S = (rng.integers(1, 1 << 10, size=n, dtype=np.uint64) << np.uint64(14)) | rng.integers(0, 128, size=n, dtype=np.uint64)
        S[S == 0] = 1                                                                               # pass-2 bin 0 only
    want = oracle.prj_join(R, S, bits)
    got = ctx.run("prj", R, S, radixBits=bits, prjMode=2)
    assert got["prjPath"] == 2, got
    assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"])
    # the same context afterwards, on well-behaved input: the word is reset with the counters
    R2 = oracle.generate_data("shuffle", n)
    want = oracle.prj_join(R2, S, bits)
    got = ctx.run("prj", R2, S, radixBits=bits, prjMode=2)
    assert got["prjPath"] == (1 if skewed == "R" else 2)
    assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"])


def test_prj_histogram_free_passes_r_only_small_s_and_auto(ctx):
    """Who takes the histogram-free passes: R alone (the fork's PRO: no probe) does; a large R with a small S does not
    (both relations must qualify, else everything goes through the exact passes); AUTO's radix join does."""
    n = 1 << 25
    R = oracle.generate_data("shuffle", n)
    got = ctx.run("prj", R, None, radixBits=14)
    assert got["prjPath"] == 1 and got["totalMatches"] == 0
    assert got["prjChecksum"] == oracle.prj_join(R, None, 14)["checksum"]
    S = oracle.generate_data("shuffle", 100_000)
    want = oracle.prj_join(R, S, 14)
    got = ctx.run("prj", R, S, radixBits=14)
    assert got["prjPath"] == 0
    assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"])
    S = hj.generate_data("sorted", n)
    want = oracle.prj_join(R, S, 14)
    got = ctx.run("auto", R, S, radixBits=14)                   # a shuffled R has no locality: the radix join
    assert got["algoUsed"] == "prj" and got["prjPath"] == 1
    assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"])


@pytest.mark.parametrize("nR,nS", [(40_000_000, 1 << 25), (1 << 25, 40_000_000)])
def test_prj_unequal_sizes_across_the_chunk_length_step(nR, nS):
    """|R| and |S| on opposite sides of the size where the partitioning chunk length doubles: the smaller relation
    then has MORE chunks, and the histogram workspace must have been sized for it (round-1 ADVICE, high)."""
    R = np.arange(1, nR + 1, dtype=np.uint64)
    rng = np.random.default_rng(11)
    rng.shuffle(R[: 1 << 22])                                  # some disorder without a full 40M-element shuffle
    S = (rng.integers(0, nR + nR // 3, size=nS, dtype=np.uint64) + np.uint64(1))
    want = oracle.prj_join(R, S, 14)
    with hj.HashJoinContext(0) as c:
        # AUTO with a non-power-of-two |R| is the radix join too
        for algo in ("prj", "auto") if nR & (nR - 1) else ("prj",):
            got = c.run(algo, R, S, radixBits=14)
            assert got["algoUsed"] == "prj" and got["prjPath"] == 1     # both >= 2^25: the histogram-free passes
            assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"]), algo


def test_empty_s_with_a_pointer_and_index_limit(ctx):
    """hj_prj_join_dev / hj_join_dev with dS non-null but sSize == 0 (hj_probe_dev allows an empty S): treated as
    R-only instead of reading partS[0xFFFFFFFF]; and index 0xFFFFFFFF is refused, since (index << 32 | key) of
    index = key = 0xFFFFFFFF would be the empty pattern (round-1 ADVICE, low)."""
    n = 1 << 16
    R = oracle.generate_data("shuffle", n)
    want = oracle.prj_join(R, None, 14)
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); dS = c.dev_alloc(64)
        c.copy_h2d(dR, R)
        for algo in ("prj", "auto"):
            c.reserve(algo, n, 0, radixBits=14)
            c.join(dR, n, dS, 0)
            got = c.fetch()
            assert got["algoUsed"] == "prj" and got["totalMatches"] == 0 and got["prjChecksum"] == want["checksum"]
        c.reserve("atomic", n, 0)
        c.build(dR, n, idx_base=(1 << 32) - 1 - n)             # last index used = 2^32 - 2: fine
        assert c.fetch()["conflicts"] == 0
        with pytest.raises(hj.HashJoinError) as e:
            c.build(dR, n, idx_base=(1 << 32) - n)             # would use index 2^32 - 1
        assert e.value.status == _lib.HJ_ERR_INVALID
        c.dev_free(dR); c.dev_free(dS)


# ---- full-size, size-independent properties -------------------------------------
def test_config2_size_properties(ctx):
    """BASELINE configs[1] size (2^27): invariants that hold for any correct run, plus the
    reference's logged inputSum for `uniform` (experiments/overflow_log1)."""
    n = 1 << 27
    R = hj.generate_data("uniform", n, n, 16)
    S = hj.generate_data("sorted", n)
    got = ctx.run("atomic", R, S)
    assert got["inputSum"] == 9006807263251667
    assert got["totalMatches"] + got["conflicts"] == n
    assert got["tableSumFull"] + got["conflictSum"] == got["inputSum"]
    want = oracle.build_probe_seq(R, S, 4)
    assert (got["conflicts"], got["totalMatches"]) == (want["conflicts"], want["totalMatches"])
    # the bare-key entry points (what a radix shard runs after the exchange) on the same input: same table, same counts
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 4 + 16); dS = c.dev_alloc(n * 4 + 16)
        c.copy_h2d(dR + 4, R.astype(np.uint32)); c.copy_h2d(dS + 8, S.astype(np.uint32))
        c.reserve("atomic", n, n)
        c.build_keys(dR + 4, n, 0, 2 * n)
        c.probe_keys(dS + 8, n)
        c.checksums()
        k = c.fetch()
        assert k["buildVariant"] == 3 and k["compactFallback"] == 0          # duplicate keys: the classic rings
        for f in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"):
            assert k[f] == got[f], f
        c.dev_free(dR); c.dev_free(dS)
    del R
    R = hj.generate_data("local_shuffle", n, n, 1024)
    for algo, outsum in (("nocc", 9007199187632128), ("atomic", 9007199321849856)):   # probe_log1
        got = ctx.run(algo, R, S)
        assert (got["conflicts"], got["totalMatches"], got["inputSum"], got["outputSum"]) == (
            0, n, 9007199321849856, outsum)
    got = ctx.run("prj", R, S, radixBits=14)
    assert got["totalMatches"] == n and got["prjChecksum"] == 549688705024            # motivation_log1:8


def test_maximum_relation_size():
    """rSize = 2^31, the largest power of two the reference's uint32_t sizes hold (table of 2^32 slots = 32 GiB, slot
    numbers use all 32 bits): local_shuffle W=1024 against S = sorted. Unique keys: no conflicts, every probe matches,
    the closed-form sums; then the radix join of the same relations. ~160 GiB of HBM."""
    n = 1 << 31
    R = hj.generate_data("local_shuffle", n, n, 1024)
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); c.copy_h2d(dR, R)
        del R
        S = hj.generate_data("sorted", n)
        dS = c.dev_alloc(n * 8); c.copy_h2d(dS, S)
        del S
        c.reserve("atomic", n, n)
        c.build(dR, n); c.probe(dS, n)
        c.checksums()
        r = c.fetch()
        with hj.HashJoinContext(0) as c2:          # the radix join on the same device buffers (32-bit element indices)
            c2.reserve("prj", n, n)
            c2.prj_join(dR, n, dS, n)
            p = c2.fetch()
        assert p["totalMatches"] == n and p["radixBits"] == 16
        c.dev_free(dR); c.dev_free(dS)
    tri = n * (n + 1) // 2
    assert (r["conflicts"], r["totalMatches"], r["inputSum"], r["tableSumFull"], r["buildVariant"]) == (0, n, tri, tri, 2)
    assert r["tableSumHalf"] == tri - n            # the nocc quirk: slots [0, rSize) miss key N, which sits in slot N


def test_maximum_relation_size_on_the_wavefront_rings():
    """The same size through build variant 3 (k_build_wave: slot numbers up to 2^32 - 1, 2^25 granules, 2^19-tuple chunks)
    and the bucketised table (2^30 buckets = 2^32 slots): `sorted` R (the pre-round picks the rings), S = R."""
    n = 1 << 31
    R = hj.generate_data("sorted", n)
    tri = n * (n + 1) // 2
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); c.copy_h2d(dR, R)
        del R
        c.reserve("atomic", n, n)
        c.build(dR, n); c.probe(dR, n)
        c.checksums()
        r = c.fetch()
        assert (r["conflicts"], r["totalMatches"], r["inputSum"], r["tableSumFull"], r["buildVariant"], r["buildDeferred"]) == (
            0, n, tri, tri, 4, 0)
    with hj.HashJoinContext(0) as c:
        c.reserve("htm", n, n)
        c.build(dR, n); c.probe(dR, n)
        c.checksums()
        h = c.fetch()
        assert (h["conflicts"], h["totalMatches"], h["inputSum"], h["outputSum"], h["htmBuckets"], h["buildVariant"]) == (
            0, n, tri, tri, 1 << 30, 3)
    with hj.HashJoinContext(0) as c:
        c.dev_free(dR)


# ---- radix-sharded path: every GPU kernel of htm_hashjoin_amd/sharded.py on one device ------------
@pytest.mark.parametrize("G", [2, 8, 64])
@pytest.mark.parametrize("dist,window", [("uniform", 16), ("local_shuffle", 1024), ("random", 16)])
@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_sharded_kernels_on_one_gpu(ctx, G, dist, window, variant):
    """G ranks emulated in turn on one GPU: shard histogram + STABLE scatter to 32-bit keys per source piece, the
    all-to-all done on the host (pieces laid out in source-rank order), then hj_build_keys_dev / hj_probe_keys_dev per
    destination shard. The scatter must equal numpy's stable sort by destination element for element; totals and every
    shard's table must equal the sequential oracle on the shard's tuples in global input order."""
    if G == 64 and (variant != 2 or dist == "random"):
        pytest.skip("G = 64 is covered once per build kernel input")
    n = 1 << 16
    n_local = n // G
    n_piece = n_local + 3 if G == 2 else n_local          # ragged pieces too (the last one is shorter)
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.relS_for(dist, R)
    shift = G.bit_length() - 1
    inbox_r = [[] for _ in range(G)]
    inbox_s = [[] for _ in range(G)]
    with hj.HashJoinContext(0) as c:
        d_in = c.dev_alloc((n_piece + 1) * 8); d_out = c.dev_alloc((n_piece + 4) * 4); d_cnt = c.dev_alloc(G * 8)
        for src in range(G):
            for rel, inbox in ((R, inbox_r), (S, inbox_s)):
                piece = rel[src * n_piece:(src + 1) * n_piece]
                m = piece.size
                c.copy_h2d(d_in, piece)
                c.shard_histogram(d_in, m, G, d_cnt)
                c.shard_scatter(d_in, m, G, d_cnt, d_out)
                cnt = np.empty(G, dtype=np.uint64); c.copy_d2h(cnt, d_cnt)
                out = np.empty(m, dtype=np.uint32); c.copy_d2h(out, d_out)
                dest = (piece & np.uint64(G - 1)).astype(np.int64)
                assert np.array_equal(cnt, np.bincount(dest, minlength=G).astype(np.uint64))
                assert np.array_equal(out, piece[np.argsort(dest, kind="stable")].astype(np.uint32))     # stable, element for element
                off = 0
                for g in range(G):
                    inbox[g].append(out[off:off + int(cnt[g])]); off += int(cnt[g])
        for p in (d_in, d_out, d_cnt):
            c.dev_free(p)
    tot = {k: 0 for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum")}
    table_size = 2 * n_local
    for g in range(G):
        got_r = np.concatenate(inbox_r[g]); got_s = np.concatenate(inbox_s[g])
        assert np.array_equal(got_r, R[(R & np.uint64(G - 1)) == g].astype(np.uint32))       # global input order
        want = oracle.build_probe_seq_ts(got_r.astype(np.uint64), got_s.astype(np.uint64), table_size, shift, want_table=True)
        with hj.HashJoinContext(0) as c:
            r = table_size // 2
            while r < got_r.size:
                r *= 2
            c.reserve("atomic", r, got_s.size, buildVariant=variant)
            # odd offsets: the key buffers of an exchange start wherever the previous peer's keys ended
            d_r = c.dev_alloc((got_r.size + 8) * 4); d_s = c.dev_alloc((got_s.size + 8) * 4)
            c.copy_h2d(d_r + 4, got_r) if got_r.size else None
            c.copy_h2d(d_s + 12, got_s) if got_s.size else None
            c.build_keys(d_r + 4, got_r.size, shift, table_size)
            c.probe_keys(d_s + 12, got_s.size)
            c.checksums()
            res = c.fetch()
            assert res["buildVariant"] == _variant_that_runs(variant, table_size, res["compactFallback"] == 0)
            if variant == 4 and dist == "uniform" and table_size >= 1024:
                assert res["buildVariant"] == 4, (g, res["compactFallback"], got_r.size)   # near-sorted keys stay near-sorted inside a shard: the compact table holds
            for k in tot:
                assert res[k] == want[k], (g, k)
                tot[k] += res[k]
            assert np.array_equal(c.export_table(table_size), want["table"])
            c.dev_free(d_r); c.dev_free(d_s)
    assert tot == oracle.sharded_reference(R, S, G)


@pytest.mark.parametrize("G,log2n,dist", [(4, 26, "uniform"), (2, 25, "shuffle"), (8, 25, "random")])
def test_shard_split_is_stable_at_size(G, log2n, dist):
    """The stable split at sizes with thousands of chunks and several tiles per chunk: every destination's keys must be
    exactly the input's keys of that destination in input order (a mask select on the host, no sort involved)."""
    n = (1 << log2n) - 5                                   # ragged: the last tile and the last chunk are partial
    R = hj.generate_data(dist, 1 << log2n, 1 << log2n, 16)[:n]
    with hj.HashJoinContext(0) as c:
        d_in = c.dev_alloc(n * 8); d_out = c.dev_alloc(n * 4 + 16); d_cnt = c.dev_alloc(G * 8)
        c.copy_h2d(d_in, R)
        c.shard_histogram(d_in, n, G, d_cnt)
        c.shard_scatter(d_in, n, G, d_cnt, d_out)
        cnt = np.empty(G, dtype=np.uint64); c.copy_d2h(cnt, d_cnt)
        out = np.empty(n, dtype=np.uint32); c.copy_d2h(out, d_out)
        for p in (d_in, d_out, d_cnt):
            c.dev_free(p)
    keys = R.astype(np.uint32)
    dest = keys & np.uint32(G - 1)
    off = 0
    for g in range(G):
        want = keys[dest == g]
        assert int(cnt[g]) == want.size
        assert np.array_equal(out[off:off + want.size], want), g
        off += want.size


@pytest.mark.parametrize("dist,window", [("uniform", 16), ("shuffle", 16)])
def test_range_split_on_one_gpu(dist, window):
    """The HIGH-bit (range) split, G = 4 ranks emulated on one GPU: dest = ((key - 1) >> 14) & 3 for keys 1..2^16,
    stable, and the shards (home slot = key & mask, no shift) equal the sharded reference."""
    G, n = 4, 1 << 16
    n_local = n // G
    mode = 14 | hj.SHARD_ONE_BASED
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.generate_data("sorted", n)
    inbox_r = [[] for _ in range(G)]; inbox_s = [[] for _ in range(G)]
    moved = 0
    with hj.HashJoinContext(0) as c:
        d_in = c.dev_alloc(n_local * 8); d_out = c.dev_alloc(n_local * 4 + 16); d_cnt = c.dev_alloc(G * 8)
        for src in range(G):
            for rel, inbox in ((R, inbox_r), (S, inbox_s)):
                piece = rel[src * n_local:(src + 1) * n_local]
                c.copy_h2d(d_in, piece)
                c.shard_histogram(d_in, n_local, G, d_cnt, mode)
                with pytest.raises(hj.HashJoinError):                     # the scatter must name the histogram's mode
                    c.shard_scatter(d_in, n_local, G, d_cnt, d_out, 0)
                c.shard_scatter(d_in, n_local, G, d_cnt, d_out, mode)
                cnt = np.empty(G, dtype=np.uint64); c.copy_d2h(cnt, d_cnt)
                out = np.empty(n_local, dtype=np.uint32); c.copy_d2h(out, d_out)
                dest = (((piece - np.uint64(1)) >> np.uint64(14)) & np.uint64(G - 1)).astype(np.int64)
                assert np.array_equal(out, piece[np.argsort(dest, kind="stable")].astype(np.uint32))
                moved += int((dest != src).sum())
                off = 0
                for g in range(G):
                    inbox[g].append(out[off:off + int(cnt[g])]); off += int(cnt[g])
        for p in (d_in, d_out, d_cnt):
            c.dev_free(p)
    if dist == "uniform":
        assert moved < n // 100                                           # contiguous pieces of a near-sorted relation stay put
    tot = {k: 0 for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum")}
    for g in range(G):
        got_r = np.concatenate(inbox_r[g]); got_s = np.concatenate(inbox_s[g])
        with hj.HashJoinContext(0) as c:
            r = n_local
            while r < got_r.size:
                r *= 2
            c.reserve("atomic", r, got_s.size)
            d_r = c.dev_alloc(got_r.size * 4 + 16); d_s = c.dev_alloc(got_s.size * 4 + 16)
            c.copy_h2d(d_r, got_r); c.copy_h2d(d_s, got_s)
            c.build_keys(d_r, got_r.size, 0, 2 * n_local)
            c.probe_keys(d_s, got_s.size)
            c.checksums()
            res = c.fetch()
            for k in tot:
                tot[k] += res[k]
            c.dev_free(d_r); c.dev_free(d_s)
    assert tot == oracle.sharded_reference(R, S, G, digit_shift=14, one_based=True)


@pytest.mark.parametrize("world,dist,window,split,expect", [
    (2, "uniform", 16, "low", "low"), (4, "uniform", 16, "auto", "high"), (2, "uniform", 16, "high", "high"),
    (4, "shuffle", 16, "auto", "low"), (2, "sorted", 16, "auto", "in place"), (4, "local_shuffle", 1024, "auto", "high"),
    (4, "uniform", 16, "low", "low a2a")])
def test_sharded_join_product_engine_in_process_ranks(world, dist, window, split, expect):
    """htm_hashjoin_amd/sharded.py end to end with the PRODUCT engine (HipShardEngine) at world 2 and 4 on one GPU: the
    ranks are threads of this process and exchange through tests/loopback_dist.py instead of RCCL. Low-bit split, range
    split, auto, and the join-in-place case; totals must equal the sharded reference of the split that ran."""
    import threading
    assert _TORCH_GPU, "torch could not initialise the GPU"
    from htm_hashjoin_amd.sharded import HipShardEngine, ShardedJoin
    from loopback_dist import Hub, LoopbackDist
    n_local = 1 << 16
    n = n_local * world
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.generate_data("sorted", n)
    hub = Hub(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            eng = HipShardEngine(hj, torch, 0)
            a2a = expect.endswith("a2a")                                  # one all_to_all_single per relation instead
            job = ShardedJoin(eng, torch, LoopbackDist(hub, rank), rank, world, split=split, max_key=n,
                              exchange="a2a" if a2a else "p2p")
            if not a2a:
                job.max_msg_tuples = 5000                                 # several messages per peer
            r_local = torch.from_numpy(R[rank * n_local:(rank + 1) * n_local].view(np.int64).copy()).to("cuda:0")
            s_local = torch.from_numpy(S[rank * n_local:(rank + 1) * n_local].view(np.int64).copy()).to("cuda:0")
            for _ in range(2):                                            # second step: buffers are reused
                job.step(r_local, s_local, 2 * n_local)
            res = job.result()
            res["mode"] = job.mode
            res["form"] = job.last_exchange_form
            out[rank] = res
            eng.close()
        except Exception as e:                                            # noqa: BLE001 -- reported by the main thread
            errs.append((rank, repr(e)))
            hub.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    got = out[0]
    bits = (n - 1).bit_length() - (world.bit_length() - 1)
    if expect.startswith("low"):
        assert got["mode"] == 0 and got["exchange"]["sent_r"] > 0
        assert got["form"] == ("all_to_all_single" if expect.endswith("a2a") else "batch_isend_irecv")
        want = oracle.sharded_reference(R, S, world)
    else:
        assert got["mode"] == (bits | hj.SHARD_ONE_BASED)
        want = oracle.sharded_reference(R, S, world, digit_shift=bits, one_based=True)
        assert ("in place" in got["exchange"]["split"]) == (expect == "in place")
        if expect == "in place":      # the second step took the optimistic route: membership checked by the kernels
            assert "checked inside build and probe" in got["exchange"]["split"]
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])
    assert all(o["conflicts"] == got["conflicts"] for o in out)           # the all-reduce reached every rank


def test_randomised_differential_keys_and_tuples():
    """Seeded random configurations -- table size, home shift, probe length, build variant, duplicate density, arrival
    order, element format (tuples / bare keys), odd buffer offsets -- through the split device API against the
    sequential oracle: whole table and all counters."""
    rng = np.random.default_rng(20241004)
    for case in range(160):
        log2t = int(rng.integers(13, 19))                      # table of 2^13 .. 2^18 slots
        table_size = 1 << log2t
        n = int(rng.integers(table_size // 8, table_size // 2 + 1))
        hshift = int(rng.integers(0, 4))
        probe_len = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8]))
        variant = int(rng.integers(1, 4))
        key32 = bool(rng.integers(0, 2)) or hshift > 0        # tuples have no home shift in the ABI
        span = int(rng.choice([n // 4 + 1, n, 4 * n, 1 << 31]))
        keys = rng.integers(1, span + 1, size=n, dtype=np.uint64)
        order = int(rng.integers(0, 3))
        if order == 0:
            keys.sort()
        elif order == 1:                                       # near-sorted: sorted, then shuffled inside windows of 64
            keys.sort()
            for b in range(0, n - 64, 64):
                rng.shuffle(keys[b:b + 64])
        keys = (keys << np.uint64(hshift)) | np.uint64(rng.integers(0, 1 << hshift)) if hshift else keys
        keys &= np.uint64(0xFFFFFFFF)
        keys[keys == 0] = 1
        S = np.concatenate([keys[rng.integers(0, n, size=n // 2)], rng.integers(1, 1 << 32, size=n // 3, dtype=np.uint64)])
        want = oracle.build_probe_seq_ts(keys, S, table_size, hshift, probe_len, want_table=True)
        with hj.HashJoinContext(0) as c:
            c.reserve("atomic", table_size // 2, S.size, buildVariant=variant, probeLength=probe_len)
            offR, offS = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            if key32:
                dR = c.dev_alloc(n * 4 + 32); dS = c.dev_alloc(S.size * 4 + 32)
                c.copy_h2d(dR + 4 * offR, keys.astype(np.uint32)); c.copy_h2d(dS + 4 * offS, S.astype(np.uint32))
                c.build_keys(dR + 4 * offR, n, hshift, table_size)
                c.probe_keys(dS + 4 * offS, S.size)
            else:
                if n * 2 != table_size:                        # hj_build_dev fixes the table at 2 * rSize, rSize a power of two
                    n = table_size // 2
                    keys = np.resize(keys, n); S = S[: max(1, S.size)]
                    want = oracle.build_probe_seq_ts(keys, S, table_size, 0, probe_len, want_table=True)
                dR = c.dev_alloc(n * 8 + 32); dS = c.dev_alloc(S.size * 8 + 32)
                c.copy_h2d(dR + 8 * (offR & 1), keys); c.copy_h2d(dS + 8 * (offS & 1), S)
                c.build(dR + 8 * (offR & 1), n)
                c.probe(dS + 8 * (offS & 1), S.size)
            c.checksums()
            got = c.fetch()
            tag = (case, log2t, n, hshift, probe_len, variant, key32, span, order)
            for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"):
                assert got[k] == want[k], (k, got[k], want[k], tag)
            assert np.array_equal(c.export_table(table_size), want["table"]), tag
            c.dev_free(dR); c.dev_free(dS)


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_shard_check_counts_foreign_tuples(variant):
    """hj_set_shard_check: builds and probes count the tuples whose destination is another shard, on tuples and on keys,
    in both build kernels; off again afterwards."""
    n, G = 1 << 16, 4
    mode = 14 | hj.SHARD_ONE_BASED
    R = oracle.generate_data("uniform", n, n, 16)
    S = oracle.generate_data("sorted", n)
    dest = lambda k: (((k - np.uint64(1)) >> np.uint64(14)) & np.uint64(G - 1))          # noqa: E731
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); dS = c.dev_alloc(n * 8 + 16); dK = c.dev_alloc(n * 4 + 16)
        c.copy_h2d(dR, R); c.copy_h2d(dS + 8, S); c.copy_h2d(dK + 4, S.astype(np.uint32))
        c.reserve("atomic", n, n, buildVariant=variant)
        for shard in (0, 3):
            c.set_shard_check(G, mode, shard)
            c.build(dR, n); c.probe(dS + 8, n); c.probe_keys(dK + 4, n)
            got = c.fetch()
            assert got["foreignTuples"] == int((dest(R) != shard).sum()) + 2 * int((dest(S) != shard).sum())
        c.set_shard_check(0)
        c.build(dR, n); c.probe(dS + 8, n)
        assert c.fetch()["foreignTuples"] == 0
        with pytest.raises(hj.HashJoinError):
            c.set_shard_check(3, 0, 0)
        for p_ in (dR, dS, dK):
            c.dev_free(p_)


def test_shard_split_flags_payload_bits(ctx):
    """A tuple with payload bits set cannot be told from a valid one once only keys travel: the split sends it as
    key 0 (to shard 0), where the build reports it like hj_build_dev does (HJ_ERR_KEY_RANGE)."""
    n, G = 1 << 13, 4
    R = oracle.generate_data("sorted", n)
    R[1234] |= np.uint64(1) << np.uint64(40)
    with hj.HashJoinContext(0) as c:
        d_in = c.dev_alloc(n * 8); d_out = c.dev_alloc(n * 4); d_cnt = c.dev_alloc(G * 8)
        c.copy_h2d(d_in, R)
        c.shard_histogram(d_in, n, G, d_cnt)
        c.shard_scatter(d_in, n, G, d_cnt, d_out)
        cnt = np.empty(G, dtype=np.uint64); c.copy_d2h(cnt, d_cnt)
        assert cnt.tolist() == [n // G + 1, n // G, n // G, n // G - 1]       # key 1235 (low bits 3) went to shard 0 as key 0
        c.reserve("atomic", n // G * 2, 0)
        c.build_keys(d_out, int(cnt[0]), 2, 2 * (n // G) * 2)
        with pytest.raises(hj.HashJoinError) as e:
            c.fetch()
        assert e.value.status == hj.HJ_ERR_KEY_RANGE


def test_skew_probe_side_zipf(ctx):
    """BASELINE config 5 at reduced size: R unique (shuffle), |S| = 8|R| + 3 drawn Zipf(0.9) over R's key
    domain (mc/src/genzipf.c method). Every S tuple finds exactly its one R tuple: totalMatches = |S|,
    on the open-addressing path (both build variants) and on PRJ."""
    n = 1 << 18
    R = oracle.generate_data("shuffle", n)
    S = hj.generate_data("zipf", 8 * n + 3, n, 16, zipf_theta=0.9)
    assert np.array_equal(S, oracle.generate_zipf(8 * n + 3, n, 0.9, 0))
    want = oracle.build_probe_seq(R, S, 4)
    assert want["conflicts"] == 0 and want["totalMatches"] == S.size
    for variant in (1, 2, 3, 4):
        got = ctx.run("atomic", R, S, buildVariant=variant)
        check_oa(got, want)
    got = ctx.run("prj", R, S)
    assert got["totalMatches"] == S.size


def test_streaming_zipf_equals_the_host_generator():
    """hj_zipf_open / hj_zipf_next_dev: the device-side binary search over the host's rand() stream, in slices of odd
    lengths, gives element for element what hj_generate_data("zipf") (seed 0) and hj_generate_relation("zipf", seed)
    give in one piece -- which tests/test_oracle_golden.py pins to the reference's own gen_zipf."""
    for alphabet, theta, seed, slices in ((1 << 16, 0.9, 0, (1000, 1, 65536 + 7, 12345)), (1000, 1.05, 54321, (70000, 5)),
                                          (1, 0.9, 0, (10,)), (1 << 18, 0.0, 12345, (1 << 17, 3))):
        total = sum(slices)
        want = (hj.generate_data("zipf", total, alphabet, 16, zipf_theta=theta) if seed == 0 else
                hj.generate_relation("zipf", total, alphabet, 0, theta, seed))
        with hj.HashJoinContext(0) as c:
            d = c.dev_alloc(total * 8)
            c.zipf_open(alphabet, theta, seed)
            off = 0
            for m in slices:
                c.zipf_next(m, d + 8 * off)
                off += m
            got = np.empty(total, dtype=np.uint64)
            c.copy_d2h(got, d)
            c.zipf_close()
            with pytest.raises(hj.HashJoinError):
                c.zipf_next(1, d)
            c.dev_free(d)
        assert np.array_equal(got, want), (alphabet, theta, seed)


def test_main_cli_on_gpu():
    """The reference's command line end to end on the GPU: `main --algo atomic|htm|prj` prints the reference's
    JSON fields (NoCCHashBuild.hpp:127-146 order) with the pinned sequential-order values."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main = os.path.join(root, "htm-hashjoin_amd", "bin", "main")

    def run(*args):
        r = subprocess.run([main, *map(str, args)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout)

    j = run("--algo", "atomic", "--rSize", 1048576, "--probeLength", 4, "--dataDistr", "uniform")
    assert list(j)[:8] == ["algo", "rSize", "probeLength", "hashBuildTimeInMicroseconds", "conflicts", "totalMatches",
                           "inputSum", "outputSum"]
    assert (j["conflicts"], j["totalMatches"], j["inputSum"]) == (176864, 871712, 549507039110)
    assert j["outputSum"] == j["inputSum"] and j["device"] == "hip"     # table sum + dropped keys = input sum
    j = run("--algo", "htm", "--transactionSize", 16, "--rSize", 1048576, "--dataDistr", "local_shuffle", "--shuffleRange", 1024)
    assert (j["algo"], j["transactionSize"], j["conflictCount"], j["failedTransactions"], j["totalMatches"], j["inputSum"],
            j["outputSum"], j["numBuckets"], j["overflowBuckets"]) == ("htm", 16, 0, 0, 1048576, 549756338176, 549756338176, 1 << 19, 0)
    j = run("--algo", "htm", "--rSize", 1048576, "--dataDistr", "uniform")           # duplicate keys: buckets overflow into chains
    assert j["conflictCount"] > 0 and j["overflowBuckets"] > 0 and j["outputSum"] == j["inputSum"] == 549507039110
    j = run("--algo", "atomic", "--rSize", 65536, "--dataDistr", "sorted", "--probe", 0)                 # ENABLE_PROBE 0
    assert "totalMatches" not in j and j["conflicts"] == 0
    j = run("--algo", "auto", "--rSize", 1048576, "--dataDistr", "local_shuffle", "--shuffleRange", 1024)
    assert (j["algoUsed"], j["conflicts"], j["totalMatches"], j["outputSum"]) == ("atomic", 0, 1048576, 549756338176)
    j = run("--algo", "auto", "--rSize", 1048576, "--dataDistr", "shuffle", "--radixBits", 14)
    assert (j["algoUsed"], j["totalMatches"], j["results"]) == ("prj", 1048576, 33030144)
    j = run("--algo", "prj", "--rSize", 1048576, "--dataDistr", "shuffle", "--radixBits", 14)
    assert j["totalMatches"] == 1048576 and j["results"] == 33030144                                      # mc PRO "Results"
