"""Randomised differential test of the open-addressing builds against the sequential oracle: inputs the DataGen
distributions do not produce -- near-sorted multisets with bursts of equal keys, key domains from a quarter of the
relation (most tuples are dropped) to the whole table (sparse keys, wrap-around at the table's end), disorder from none
to far beyond what the rings hold, every probe length -- through the device's pick (0), the workgroup window (2), the
classic rings (3) and the compact rings (4; where they cannot hold they must hand over, and the table must still be the sequential one).
Counters and the whole table, slot for slot. HJ_FUZZ_CASES (default 36) sets the number of cases; the seed is fixed."""
import os

import numpy as np
import pytest

import htm_hashjoin_amd as hj
from oracle import oracle

pytestmark = pytest.mark.gpu


def make_relation(rng, n):
    """A near-sorted relation of n tuples (value = key, as DataGen's): sorted keys with bursts, then displaced by < W."""
    table = 2 * n
    kind = rng.integers(0, 5)
    if kind == 0:          # dense unique keys
        keys = np.arange(1, n + 1, dtype=np.uint64)
    elif kind == 1:        # random multiset over a domain of n * f keys
        f = rng.choice([0.25, 0.5, 1.0, 1.5, 1.99])
        keys = np.sort(rng.integers(1, max(2, int(n * f)), size=n, dtype=np.uint64))
    elif kind == 2:        # bursts: few distinct keys, geometric multiplicities
        distinct = np.sort(rng.choice(np.arange(1, table, dtype=np.uint64), size=max(1, n // int(rng.integers(2, 9))), replace=False))
        counts = rng.geometric(0.3, size=distinct.size)
        keys = np.repeat(distinct, counts)[:n]
        if keys.size < n:
            keys = np.concatenate([keys, np.arange(1, n - keys.size + 1, dtype=np.uint64) + keys[-1]])
        keys = np.sort(keys)
    elif kind == 3:        # sparse keys over the whole table (walks wrap around its end) and a dense stretch at the very top
        keys = np.sort(np.concatenate([rng.integers(1, table, size=n - n // 8, dtype=np.uint64),
                                       np.arange(table - n // 8, table, dtype=np.uint64)]))
    else:                  # two interleaved dense runs (every key twice, far apart in value order only by 1)
        keys = np.sort(np.concatenate([np.arange(1, n // 2 + 1, dtype=np.uint64)] * 2))
    keys = keys[:n].astype(np.uint64)
    keys = np.sort((keys - np.uint64(1)) % np.uint64(table - 1) + np.uint64(1))        # into [1, table - 1], still sorted
    w = int(rng.choice([1, 2, 4, 8, 16, 16, 16, 32, 48, 64, 100, 300, 2000]))
    if w > 1:
        order = np.argsort(np.arange(n) + rng.uniform(0, w, size=n), kind="stable")
        keys = keys[order]
    return np.ascontiguousarray(keys), w


@pytest.mark.parametrize("block", range(4))
def test_ring_builds_on_random_near_sorted_relations(block):
    cases = int(os.environ.get("HJ_FUZZ_CASES", "36"))
    rng = np.random.default_rng(20260000 + block)
    with hj.HashJoinContext(0) as ctx:
        for case in range(block, cases, 4):
            n = 1 << int(rng.integers(12, 21))
            R, w = make_relation(rng, n)
            plen = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8]))
            S = np.ascontiguousarray(rng.permutation(R)[: n]) if rng.integers(0, 2) else np.arange(1, n + 1, dtype=np.uint64)
            want = oracle.build_probe_seq(R, S, plen, want_table=True)
            for variant in (0, 2, 3, 4):
                got = ctx.run("atomic", R, S, probeLength=plen, buildVariant=variant)
                tag = (block, case, n, w, plen, variant, got["buildVariant"], got["compactFallback"])
                for k in ("conflicts", "totalMatches", "inputSum", "tableSumHalf", "tableSumFull", "conflictSum"):
                    assert got[k] == want[k], (k, got[k], want[k], tag)
                assert np.array_equal(ctx.export_table(2 * n), want["table"]), tag
                if variant == 4 and got["buildVariant"] == 3:
                    assert got["compactFallback"] != 0, tag              # a hand-over always says why
                if variant == 3:
                    assert got["buildVariant"] == 3 and got["compactFallback"] == 0, tag


@pytest.mark.parametrize("block", range(2))
def test_bucketised_table_on_random_near_sorted_relations(block):
    """The same relations through --algo htm: rings with the chain phase in LDS (or, where that gives up, the generic
    chain kernels: bit 8 of compactFallback), window, global atomics, and the device's pick -- every counter, the primary
    buckets tuple for tuple, the overflow buckets as a multiset, and (small cases) every chain in walk order."""
    cases = int(os.environ.get("HJ_FUZZ_CASES", "36")) // 2
    rng = np.random.default_rng(20261000 + block)
    with hj.HashJoinContext(0) as ctx:
        for case in range(block, cases, 2):
            n = 1 << int(rng.integers(10, 19))
            R, w = make_relation(rng, n)
            R = np.ascontiguousarray(R[: n - int(rng.integers(0, 3))])          # htm takes any size
            S = np.ascontiguousarray(rng.permutation(R)) if rng.integers(0, 2) else np.arange(1, R.size + 1, dtype=np.uint64)
            want = oracle.htm_build_probe_seq(R, S, want_buckets=True)
            key = lambda o: np.sort(o[1:].view(np.uint64).reshape(-1, 4)[:, :3].sum(axis=1))          # noqa: E731
            for variant in (0, 3, 2, 1):
                got = ctx.run("htm", R, S, buildVariant=variant)
                tag = (block, case, R.size, w, variant, got["buildVariant"], got["compactFallback"])
                assert (got["conflicts"], got["conflictSum"], got["totalMatches"], got["inputSum"], got["tableSumFull"],
                        got["htmOverflowBuckets"], got["htmOverflowSum"], got["outputSum"]) == (
                    want["conflictCount"], want["conflictSum"], want["totalMatches"], want["inputSum"], want["bucketSum"],
                    want["overflowBuckets"], want["overflowSum"], want["outputSum"]), tag
                buckets, overflows = ctx.export_buckets(want["numBuckets"])
                assert np.array_equal(buckets["tuples"], want["buckets"]["tuples"]) and np.array_equal(buckets["count"], want["buckets"]["count"]), tag
                assert overflows.size == want["overflows"].size and np.array_equal(key(overflows), key(want["overflows"])), tag
                if R.size <= 1 << 13:
                    a, ao = oracle.htm_chains(buckets, overflows)
                    b, bo = oracle.htm_chains(want["buckets"], want["overflows"])
                    assert np.array_equal(ao, bo) and np.array_equal(a, b), tag


@pytest.mark.parametrize("block", range(2))
def test_radix_join_on_random_relations(block):
    """--algo prj on relations of any size and shape: |R| != |S|, key domains from dense to 31 bits, skew in the low key bits
    (pass 1 of the histogram-free path must give up and the exact passes redo the join), in the middle bits (pass 2 gives
    up), many equal keys (partitions beyond one LDS table), through prjMode 0 (the size rule), 1 (exact passes) and 2
    (histogram-free at any size) and several radix widths. Match count and the fork's checksum against the oracle."""
    cases = int(os.environ.get("HJ_FUZZ_CASES", "36")) // 2
    rng = np.random.default_rng(20262000 + block)
    with hj.HashJoinContext(0) as ctx:
        for case in range(block, cases, 2):
            nr = int(rng.integers(1 << 10, 1 << 21))
            ns = int(rng.integers(1 << 10, 1 << 21))
            shape = int(rng.integers(0, 5))
            if shape == 0:                                   # dense domain
                hi = max(nr, ns)
                R = rng.integers(1, hi + 1, size=nr, dtype=np.uint64); S = rng.integers(1, hi + 1, size=ns, dtype=np.uint64)
            elif shape == 1:                                 # 31-bit keys (few matches)
                R = rng.integers(1, 1 << 31, size=nr, dtype=np.uint64); S = np.concatenate([R[: min(nr, ns) // 2], rng.integers(1, 1 << 31, size=ns - min(nr, ns) // 2, dtype=np.uint64)])
            elif shape == 2:                                 # low bits constant: one pass-1 bin takes everything
                c = int(rng.integers(0, 256))
                R = (rng.integers(1, 1 << 20, size=nr, dtype=np.uint64) << np.uint64(8)) | np.uint64(c); S = (rng.integers(1, 1 << 20, size=ns, dtype=np.uint64) << np.uint64(8)) | np.uint64(c)
            elif shape == 3:                                 # middle bits constant: pass 2 cannot spread a partition
                R = (rng.integers(1, 1 << 12, size=nr, dtype=np.uint64) << np.uint64(16)) | rng.integers(0, 256, size=nr, dtype=np.uint64)
                S = (rng.integers(1, 1 << 12, size=ns, dtype=np.uint64) << np.uint64(16)) | rng.integers(0, 256, size=ns, dtype=np.uint64)
            else:                                            # repeated keys: ~2 ... 64 tuples per key (more is quadratic work for any hash join, the oracle's included)
                d = int(rng.integers(max(nr, ns) // 64 + 1, max(nr, ns)))
                R = rng.integers(1, d + 1, size=nr, dtype=np.uint64); S = rng.integers(1, d + 1, size=ns, dtype=np.uint64)
            R[R == 0] = 1; S[S == 0] = 1
            bits = int(rng.choice([0, 8, 11, 12, 14, 16]))
            for mode in (0, 1, 2):
                got = ctx.run("prj", R, S, radixBits=bits, prjMode=mode)
                want = oracle.prj_join(R, S, got["radixBits"])
                tag = (block, case, nr, ns, shape, bits, mode, got["prjPath"])
                assert (got["totalMatches"], got["prjChecksum"]) == (want["matches"], want["checksum"]), tag
            if shape in (0, 1):
                assert got["totalMatches"] == oracle.true_cardinality(R, S), (block, case)


@pytest.mark.parametrize("block", range(2))
def test_key_builds_of_a_radix_shard_on_random_keys(block):
    """hj_build_keys_dev / hj_probe_keys_dev (what a radix shard runs on the 32-bit keys it received): any number of keys,
    any home shift 0..6, tables from tight to roomy, unaligned key buffers, every probe length, all build variants --
    counters and the whole table against the sequential oracle with the shard's table size and shift."""
    cases = int(os.environ.get("HJ_FUZZ_CASES", "36")) // 2
    rng = np.random.default_rng(20263000 + block)
    for case in range(block, cases, 2):
        n2 = 1 << int(rng.integers(11, 20))
        keys64, w = make_relation(rng, n2)
        shift = int(rng.integers(0, 7))
        m = int(rng.integers(n2 // 2 + 1, n2 + 1))                          # the shard's share: any count
        keys = np.ascontiguousarray(((keys64[:m] << np.uint64(shift)) | np.uint64(rng.integers(0, 1 << shift))).astype(np.uint32))
        keys = keys[keys != 0]
        m = keys.size
        table_size = 2 * n2
        plen = int(rng.choice([1, 2, 4, 4, 4, 8]))
        S = np.ascontiguousarray(rng.permutation(keys)[: max(1, m // 2)])
        want = oracle.build_probe_seq_ts(keys.astype(np.uint64), S.astype(np.uint64), table_size, shift, plen, want_table=True)
        for variant in (0, 1, 2, 3, 4):
            with hj.HashJoinContext(0) as c:
                c.reserve("atomic", n2, S.size, buildVariant=variant, probeLength=plen)
                ro, so = 4 * int(rng.integers(0, 4)), 4 * int(rng.integers(0, 4))
                d_r = c.dev_alloc((m + 8) * 4); d_s = c.dev_alloc((S.size + 8) * 4)
                c.copy_h2d(d_r + ro, keys); c.copy_h2d(d_s + so, S)
                c.build_keys(d_r + ro, m, shift, table_size)
                c.probe_keys(d_s + so, S.size)
                c.checksums()
                got = c.fetch()
                tag = (block, case, n2, m, w, shift, plen, variant, got["buildVariant"], got["compactFallback"])
                for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"):
                    assert got[k] == want[k], (k, got[k], want[k], tag)
                assert np.array_equal(c.export_table(table_size), want["table"]), tag
                c.dev_free(d_r); c.dev_free(d_s)


@pytest.mark.parametrize("block", range(2))
def test_a_reused_context_on_a_random_sequence_of_relations(block):
    """hj_build_dev with the device's pick on ONE reserved context while the relation changes at random from step to step
    (tight, loose, no locality, duplicate-heavy, unique): only the kernels of the previously preferred variant are enqueued,
    so every step runs on whatever the last sample left -- and must still produce the sequential table."""
    steps = int(os.environ.get("HJ_FUZZ_CASES", "36"))
    rng = np.random.default_rng(20264000 + block)
    n = 1 << (16 + block)
    S = np.arange(1, n + 1, dtype=np.uint64)
    seen = set()
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); dS = c.dev_alloc(n * 8)
        c.copy_h2d(dS, S)
        c.reserve("atomic", n, n)
        for step in range(steps):
            if rng.integers(0, 4) == 0:
                R = rng.permutation(np.arange(1, n + 1, dtype=np.uint64))          # no locality at all
            else:
                R, _ = make_relation(rng, n)
            R = np.ascontiguousarray(R)
            c.copy_h2d(dR, R)
            c.build(dR, n); c.probe(dS, n); c.checksums()
            got = c.fetch()
            want = oracle.build_probe_seq(R, S, 4, want_table=True)
            seen.add(got["buildVariant"])
            for k in ("conflicts", "totalMatches", "inputSum", "tableSumHalf", "tableSumFull", "conflictSum"):
                assert got[k] == want[k], (block, step, k, got[k], want[k], got["buildVariant"], got["compactFallback"])
            assert np.array_equal(c.export_table(2 * n), want["table"]), (block, step, got["buildVariant"])
        c.dev_free(dR); c.dev_free(dS)
    assert steps < 20 or len(seen) >= 3, seen                                      # the sequence really moved between the builds
