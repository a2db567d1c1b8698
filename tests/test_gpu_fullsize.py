"""The metric's own configurations at their full size (BASELINE.json: |R| = |S| = 2^30) on the GPU, against the CPU
oracle and the closed forms the reference's logs pin -- not only through size-independent properties.

  * configs[1]'s operator at the metric's size: `uniform` W=16 (the reference's default --shuffleRange), S = sorted,
    open-addressing build+probe through hj_build_dev / hj_probe_dev, every counter and checksum against the sequential
    oracle (oracle.build_probe_seq: the reference's loops, NoCCHashBuild.hpp:37-81, walked by one thread in input order).
  * configs[2] exactly: PRJ, 2^30, `local_shuffle` W=1024: totalMatches = 2^30 and the fork's printed checksum in
    closed form at the radix bits that actually ran (16 at this size), plus inputSum = 2^59 + 2^29 on the same R.

Host memory: ~40 GiB (R, S, the oracle's 16 GiB table). A few minutes, most of it the serial rand() stream of DataGen."""
import numpy as np
import pytest

import htm_hashjoin_amd as hj
from oracle import oracle

pytestmark = pytest.mark.gpu

N = 1 << 30


def pro_closed_form(n, bits):
    """sum over k = 1..n of (k >> bits) & (nextpow2(n / 2^bits) - 1): the fork's PRO "Results" for the unique keys 1..n
    (mc/src/parallel_radix_join.c:242-256; tests/test_oracle_golden.py pins it to the reference binary and to
    experiments/new_backup/motivation_log1:8). Evaluated per value of k >> bits, so it costs nothing at 2^30."""
    per = max(n >> bits, 1)
    mask = (1 << (per - 1).bit_length()) - 1 if per > 1 else 0
    total = 0
    for v in range((n >> bits) + 1):
        lo, hi = max(v << bits, 1), min(((v + 1) << bits) - 1, n)
        if hi >= lo:
            total += (v & mask) * (hi - lo + 1)
    return total


def test_closed_form_helper_matches_the_small_one():
    k = np.arange(1, (1 << 20) + 1, dtype=np.uint64)
    for bits in (9, 14, 16):
        per = (1 << 20) >> bits
        mask = (1 << (per - 1).bit_length()) - 1
        assert pro_closed_form(1 << 20, bits) == int(((k >> np.uint64(bits)) & np.uint64(mask)).sum())
    assert pro_closed_form(1 << 27, 14) == 549688705024                  # motivation_log1:8


def test_metric_size_uniform_against_the_sequential_oracle():
    R = hj.generate_data("uniform", N, N, 16)
    S = hj.generate_data("sorted", N)
    want = oracle.build_probe_seq(R, S, 4)
    assert want["totalMatches"] + want["conflicts"] == N
    R32, S32 = R.astype(np.uint32), S.astype(np.uint32)
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(N * 8); c.copy_h2d(dR, R)
        dS = c.dev_alloc(N * 8); c.copy_h2d(dS, S)
        del R, S
        for variant in (0, 4, 2):                  # what the bench runs (auto -> the classic rings: duplicate keys), the compact rings, the workgroup window
            c.reserve("atomic", N, N, buildVariant=variant)
            c.build(dR, N)
            c.probe(dS, N)
            c.checksums()
            got = c.fetch()
            assert got["buildVariant"] == (3 if variant == 0 else variant) and got["compactFallback"] == 0
            for k in ("conflicts", "totalMatches", "inputSum", "tableSumHalf", "tableSumFull", "conflictSum"):
                assert got[k] == want[k], (variant, k, got[k], want[k])
            assert got["outputSum"] == want["outputSumAtomic"]
        # the bare-key entry points a radix shard runs after the exchange, at the same size: at 2^30 the wavefront build cuts
        # the relation into eight rounds' worth of chunks, and the 32-bit instances of the kernels take that path too
        c.copy_h2d(dR, R32); c.copy_h2d(dS, S32)
        del R32, S32
        c.reserve("atomic", N, N)
        c.build_keys(dR, N, 0, 2 * N)
        c.probe_keys(dS, N)
        c.checksums()
        got = c.fetch()
        assert got["buildVariant"] == 3 and got["compactFallback"] == 0
        for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"):
            assert got[k] == want[k], ("keys", k, got[k], want[k])
        c.reserve("atomic", N, N, buildVariant=4)          # ... and the compact rings on the bare keys
        c.build_keys(dR, N, 0, 2 * N)
        c.probe_keys(dS, N)
        c.checksums()
        got = c.fetch()
        assert got["buildVariant"] == 4 and got["compactFallback"] == 0
        for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"):
            assert got[k] == want[k], ("keys compact", k, got[k], want[k])
        c.dev_free(dR); c.dev_free(dS)
    # the numbers every bench line of rounds 1 and 2 printed for this workload
    assert (want["conflicts"], want["totalMatches"]) == (180852797, 892889027)


def test_config3_prj_local_shuffle_1024_closed_forms():
    R = hj.generate_data("local_shuffle", N, N, 1024)
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(N * 8); c.copy_h2d(dR, R)
        del R
        dS = c.dev_alloc(N * 8); c.copy_h2d(dS, np.arange(1, N + 1, dtype=np.uint64))
        for mode, path in ((0, 1), (1, 0)):         # the histogram-free passes (what the bench times), then the exact ones
            c.reserve("prj", N, N, prjMode=mode)
            c.prj_join(dR, N, dS, N)
            p = c.fetch()
            assert p["prjPath"] == path
            assert p["radixBits"] == 16 and p["prjPartitions"] == 1 << 16
            assert p["totalMatches"] == N
            assert p["prjChecksum"] == pro_closed_form(N, 16)
        with hj.HashJoinContext(0) as c2:          # the same relation through the table join: inputSum and the unique-key sums
            c2.reserve("atomic", N, N)
            c2.build(dR, N); c2.probe(dS, N); c2.checksums()
            r = c2.fetch()
        assert (r["conflicts"], r["totalMatches"], r["inputSum"], r["tableSumFull"]) == (0, N, 576460752840294400, 576460752840294400)
        c.dev_free(dR); c.dev_free(dS)


def test_config5_skew_stress_full_size():
    """BASELINE configs[4]: |R| = 2^28 unique keys, |S| = 2^32 (> 4 * 10^9) DISTINCT Zipf(0.9) draws over R's key domain,
    streamed in 16 slices of 2^28 (hj_zipf_next_dev: the serial rand() stream on the host, the LUT search on the GPU).
    R holds every key of the domain exactly once, so every probe finds exactly its one partner: totalMatches = |S|, on
    the open-addressing table and on the bucketised one. A slice is also copied back and compared with the host
    generator's first 2^20 draws."""
    n, slices, per = 1 << 28, 16, 1 << 28
    R = hj.generate_data("local_shuffle", n, n, 1024)
    with hj.HashJoinContext(0) as c, hj.HashJoinContext(0) as h:
        dR = c.dev_alloc(n * 8); c.copy_h2d(dR, R)
        del R
        dS = c.dev_alloc(per * 8)
        c.reserve("atomic", n, per)
        c.build(dR, n)
        h.reserve("htm", n, per)
        h.build(dR, n)
        c.zipf_open(n, 0.9, 0)
        for k in range(slices):
            c.zipf_next(per, dS)
            c.probe(dS, per)
            c.synchronize()                        # dS is reused; h runs on its own stream
            h.probe(dS, per)
            h.synchronize()
            if k == 0:
                head = np.empty(1 << 20, dtype=np.uint64)
                c.copy_d2h(head, dS)
        c.zipf_close()
        r, rh = c.fetch(), h.fetch()
        c.dev_free(dR); c.dev_free(dS)
    assert (r["conflicts"], r["sSize"], r["totalMatches"]) == (0, slices * per, slices * per)
    assert (rh["conflicts"], rh["totalMatches"]) == (0, slices * per)
    assert slices * per >= 4_000_000_000
    # the stream's head against the one-piece host generator at a smaller total (same seed, same alphabet: same prefix)
    assert np.array_equal(head[:4096], hj.generate_data("zipf", 4096, n, 16, zipf_theta=0.9))


def test_htm_uniform_2p27_chains_in_lds_against_the_sequential_oracle():
    """The bucketised table at the size its numbers are quoted at (2^27, `uniform`: 30 M conflicts in 17 M overflow
    buckets): every counter of the sequential restatement of HTMHashBuild.hpp, through the ring build with the chain phase
    in LDS (hj_htm.hip: `compactFallback` bit 8 clear = it did not hand over to the generic kernels) and, forced, through
    the window build with the generic chain kernels."""
    n = 1 << 27
    R = hj.generate_data("uniform", n, n, 16)
    S = hj.generate_data("sorted", n)
    want = oracle.htm_build_probe_seq(R, S)
    assert (want["conflictCount"], want["overflowBuckets"]) == (30065252, 17330401)       # DESIGN.md 4.6 quotes these
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); c.copy_h2d(dR, R)
        dS = c.dev_alloc(n * 8); c.copy_h2d(dS, S)
        for variant in (0, 2):
            c.reserve("htm", n, n, buildVariant=variant)
            c.build(dR, n)
            c.probe(dS, n)
            c.checksums()
            got = c.fetch()
            assert got["buildVariant"] == (3 if variant == 0 else 2)
            assert got["compactFallback"] == 0, (variant, got["compactFallback"])
            assert (got["conflicts"], got["conflictSum"], got["totalMatches"], got["inputSum"], got["tableSumFull"],
                    got["htmOverflowBuckets"], got["htmOverflowSum"], got["outputSum"]) == (
                want["conflictCount"], want["conflictSum"], want["totalMatches"], want["inputSum"], want["bucketSum"],
                want["overflowBuckets"], want["overflowSum"], want["outputSum"]), variant
        c.dev_free(dR); c.dev_free(dS)
