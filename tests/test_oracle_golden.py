"""The oracle (oracle/hj_oracle.c) against every golden vector the reference holds
for this path: its committed run logs, the survey's sequential runs of the
reference headers, and oracle/_ref/mchashjoins outputs. CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_datagen_leading_values(golden_dir):
    for row in _load(golden_dir, "survey_appendix.json")["appendix_c"]:
        a = oracle.generate_data(row["dist"], row["n"], row["n"], row["window"])
        assert a[: len(row["first"])].tolist() == row["first"], row["dist"]
        assert a[-len(row["last"]):].tolist() == row["last"], row["dist"]
        assert int(a.sum()) == row["sum"]


@pytest.mark.parametrize("max_size", [1 << 20])
def test_appendix_b_sequential(golden_dir, max_size):
    for row in _load(golden_dir, "survey_appendix.json")["appendix_b"]:
        if row["rSize"] > max_size:
            continue
        R = oracle.generate_data(row["dist"], row["rSize"], row["rSize"], row["window"])
        S = oracle.relS_for(row["dist"], R)
        got = oracle.build_probe_seq(R, S, 4)
        for k in ("conflicts", "totalMatches", "inputSum", "outputSumNocc"):
            assert got[k] == row[k], (row, k, got[k])
        if "outputSumAtomic" in row:
            assert got["outputSumAtomic"] == row["outputSumAtomic"]
        # invariant for S = 1..N and keys in [1, N]
        if row["dist"] != "random":
            assert got["totalMatches"] + got["conflicts"] == row["rSize"]


@pytest.mark.slow
def test_appendix_b_16m(golden_dir):
    for row in _load(golden_dir, "survey_appendix.json")["appendix_b"]:
        if row["rSize"] != 1 << 24 or row["dist"] != "uniform":
            continue
        R = oracle.generate_data(row["dist"], row["rSize"], row["rSize"], row["window"])
        got = oracle.build_probe_seq(R, oracle.relS_for(row["dist"], R), 4)
        for k in ("conflicts", "totalMatches", "inputSum", "outputSumNocc"):
            assert got[k] == row[k], (k, got[k], row[k])


def test_reference_logs_are_consistent(golden_dir):
    """Every nocc/atomic line of the reference's logs carries the same counts; keep the
    distinct expectations so the two heavyweight tests below cover all 300 lines."""
    logs = _load(golden_dir, "reference_logs.json")
    seen = {(c["algo"], c["probe"], c["conflicts"], c.get("totalMatches"), c["inputSum"], c["outputSum"])
            for c in logs["cases"] if c["algo"] in ("nocc", "atomic")}
    n = 1 << 27
    tri = n * (n + 1) // 2
    assert seen == {("nocc", 1, 0, n, tri, tri - n), ("atomic", 1, 0, n, tri, tri),
                    ("nocc", 0, 0, None, tri, tri - n), ("atomic", 0, 0, None, tri, tri)}


@pytest.mark.slow
def test_reference_log_pins_2p27_local_shuffle(golden_dir):
    """experiments/new_backup/probe_log*: local_shuffle at 2^27 (W = 1024 run in full)."""
    logs = _load(golden_dir, "reference_logs.json")
    want = [c for c in logs["cases"] if c["algo"] == "nocc" and c["probe"] == 1 and c["shuffleRange"] == 1024][0]
    n = want["rSize"]
    R = oracle.generate_data("local_shuffle", n, n, 1024)
    S = oracle.generate_data("sorted", n)
    got = oracle.build_probe_seq(R, S, 4)
    assert got["conflicts"] == want["conflicts"] == 0
    assert got["totalMatches"] == want["totalMatches"]
    assert got["inputSum"] == want["inputSum"]
    assert got["outputSumNocc"] == want["outputSum"]
    atomic = [c for c in logs["cases"] if c["algo"] == "atomic" and c["probe"] == 1][0]
    assert got["outputSumAtomic"] == atomic["outputSum"]


def test_reference_logs_htm_lines_are_consistent(golden_dir):
    """All 150 `htm` lines of the reference's logs (probe.sh with transactionSize 16 and the probe, AtomicsVsHTMVsNoCC.sh
    build only; unique keys): no bucket ever overflows, every probe matches, the checksum equals the input sum."""
    logs = _load(golden_dir, "reference_logs.json")
    n = 1 << 27
    tri = n * (n + 1) // 2
    seen = {(c["probe"], c["conflictCount"], c.get("totalMatches"), c["inputSum"], c["outputSum"])
            for c in logs["cases"] if c["algo"] == "htm"}
    assert seen == {(1, 0, n, tri, tri), (0, 0, None, tri, tri)}


@pytest.mark.parametrize("dist,window", [("local_shuffle", 1), ("local_shuffle", 1024), ("local_shuffle", 1 << 19),
                                         ("sorted", 16), ("shuffle", 16)])
def test_htm_oracle_on_unique_keys_follows_the_log_pins(dist, window):
    """The distributions of those log lines at a size the CPU suite affords: same closed forms (conflictCount 0,
    totalMatches = N, outputSum = inputSum = N(N+1)/2, also for the checksum expression evaluated as written)."""
    n = 1 << 20
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.generate_data("sorted", n)
    got = oracle.htm_build_probe_seq(R, S)
    tri = n * (n + 1) // 2
    assert (got["conflictCount"], got["totalMatches"], got["inputSum"], got["outputSum"], got["outputSumAsWritten"]) == (
        0, n, tri, tri, tri)
    assert got["numBuckets"] == 1 << 19 and got["overflowBuckets"] == 0


@pytest.mark.slow
def test_htm_oracle_2p27_log_pin(golden_dir):
    """experiments/new_backup/probe_log*: the htm line for local_shuffle W = 1024 at 2^27, run in full."""
    logs = _load(golden_dir, "reference_logs.json")
    want = [c for c in logs["cases"] if c["algo"] == "htm" and c["probe"] == 1 and c["shuffleRange"] == 1024][0]
    n = want["rSize"]
    R = oracle.generate_data("local_shuffle", n, n, 1024)
    S = oracle.generate_data("sorted", n)
    got = oracle.htm_build_probe_seq(R, S)
    assert (got["conflictCount"], got["totalMatches"], got["inputSum"], got["outputSumAsWritten"], got["outputSum"]) == (
        want["conflictCount"], want["totalMatches"], want["inputSum"], want["outputSum"], want["outputSum"])


def test_htm_oracle_duplicate_keys_invariants():
    """Duplicate keys (no reference-held golden: the reference's own numbers are run dependent there): conflictCount is
    the order-independent sum over buckets of max(0, tuples - 3); every tuple is stored exactly once (bucket or chain),
    so the probe counts true multiplicities and bucketSum + overflowSum = inputSum."""
    for dist in ("uniform", "random"):
        n = 1 << 16
        R = oracle.generate_data(dist, n, n, 16)
        S = oracle.relS_for(dist, R)
        got = oracle.htm_build_probe_seq(R, S, want_buckets=True)
        b = ((R // np.uint64(3)) & np.uint64(got["numBuckets"] - 1)).astype(np.int64)
        per = np.bincount(b, minlength=got["numBuckets"])
        assert got["conflictCount"] == int(np.maximum(per - 3, 0).sum()) > 0
        assert got["overflowBuckets"] == int(((np.maximum(per - 3, 0) + 2) // 3).sum())
        assert got["totalMatches"] == oracle.true_cardinality(R, S)
        assert got["outputSum"] == got["inputSum"] and got["overflowSum"] == got["conflictSum"]
        assert np.array_equal(got["buckets"]["count"], np.minimum(per, 3).astype(np.uint32))
        flat, off = oracle.htm_chains(got["buckets"], got["overflows"])
        assert np.array_equal(np.diff(off), per)                        # bucket + chain hold exactly the bucket's tuples
        # primary buckets: the first three tuples of the bucket in input order; chain: the rest, newest group first
        for k in np.flatnonzero(per > 3)[:200]:
            mine = R[b == k]
            assert got["buckets"]["tuples"][k].tolist() == mine[:3].tolist()
            rest, walk = mine[3:].tolist(), flat[off[k] + 3: off[k + 1]].tolist()
            groups = [rest[i:i + 3] for i in range(0, len(rest), 3)]
            assert walk == [x for g in reversed(groups) for x in g]


@pytest.mark.slow
def test_reference_log_pin_2p27_uniform_inputsum(golden_dir):
    """experiments/overflow_log1: inputSum of `uniform` at 2^27 (DataGen + glibc rand)."""
    row = _load(golden_dir, "reference_logs.json")["uniform_input_sums"][0]
    R = oracle.generate_data("uniform", row["rSize"], row["rSize"], row["shuffleRange"])
    assert int(R.sum(dtype=np.uint64)) == row["inputSum"]


@pytest.mark.parametrize("log2n", [18, 20, 22, 24])
def test_reference_log_pins_uniform_inputsum_sizes(golden_dir, log2n):
    """experiments/old/uniform_log: inputSum of `uniform` at 2^18..2^24, for the oracle's DataGen AND the product's
    (hj_generate_data needs no GPU). The log's conflict counts come from parallel runs and are not pins."""
    import htm_hashjoin_amd as hj
    rows = {r["rSize"]: r for r in _load(golden_dir, "reference_logs.json")["uniform_input_sums"]}
    row = rows[1 << log2n]
    R = oracle.generate_data("uniform", row["rSize"], row["rSize"], row["shuffleRange"])
    assert int(R.sum(dtype=np.uint64)) == row["inputSum"]
    P = hj.generate_data("uniform", row["rSize"], row["rSize"], row["shuffleRange"])
    assert int(P.sum(dtype=np.uint64)) == row["inputSum"]
    # the atomic rows of the same log print outputSum == inputSum: the whole-table checksum incl. dropped tuples
    got = oracle.build_probe_seq(R, None, 4)
    assert got["outputSumAtomic"] == row["inputSum"]


def _pro_closed_form(n, bits=14):
    """sum over k=1..N of (k >> bits) & (nextpow2(N / 2^bits) - 1): the fork's PRO "Results"
    for unique keys 1..N (every partition holds N/2^bits tuples)."""
    per = max(n >> bits, 1)
    mask = (1 << (per - 1).bit_length()) - 1 if per > 1 else 0
    k = np.arange(1, n + 1, dtype=np.uint64)
    return int(((k >> np.uint64(bits)) & np.uint64(mask)).sum())


def test_prj_checksum_vs_reference_binary(golden_dir):
    """oracle.prj_join against oracle/_ref/mchashjoins outputs (tests/golden/mc_ref.json)."""
    for row in _load(golden_dir, "mc_ref.json")["rows"]:
        n = row["rSize"]
        if n > 1 << 22:
            continue
        R = oracle.generate_data("shuffle", n)
        S = oracle.generate_data("sorted", n)
        got = oracle.prj_join(R, S, 14)
        if row["algo"] == "PRO":
            assert got["checksum"] == row["results"], (n, got)
            assert got["checksum"] == _pro_closed_form(n)
        else:  # NPO: true match count
            assert got["matches"] == row["results"], (n, got)


def test_prj_checksum_vs_reference_log(golden_dir):
    """experiments/new_backup/motivation_log1:8 -- PRO Results = 549688705024 at 2^27."""
    mc = {r["algo"]: r for r in _load(golden_dir, "reference_logs.json")["mc"]}
    assert mc["PRO"]["results"] == [_pro_closed_form(1 << 27)]
    assert mc["NPO"]["results"] == [1 << 27]


def test_prj_matches_equal_true_cardinality():
    for dist in ("uniform", "random", "shuffle"):
        n = 1 << 16
        R = oracle.generate_data(dist, n)
        S = oracle.relS_for(dist, R)
        for bits in (4, 9, 14):
            assert oracle.prj_join(R, S, bits)["matches"] == oracle.true_cardinality(R, S)


def test_mt_port_agrees_on_unique_keys():
    n = 1 << 18
    R = oracle.generate_data("local_shuffle", n, n, 1024)
    S = oracle.generate_data("sorted", n)
    seq = oracle.build_probe_seq(R, S)
    for atomic in (False, True):
        mt = oracle.build_probe_mt(R, S, 4, 64, 4, atomic)
        for k in ("conflicts", "totalMatches", "inputSum", "outputSumNocc", "outputSumAtomic"):
            assert mt[k] == seq[k]


# ---- Zipf generator: pinned by the reference's own gen_zipf -----------------------------------------------------------
def _fnv1a64_u32(keys):
    """FNV-1a-64 over the keys as little-endian uint32 (what oracle/zipf_ref_driver.c prints); vectorised per byte lane
    is not possible for a running hash, so: plain loop over bytes in chunks, in C speed via int.from_bytes tricks is not
    either -- the streams here are <= 2^21 keys = 8 MiB, a python loop over a bytes object is seconds."""
    h = 1469598103934665603
    for b in np.ascontiguousarray(keys, dtype="<u4").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _zipf_rows(golden_dir):
    return _load(golden_dir, "zipf_ref.json")["rows"]


def test_zipf_oracle_matches_the_references_gen_zipf(golden_dir):
    """orc_generate_zipf (oracle/hj_oracle.c) against the reference's gen_zipf compiled from mc/src/genzipf.c
    (tests/golden/zipf_ref.json, made by tests/golden/make_zipf_ref.py): every key of every stream (first keys, sum,
    hash over all). Round 1 compared two restatements by the same hand with each other only."""
    for row in _zipf_rows(golden_dir):
        got = oracle.generate_zipf(row["stream_size"], row["alphabet_size"], row["theta"], row["seed"])
        assert got[:32].tolist() == row["first"], row
        assert int(got.sum()) == row["sum"], row
        if row["stream_size"] <= 1 << 18:
            assert _fnv1a64_u32(got) == row["fnv1a64"], row


def test_zipf_product_generator_matches_the_references_gen_zipf(golden_dir):
    """hj_generate_data("zipf") -- the product's own restatement of glibc rand() + gen_zipf -- for the seed-0 rows."""
    import htm_hashjoin_amd as hj
    n_checked = 0
    for row in _zipf_rows(golden_dir):
        if row["seed"] != 0:
            continue
        got = hj.generate_data("zipf", row["stream_size"], row["alphabet_size"], 16, zipf_theta=row["theta"])
        assert got[:32].tolist() == row["first"], row
        assert int(got.sum()) == row["sum"], row
        if row["stream_size"] <= 1 << 18:
            assert _fnv1a64_u32(got) == row["fnv1a64"], row
        n_checked += 1
    assert n_checked >= 5


def test_zipf_reference_binary_still_agrees_with_the_fixture(golden_dir):
    """Where the reference checkout is present the fixture is re-derived from the freshly built binary."""
    import subprocess
    exe = os.path.join(os.path.dirname(oracle.REF_MCHASHJOINS), "genzipf_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/genzipf_ref not built (no reference checkout here)")
    for row in _zipf_rows(golden_dir)[:4]:
        out = subprocess.run([exe, str(row["stream_size"]), str(row["alphabet_size"]), repr(row["theta"]), str(row["seed"]), "32"],
                             capture_output=True, text=True, check=True).stdout
        assert json.loads(out) == row


# ---- mc's relation generators, pinned by the reference binary's true join cardinalities -------------------------------
def _mc_relations(row, gen):
    skew = next((float(f.split("=")[1]) for f in row["flags"] if f.startswith("--skew=")), 0.0)
    win = next((int(f.split("=")[1]) for f in row["flags"] if f.startswith("--local-shuffle-range=")), 0)
    R = gen(row["rKind"], row["rSize"], row["rSize"], win, 0.0, row["rSeed"])
    S = gen(row["sKind"], row["sSize"], row["rSize"], 0, skew, row["sSeed"])        # maxid of S = |R| (mc/src/main.c:391-407)
    return R, S


def test_mc_generators_reproduce_the_reference_binarys_cardinalities(golden_dir):
    """orc_generate_relation (libc rand(), as mc/src/generator.c) feeds a reference-free exact count; it must equal what
    the reference's own NPO printed for the same command line (tests/golden/mc_workloads.json). For --non-unique the
    count depends on every key of both streams."""
    for row in _load(golden_dir, "mc_workloads.json")["rows"]:
        R, S = _mc_relations(row, oracle.generate_relation)
        assert oracle.true_cardinality(R, S) == row["results"], row
        if row["rKind"].startswith("pk"):
            assert np.array_equal(np.sort(R), np.arange(1, row["rSize"] + 1, dtype=np.uint64))
        if "--non-unique" in row["flags"]:
            assert int(R.min()) == 0 and int(R.max()) < row["rSize"]                 # RAND_RANGE(maxid): 0 included


def test_mc_generators_product_equals_libc_restatement(golden_dir):
    """hj_generate_relation (the product's own glibc TYPE_3 stream) element for element against the libc-driven one."""
    import htm_hashjoin_amd as hj
    for row in _load(golden_dir, "mc_workloads.json")["rows"]:
        R, S = _mc_relations(row, oracle.generate_relation)
        Rp, Sp = _mc_relations(row, hj.generate_relation)
        assert np.array_equal(R, Rp) and np.array_equal(S, Sp), row
    with pytest.raises(hj.HashJoinError):
        hj.generate_relation("no_such_kind", 10)
