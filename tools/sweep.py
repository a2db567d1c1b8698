#!/usr/bin/env python3
"""The reference's shuffle-window sweep at the reference's protocol, on the MI355X engine with the CPU beside it.

experiments/probe.sh and motivation.sh run, for W = 2^0 .. 2^27 at rSize = 2^27 with `local_shuffle`, the operators
nocc, atomic, htm (transactionSize 16) and mc's PRO, one JSON line per run; experiments/runner.sh:3-43 repeats every
script N = 5 times (figs/perf.png plots the result). This harness does the same sweep:

    for every W:   GPU  atomic (open-addressing table), htm (bucketised table), prj (radix join), auto (adaptive)
                   CPU  nocc and atomic: the product's host-thread port of the reference's loops (`main --algo nocc |
                        cpu-atomic`, csrc/main.cpp) on this machine's cores -- at EVERY W, like the reference
    N = 5 repeats each, the MEDIAN reported (`hashBuildTimeInMicroseconds`), all repeats kept in `runs_us`

and emits the reference's JSON fields first and in its order, so a parser of the reference's logs reads these too.
Every repeat's counters are ASSERTED against the pins the reference's own logs hold for this sweep (unique keys:
conflicts 0, totalMatches = rSize, inputSum = N(N+1)/2, outputSum = the per-operator value of
tests/golden/reference_logs.json; PRO's checksum in closed form), so a line that is printed is a line that is right.

    python tools/sweep.py [--log2n 27] [--repeats 5] [--no-cpu] [--max-log2w K] > profiles/rNN_sweep.jsonl

--protocol motivation: the reference's OTHER sweep script, experiments/motivation.sh:9-31 -- the same windows, BUILD ONLY
(its binaries were built with ENABLE_PROBE 0, config.h:4), with mc's radix join as the first leg:

    for every W:   CPU  mc PRO, the reference's own binary built from its sources (oracle/_ref/mchashjoins --algo=PRO
                        --r-size=2^27 --s-size=2 --local-shuffle-range=W), its printed "Results" asserted against the value
                        every block of experiments/new_backup/motivation_log* holds (549688705024 at 2^27 = the closed form)
                   CPU  nocc and atomic build-only (`main --algo nocc|cpu-atomic --probe 0`)
                   GPU  atomic / htm / prj build-only (table build; bucketised table build; partition R + per-partition
                        tables = what PRO does in this fork)
    then experiments/AtomicsVsHTMVsNoCC.sh:6-11: build-only on `sorted` and `shuffle`, CPU and GPU, asserted against
    experiments/new_backup/AtomicsVsHTMVsNoCC_log1:1-6 (conflicts 0, inputSum, the two outputSum values)
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAIN = os.path.join(ROOT, "htm-hashjoin_amd", "bin", "main")


def expected(algo, n, radix_bits=None):
    """What every run of the sweep must print (local_shuffle = unique keys 1..n): the reference's logged values
    (experiments/new_backup/probe_log*, motivation_log*: tests/golden/reference_logs.json) as closed forms in n."""
    tri = n * (n + 1) // 2
    if algo == "nocc":                          # NoCCHashBuild.hpp:94 sums table[0..rSize) only: key n sits in slot n
        return {"conflicts": 0, "totalMatches": n, "inputSum": tri, "outputSum": tri - n}
    if algo in ("atomic", "auto-table"):
        return {"conflicts": 0, "totalMatches": n, "inputSum": tri, "outputSum": tri}
    if algo == "htm":
        return {"conflictCount": 0, "totalMatches": n, "inputSum": tri, "outputSum": tri}
    if algo in ("prj", "auto-prj"):             # fork's PRO "Results" = sum of (k >> bits) & (nextpow2(n / 2^bits) - 1)
        per = max(n >> radix_bits, 1)
        mask = (1 << (per - 1).bit_length()) - 1 if per > 1 else 0
        total = 0
        for v in range((n >> radix_bits) + 1):
            lo, hi = max(v << radix_bits, 1), min(((v + 1) << radix_bits) - 1, n)
            if hi >= lo:
                total += (v & mask) * (hi - lo + 1)
        return {"totalMatches": n, "results": total}
    raise ValueError(algo)


def check(line, want):
    for k, v in want.items():
        if line[k] != v:
            raise AssertionError(f"sweep pin violated: {k} = {line[k]}, expected {v}: {line}")


def summarise(runs_us):
    return {"hashBuildTimeInMicroseconds": int(statistics.median(runs_us)), "repeats": len(runs_us),
            "runs_us": [int(x) for x in runs_us]}


def cpu_lines(algo, n, window, repeats):
    """`main --algo nocc|cpu-atomic --repeat N`: the reference's own loops on host threads (one data generation)."""
    out = subprocess.run([MAIN, "--algo", algo, "--rSize", str(n), "--probeLength", "4", "--dataDistr", "local_shuffle",
                          "--shuffleRange", str(window), "--repeat", str(repeats)], capture_output=True, text=True, check=True).stdout
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


MC = os.path.join(ROOT, "oracle", "_ref", "mchashjoins")


def effective_cpus():
    sys.path.insert(0, ROOT)
    from bench import effective_cpus as f
    return f()


def mc_pro_lines(n, window, repeats, threads):
    """mc PRO from the reference's sources, motivation.sh:11 (s-size 2: the fork's PRO ignores S). One dict per repeat."""
    import re
    rows = []
    for _ in range(repeats):
        out = subprocess.run([MC, f"--nthreads={threads}", f"--r-size={n}", "--s-size=2", "--algo=PRO",
                              f"--local-shuffle-range={window}"], capture_output=True, text=True, check=True).stdout
        rows.append({"us": float(re.search(r"TOTAL-TIME-USECS[^\n]*\n\s*([0-9.]+)", out).group(1)),
                     "results": int(re.search(r"Results = (\d+)\. DONE", out).group(1))})
    return rows


def cpu_build_only(algo, n, dist, window, repeats):
    out = subprocess.run([MAIN, "--algo", algo, "--rSize", str(n), "--probeLength", "4", "--dataDistr", dist,
                          "--shuffleRange", str(window), "--probe", "0", "--repeat", str(repeats)],
                         capture_output=True, text=True, check=True).stdout
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def motivation(a, hj):
    """experiments/motivation.sh + AtomicsVsHTMVsNoCC.sh (see the module docstring)."""
    n = 1 << a.log2n
    top = a.log2n if a.max_log2w is None else min(a.max_log2w, a.log2n)
    tri = n * (n + 1) // 2
    threads = min(64, effective_cpus())
    have_mc = os.path.exists(MC) and not a.no_cpu

    def emit(line):
        line["mtuples_per_s"] = n / max(line["hashBuildTimeInMicroseconds"], 1)        # build only: |R| / t, as BASELINE.md quotes
        print(json.dumps(line), flush=True)

    def gpu_build_only(ctxs, dR, tag):
        ctx, hctx, pctx = ctxs
        runs, last = [], None
        for _ in range(a.repeats):
            ctx.build(dR, n); ctx.checksums()
            last = ctx.fetch()
            for k, v in (("conflicts", 0), ("inputSum", tri), ("outputSum", tri)):
                assert last[k] == v, (k, last[k], v, tag)
            runs.append(last["build_us"] + last["clear_us"])
        emit({"algo": "atomic", "rSize": n, "probeLength": 4, **summarise(runs), "conflicts": 0, "inputSum": last["inputSum"],
              "outputSum": last["outputSum"], **tag, "device": "hip", "buildVariant": last["buildVariant"], "probe": 0})
        runs = []
        for _ in range(a.repeats):
            hctx.build(dR, n); hctx.checksums()
            last = hctx.fetch()
            for k, v in (("conflicts", 0), ("inputSum", tri), ("outputSum", tri)):
                assert last[k] == v, (k, last[k], v, tag)
            runs.append(last["build_us"] + last["clear_us"])
        emit({"algo": "htm", "rSize": n, "transactionSize": 16, "probeLength": 4, **summarise(runs), "conflictCount": 0,
              "failedTransactions": 0, "inputSum": last["inputSum"], "outputSum": last["outputSum"], **tag, "device": "hip",
              "buildVariant": last["buildVariant"], "probe": 0})
        runs = []
        for _ in range(a.repeats):
            pctx.prj_join(dR, n, 0, 0)
            last = pctx.fetch()
            assert last["prjChecksum"] == expected("prj", n, last["radixBits"])["results"], (last, tag)
            runs.append(last["total_us"])
        emit({"algo": "prj", "rSize": n, **summarise(runs), "results": last["prjChecksum"], "radixBits": last["radixBits"], **tag,
              "device": "hip", "probe": 0})

    with hj.HashJoinContext(0) as ctx, hj.HashJoinContext(0) as hctx, hj.HashJoinContext(0) as pctx:
        dR = ctx.dev_alloc(n * 8)
        ctx.reserve("atomic", n, 0); hctx.reserve("htm", n, 0); pctx.reserve("prj", n, 0)
        for e in range(0, top + 1):
            W = 1 << e
            tag = {"dataDistr": "local_shuffle", "shuffleRange": W, "protocol": "motivation.sh"}
            if have_mc:
                rows = mc_pro_lines(n, W, a.repeats, threads)
                want = expected("prj", n, 14)["results"]          # NUM_RADIX_BITS 14 (prj_params.h:16): 549688705024 at 2^27
                for r in rows:
                    assert r["results"] == want, (r, want)
                emit({"algo": "PRO", "rSize": n, **summarise([r["us"] for r in rows]), "results": want, **tag, "device": "cpu",
                      "kind": "reference (mc/src compiled where it lies: oracle/_ref/mchashjoins)", "cpu_threads": threads})
            if not a.no_cpu:
                for algo, name, osum in (("nocc", "nocc", tri - n), ("cpu-atomic", "atomic", tri)):
                    rows = cpu_build_only(algo, n, "local_shuffle", W, a.repeats)
                    for r in rows:
                        assert (r["conflicts"], r["inputSum"], r["outputSum"]) == (0, tri, osum), r
                    emit({"algo": name, "rSize": n, "probeLength": 4, **summarise([r["hashBuildTimeInMicroseconds"] for r in rows]),
                          "conflicts": 0, "inputSum": tri, "outputSum": osum, **tag, "device": "cpu", "cpu_threads": rows[0]["cpu_threads"], "probe": 0})
            R = hj.generate_data("local_shuffle", n, n, W)
            ctx.copy_h2d(dR, R)
            del R
            gpu_build_only((ctx, hctx, pctx), dR, tag)
        # experiments/AtomicsVsHTMVsNoCC.sh:6-11
        for dist in ("sorted", "shuffle"):
            tag = {"dataDistr": dist, "shuffleRange": 16, "protocol": "AtomicsVsHTMVsNoCC.sh"}
            if not a.no_cpu:
                for algo, name, osum in (("nocc", "nocc", tri - n), ("cpu-atomic", "atomic", tri)):
                    rows = cpu_build_only(algo, n, dist, 16, a.repeats)
                    for r in rows:
                        assert (r["conflicts"], r["inputSum"], r["outputSum"]) == (0, tri, osum), r      # AtomicsVsHTMVsNoCC_log1:1-4
                    emit({"algo": name, "rSize": n, "probeLength": 4, **summarise([r["hashBuildTimeInMicroseconds"] for r in rows]),
                          "conflicts": 0, "inputSum": tri, "outputSum": osum, **tag, "device": "cpu", "cpu_threads": rows[0]["cpu_threads"], "probe": 0})
            R = hj.generate_data(dist, n, n, 16)
            ctx.copy_h2d(dR, R)
            del R
            gpu_build_only((ctx, hctx, pctx), dR, tag)
        ctx.dev_free(dR)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=27)
    ap.add_argument("--repeats", type=int, default=5, help="experiments/runner.sh: N=5")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--max-log2w", type=int, default=None, help="stop the window sweep early (default: up to rSize)")
    ap.add_argument("--protocol", default="probe", choices=["probe", "motivation"],
                    help="probe: experiments/probe.sh (build + probe); motivation: experiments/motivation.sh + "
                         "AtomicsVsHTMVsNoCC.sh (build only, mc PRO as the CPU radix leg)")
    a = ap.parse_args()
    import htm_hashjoin_amd as hj
    if a.protocol == "motivation":
        return motivation(a, hj)
    n = 1 << a.log2n
    top = a.log2n if a.max_log2w is None else min(a.max_log2w, a.log2n)
    S = hj.generate_data("sorted", n)
    with hj.HashJoinContext(0) as ctx, hj.HashJoinContext(0) as hctx, hj.HashJoinContext(0) as pctx, hj.HashJoinContext(0) as actx:
        dS = ctx.dev_alloc(n * 8)
        dR = ctx.dev_alloc(n * 8)
        ctx.copy_h2d(dS, S)
        del S
        ctx.reserve("atomic", n, n)
        hctx.reserve("htm", n, n)
        pctx.reserve("prj", n, n)
        actx.reserve("auto", n, n)
        for e in range(0, top + 1):
            W = 1 << e
            tag = {"dataDistr": "local_shuffle", "shuffleRange": W}
            if not a.no_cpu:
                for algo, name in (("nocc", "nocc"), ("cpu-atomic", "atomic")):
                    rows = cpu_lines(algo, n, W, a.repeats)
                    for r in rows:
                        check(r, expected(name, n))
                    line = {"algo": name, "rSize": n, "probeLength": 4,
                            **summarise([r["hashBuildTimeInMicroseconds"] for r in rows]), "conflicts": 0, "totalMatches": n,
                            "inputSum": rows[0]["inputSum"], "outputSum": rows[0]["outputSum"], **tag, "device": "cpu",
                            "cpu_threads": rows[0]["cpu_threads"]}
                    line["mtuples_per_s"] = 2 * n / max(line["hashBuildTimeInMicroseconds"], 1)
                    print(json.dumps(line), flush=True)
            R = hj.generate_data("local_shuffle", n, n, W)
            ctx.copy_h2d(dR, R)
            del R
            # ---- GPU: open-addressing table (NoCC / Atomic semantics, sequential-order result) ----
            runs, last = [], None
            for _ in range(a.repeats):
                ctx.build(dR, n); ctx.probe(dS, n); ctx.checksums()
                last = ctx.fetch()
                check(last, expected("atomic", n))
                runs.append(last["total_us"] + last["clear_us"])
            line = {"algo": "atomic", "rSize": n, "probeLength": 4, **summarise(runs), "conflicts": last["conflicts"],
                    "totalMatches": last["totalMatches"], "inputSum": last["inputSum"], "outputSum": last["outputSum"], **tag,
                    "device": "hip", "buildVariant": last["buildVariant"], "buildDeferred": last["buildDeferred"],
                    "build_us": last["build_us"], "probe_us": last["probe_us"]}
            line["mtuples_per_s"] = 2 * n / max(line["hashBuildTimeInMicroseconds"], 1)
            print(json.dumps(line), flush=True)
            # ---- GPU: bucketised table (HTMHashBuild semantics) ----
            runs = []
            for _ in range(a.repeats):
                hctx.build(dR, n); hctx.probe(dS, n); hctx.checksums()
                last = hctx.fetch()
                check({"conflictCount": last["conflicts"], **last}, expected("htm", n))
                runs.append(last["total_us"])
            line = {"algo": "htm", "rSize": n, "transactionSize": 16, "probeLength": 4, **summarise(runs), "conflictCount": 0,
                    "failedTransactions": 0, "totalMatches": last["totalMatches"], "inputSum": last["inputSum"],
                    "outputSum": last["outputSum"], **tag, "device": "hip", "buildVariant": last["buildVariant"],
                    "build_us": last["build_us"], "probe_us": last["probe_us"]}
            line["mtuples_per_s"] = 2 * n / max(line["hashBuildTimeInMicroseconds"], 1)
            print(json.dumps(line), flush=True)
            # ---- GPU: radix join (mc PRO + the probe the fork disabled) ----
            runs = []
            for _ in range(a.repeats):
                pctx.prj_join(dR, n, dS, n)
                last = pctx.fetch()
                check({"results": last["prjChecksum"], **last}, expected("prj", n, last["radixBits"]))
                runs.append(last["total_us"])
            line = {"algo": "prj", "rSize": n, **summarise(runs), "totalMatches": last["totalMatches"], "results": last["prjChecksum"],
                    "radixBits": last["radixBits"], **tag, "device": "hip", "partition_us": last["partition_us"], "join_us": last["join_us"]}
            line["mtuples_per_s"] = 2 * n / max(line["hashBuildTimeInMicroseconds"], 1)
            print(json.dumps(line), flush=True)
            # ---- GPU: adaptive (locality pre-round -> table join or radix join) ----
            runs = []
            for _ in range(a.repeats):
                actx.join(dR, n, dS, n)
                last = actx.fetch()
                if last["algoUsed"] == "prj":
                    check({"results": last["prjChecksum"], **last}, expected("auto-prj", n, last["radixBits"]))
                else:
                    actx.checksums()
                    last = actx.fetch()
                    check(last, expected("auto-table", n))
                runs.append(last["total_us"] + last["clear_us"])
            line = {"algo": "auto", "algoUsed": last["algoUsed"], "rSize": n, **summarise(runs), "totalMatches": last["totalMatches"],
                    **tag, "device": "hip"}
            line["mtuples_per_s"] = 2 * n / max(line["hashBuildTimeInMicroseconds"], 1)
            print(json.dumps(line), flush=True)
        ctx.dev_free(dR)
        ctx.dev_free(dS)


if __name__ == "__main__":
    main()
