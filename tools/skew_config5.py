#!/usr/bin/env python3
"""BASELINE configs[4] (skew stress) at full size: |R| = 2^28 unique keys (local_shuffle W=1024), |S| = 2^32 > 4 * 10^9
DISTINCT Zipf(0.9) draws over R's key domain, streamed in 16 slices of 2^28 by hj_zipf_next_dev (the serial rand() stream
on the host, gen_zipf's LUT search on the GPU). Per table kind one JSON line: device time of the probes (HIP events),
probes per second, wall time including the host-bound generation, and the check totalMatches = |S|.

    python tools/skew_config5.py [--log2r 28] [--slices 16] > profiles/rNN_skew_config5.jsonl

Each line carries a `roofline` object for its probe kernel: bound = HBM, achieved = the algorithmic 16 B per probe (S tuple 8 +
home slot 8) over the kernel's HIP-event time; `traffic` = HBM bytes per launch from separate rocprofv3 --pmc
FETCH_SIZE / --pmc WRITE_SIZE passes of this very tool (tools/pmc_cmd.sh ... python3 tools/skew_config5.py --slices 4), handed
in with --fetch-json / --write-json (FETCH doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads; for
the scattered table reads of this workload the doubling is an upper bound -- both values are given)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import htm_hashjoin_amd as hj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2r", type=int, default=28)
    ap.add_argument("--slices", type=int, default=16)
    ap.add_argument("--theta", type=float, default=0.9)
    ap.add_argument("--fetch-json", default=None, help="gpurun_out/pmc_TAG/pmc.json of a --pmc FETCH_SIZE pass of this tool")
    ap.add_argument("--write-json", default=None)
    a = ap.parse_args()

    def pmc_bytes(path, counter, kernel):
        if not path or not os.path.exists(path):
            return None
        for k, v in json.load(open(path)).items():
            if kernel in k and counter in v:
                return v[counter]["mean"] * 1024.0          # KiB per launch
        return None
    n = per = 1 << a.log2r
    R = hj.generate_data("local_shuffle", n, n, 1024)
    for algo in ("atomic", "htm"):
        with hj.HashJoinContext(0) as c:
            dR = c.dev_alloc(n * 8); c.copy_h2d(dR, R)
            dS = c.dev_alloc(per * 8)
            c.reserve(algo, n, per)
            c.build(dR, n)
            built = c.fetch()
            t0 = time.perf_counter()
            c.zipf_open(n, a.theta, 0)
            t_tables = time.perf_counter() - t0
            probe_us, gen_s = 0.0, 0.0
            t0 = time.perf_counter()
            for _ in range(a.slices):
                g0 = time.perf_counter()
                c.zipf_next(per, dS)
                gen_s += time.perf_counter() - g0
                c.probe(dS, per)
                probe_us += c.fetch()["probe_us"]
            wall = time.perf_counter() - t0
            c.zipf_close()
            r = c.fetch()
            total = a.slices * per
            kern = "k_htm_probe" if algo == "htm" else "k_probe"
            us_per_launch = probe_us / a.slices
            alg = 16.0 * per
            fetch, write = pmc_bytes(a.fetch_json, "FETCH_SIZE", kern), pmc_bytes(a.write_json, "WRITE_SIZE", kern)
            roofline = {"bound": "hbm", "kernel": kern, "unit": "GB/s", "peak": 8000.0,
                        "achieved": alg / (us_per_launch * 1e-6) / 1e9, "frac": alg / (us_per_launch * 1e-6) / 1e9 / 8000.0,
                        "algorithmic_bytes_per_launch": alg, "launch_us": us_per_launch,
                        "traffic": (2.0 * fetch + (write or 0.0)) if fetch is not None else None,
                        "traffic_fetch_as_reported": fetch, "traffic_write": write,
                        "note": "S is read once in order (8 B per probe); the home slots are Zipf(0.9)-distributed over a table of "
                                "2 * |R| slots: every probe that misses the caches costs a whole DRAM access for 8 useful bytes"}
            print(json.dumps({
                "config": "skew stress (BASELINE configs[4])", "algo": algo, "rSize": n, "sSize": total, "distinct_draws": True,
                "zipfTheta": a.theta, "slices": a.slices, "build_us": built["build_us"], "buildVariant": built["buildVariant"],
                "probe_us_total": probe_us, "probes_per_s": total / (probe_us * 1e-6),
                "probe_GBps_of_16B_per_probe": 16.0 * total / (probe_us * 1e-6) / 1e9,
                "zipf_tables_s (alphabet permutation + LUT, host)": t_tables,
                "wall_s_incl_host_rand_stream": wall, "host_rand_and_upload_s": gen_s,
                "roofline": roofline,
                "totalMatches": r["totalMatches"], "checks": {"every_probe_matches_once": r["totalMatches"] == total,
                                                              "conflicts": r["conflicts"]}}), flush=True)
            c.dev_free(dR); c.dev_free(dS)


if __name__ == "__main__":
    main()
