#!/usr/bin/env python3
"""Throughput against relation size: the table join (`uniform` W=16, the metric's workload) and the radix join
(`local_shuffle` W=1024, BASELINE config 3) at |R| = |S| = 2^lo .. 2^hi, median of `--reps` steps, one JSON line per size.

    python tools/size_sweep.py [--lo 22] [--hi 31] [--reps 5] > profiles/rNN_size_sweep.jsonl
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import htm_hashjoin_amd as hj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lo", type=int, default=22)
    ap.add_argument("--hi", type=int, default=31)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    for e in range(a.lo, a.hi + 1):
        n = 1 << e
        with hj.HashJoinContext(0) as c, hj.HashJoinContext(0) as p:
            dR = c.dev_alloc(n * 8); dS = c.dev_alloc(n * 8)
            c.copy_h2d(dS, np.arange(1, n + 1, dtype=np.uint64))
            R = hj.generate_data("uniform", n, n, 16); c.copy_h2d(dR, R); del R
            c.reserve("atomic", n, n)
            rows = []
            for _ in range(a.reps + 1):
                c.build(dR, n); c.probe(dS, n); rows.append(c.fetch())
            rows = rows[1:]
            med = lambda k: statistics.median(r[k] for r in rows)                      # noqa: E731
            step = med("total_us") + med("clear_us")
            line = {"log2n": e, "table_join_uniform_us": round(step, 1), "table_join_mtuples_per_s": round(2 * n / step, 1),
                    "build_us": round(med("build_us"), 1), "buildPhaseA_us": round(med("buildPhaseA_us"), 1),
                    "probe_us": round(med("probe_us"), 1), "buildVariant": rows[-1]["buildVariant"],
                    "phaseA_frac_of_8TBps": round(16.0 * n / (med("buildPhaseA_us") * 1e-6) / 8e12, 3) if med("buildPhaseA_us") else None}
            R = hj.generate_data("local_shuffle", n, n, 1024); c.copy_h2d(dR, R); del R
            p.reserve("prj", n, n)
            rows = []
            for _ in range(a.reps + 1):
                p.prj_join(dR, n, dS, n); rows.append(p.fetch())
            rows = rows[1:]
            assert rows[-1]["totalMatches"] == n
            t = statistics.median(r["total_us"] for r in rows)
            line.update({"radix_join_local_shuffle_1024_us": round(t, 1), "radix_join_mtuples_per_s": round(2 * n / t, 1),
                         "prjPath": rows[-1]["prjPath"], "radixBits": rows[-1]["radixBits"]})
            print(json.dumps(line), flush=True)
            c.dev_free(dR); c.dev_free(dS)


if __name__ == "__main__":
    main()
