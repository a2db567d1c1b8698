#!/usr/bin/env python3
"""Times the build (and probe) kernels of one build variant through the C ABI, no torch: development tool.

    python tools/time_build.py --log2n 27 --variant 3 --dists uniform:16,sorted:16 [--reps 7]

Prints one JSON line per distribution: median HIP-event times of phase A (the LDS kernel alone), of the whole build
group and of the probe, the counters, and the algorithmic GB/s of phase A (16 B per R tuple)."""
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
    ap.add_argument("--log2n", type=int, default=27)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--dists", default="uniform:16")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    n = 1 << a.log2n
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8)
        dS = c.dev_alloc(n * 8)
        c.copy_h2d(dS, np.arange(1, n + 1, dtype=np.uint64))
        c.reserve("atomic", n, n, buildVariant=a.variant)
        for spec in a.dists.split(","):
            dist, w = spec.split(":")
            R = hj.generate_data(dist, n, n, int(w))
            c.copy_h2d(dR, R)
            del R
            rows = []
            for _ in range(a.reps + 1):
                c.build(dR, n)
                c.probe(dS, n)
                rows.append(c.fetch())
            rows = rows[1:]
            med = {k: statistics.median(r[k] for r in rows) for k in ("buildPhaseA_us", "build_us", "probe_us", "clear_us")}
            r = rows[-1]
            pa = med["buildPhaseA_us"] or med["build_us"]
            print(json.dumps({"tag": a.tag, "dist": spec, "log2n": a.log2n, "variant": r["buildVariant"],
                              **{k: round(v, 1) for k, v in med.items()},
                              "phaseA_GBps": round(16.0 * n / pa / 1e3, 1), "frac_of_8TBps": round(16.0 * n / pa / 1e3 / 8000, 3),
                              "conflicts": r["conflicts"], "matches": r["totalMatches"], "deferred": r["buildDeferred"]}), flush=True)
        c.dev_free(dR)
        c.dev_free(dS)


if __name__ == "__main__":
    main()
