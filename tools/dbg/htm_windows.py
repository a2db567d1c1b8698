#!/usr/bin/env python3
"""Development tool: the bucketised table's build (+ probe) at 2^log2n, local_shuffle W in --windows, with every build
variant the table has (0 = the sampler's pick, 3 rings, 2 workgroup window, 1 global atomics): where each one stops paying."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import htm_hashjoin_amd as hj  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=27)
    ap.add_argument("--windows", default="512,1024,2048")
    ap.add_argument("--variants", default="0,2,1")
    ap.add_argument("--dist", default="local_shuffle")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    n = 1 << a.log2n
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); dS = c.dev_alloc(n * 8)
        c.copy_h2d(dS, np.arange(1, n + 1, dtype=np.uint64))
        for w in (int(x) for x in a.windows.split(",")):
            R = hj.generate_data(a.dist, n, n, w); c.copy_h2d(dR, R); del R
            for v in (int(x) for x in a.variants.split(",")):
                c.reserve("htm", n, n, buildVariant=v)
                rows = []
                for _ in range(a.reps + 1):
                    c.build(dR, n); c.probe(dS, n); rows.append(c.fetch())
                rows = rows[1:]
                med = lambda k: round(statistics.median(r[k] for r in rows), 1)                  # noqa: E731
                r = rows[-1]
                print(json.dumps({"W": w, "asked": v, "variant": r["buildVariant"], "build_us": med("build_us"), "phaseA_us": med("buildPhaseA_us"),
                                  "probe_us": med("probe_us"), "clear_us": med("clear_us"), "deferred": r["buildDeferred"],
                                  "conflicts": r["conflicts"], "matches": r["totalMatches"]}), flush=True)
        c.dev_free(dR); c.dev_free(dS)


if __name__ == "__main__":
    main()
