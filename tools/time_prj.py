#!/usr/bin/env python3
"""Times the radix join through the C ABI (development tool): python tools/time_prj.py --log2n 30 [--dist local_shuffle:1024]"""
import argparse, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import htm_hashjoin_amd as hj

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=30)
ap.add_argument("--dist", default="local_shuffle:1024")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tag", default="")
ap.add_argument("--mode", type=int, default=0, help="hj_params.prjMode: 0 auto, 1 exact passes, 2 histogram-free at any size")
a = ap.parse_args()
n = 1 << a.log2n
dist, w = a.dist.split(":")
with hj.HashJoinContext(0) as c:
    dR = c.dev_alloc(n * 8); dS = c.dev_alloc(n * 8)
    c.copy_h2d(dS, np.arange(1, n + 1, dtype=np.uint64))
    R = hj.generate_data(dist, n, n, int(w)); c.copy_h2d(dR, R); del R
    c.reserve("prj", n, n, prjMode=a.mode)
    rows = []
    for _ in range(a.reps + 1):
        c.prj_join(dR, n, dS, n)
        rows.append(c.fetch())
    rows = rows[1:]
    med = {k: round(statistics.median(r[k] for r in rows), 1) for k in ("total_us", "partition_us", "join_us", "prjScatterPass1R_us")}
    print(json.dumps({"tag": a.tag, "dist": a.dist, "log2n": a.log2n, **med, "matches": rows[-1]["totalMatches"], "prjPath": rows[-1]["prjPath"]}), flush=True)
