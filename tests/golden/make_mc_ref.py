#!/usr/bin/env python3
"""Runs oracle/_ref/mchashjoins (compiled from the reference's own mc/src by
oracle/Makefile) and records its printed "Results" into tests/golden/mc_ref.json.

PRO in this fork is R-side only and prints sum(bucket idx) (parallel_radix_join.c:
256); NPO prints the true match count. R = create_relation_pk (keys 1..N, shuffled),
S = create_relation_fk (mc/src/main.c:362-410). Run in the build container only."""
import json
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "..", "..", "oracle", "_ref", "mchashjoins")
rows = []
for log2n in (10, 14, 16, 18, 20, 22, 24):
    n = 1 << log2n
    for algo in ("PRO", "NPO"):
        out = subprocess.run([BIN, f"--algo={algo}", "--nthreads=4", f"--r-size={n}", f"--s-size={n}"],
                             capture_output=True, text=True, check=True).stdout
        res = int(re.search(r"Results = (\d+)\. DONE", out).group(1))
        rows.append({"algo": algo, "rSize": n, "sSize": n, "results": res})
        print(algo, n, res)
json.dump({"provenance": "oracle/_ref/mchashjoins (reference mc/src, gcc -O3, committed configure flags)",
           "rows": rows}, open(os.path.join(HERE, "mc_ref.json"), "w"), indent=0)
