#!/usr/bin/env python3
"""Runs the reference's own no-partitioning join, oracle/_ref/mchashjoins --algo=NPO (compiled from mc/src where it lies,
oracle/Makefile), on the workloads of mc/src/main.c:343-408 and records its printed "Results" = the true join
cardinality (probe_hashtable, mc/src/no_partitioning_join.c:270-310) into tests/golden/mc_workloads.json. These pin the
restated relation generators (hj_generate_relation / orc_generate_relation, mc/src/generator.c): for --non-unique the
count depends on every key both generators draw. Run in the build container only."""
import json
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "..", "..", "oracle", "_ref", "mchashjoins")

CASES = [  # (r_size, s_size, extra flags, how R and S are generated)
    (1 << 16, 1 << 18, [], ("pk", "fk")),
    (1 << 16, 3 * (1 << 16) + 7, [], ("pk", "fk")),
    (100000, 250001, [], ("pk", "fk")),
    (1 << 16, 1 << 17, ["--non-unique"], ("nonunique", "nonunique")),
    (100000, 300000, ["--non-unique"], ("nonunique", "nonunique")),
    (1 << 20, 1 << 20, ["--non-unique"], ("nonunique", "nonunique")),
    (1 << 16, 1 << 18, ["--skew=1.05"], ("pk", "zipf")),
    (1 << 16, 1 << 16, ["--local-shuffle-range=1024"], ("pk_lshuffle", "fk")),
]

rows = []
for r, s, flags, kinds in CASES:
    out = subprocess.run([BIN, "--algo=NPO", "--nthreads=2", f"--r-size={r}", f"--s-size={s}"] + flags,
                         capture_output=True, text=True, check=True).stdout
    res = int(re.search(r"Results = (\d+)\. DONE", out).group(1))
    rows.append({"rSize": r, "sSize": s, "flags": flags, "rKind": kinds[0], "sKind": kinds[1], "rSeed": 12345, "sSeed": 54321,
                 "results": res})
    print(r, s, flags, res)
json.dump({"provenance": "oracle/_ref/mchashjoins --algo=NPO (reference mc/src, gcc -O3, committed configure flags); "
                         "R seed 12345, S seed 54321 (mc/src/main.c:337-338)", "rows": rows},
          open(os.path.join(HERE, "mc_workloads.json"), "w"), indent=1)
