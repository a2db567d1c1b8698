#!/usr/bin/env python3
"""Development check of the compact ring build (buildVariant 4) against the sequential oracle: counters and the whole
exported table, for inputs where it must hold (reported variant 4) and where it must hand over to the classic build."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import htm_hashjoin_amd as hj
from oracle import oracle

bad = 0
with hj.HashJoinContext(0) as ctx:
    for n in (1 << 10, 1 << 14, 1 << 16, 1 << 20, 1 << 22):
        for dist, w in (("uniform", 16), ("uniform", 2), ("sorted", 16), ("local_shuffle", 16), ("local_shuffle", 128),
                        ("local_shuffle", 1024), ("shuffle", 16), ("random", 16)):
            R = oracle.generate_data(dist, n, n, w)
            S = oracle.relS_for(dist, R)
            want = oracle.build_probe_seq(R, S, 4, want_table=True)
            for variant in (4, 0):
                got = ctx.run("atomic", R, S, buildVariant=variant)
                ok = all(got[k] == want[k] for k in ("conflicts", "totalMatches", "inputSum", "tableSumHalf", "tableSumFull", "conflictSum"))
                tab = np.array_equal(ctx.export_table(2 * n), want["table"])
                flag = "ok " if ok and tab else "BAD"
                bad += flag == "BAD"
                print(flag, f"n=2^{n.bit_length()-1} {dist}:{w} asked {variant} ran {got['buildVariant']} conflicts {got['conflicts']}/{want['conflicts']} "
                      f"matches {got['totalMatches']}/{want['totalMatches']} table {tab} deferred {got['buildDeferred']} fallbackCause {got['compactFallback']}", flush=True)
    # other probe lengths
    for n in (1 << 16, 1 << 20):
        R = oracle.generate_data("uniform", n, n, 16); S = oracle.generate_data("sorted", n)
        for plen in (1, 2, 3, 5, 8):
            want = oracle.build_probe_seq(R, S, plen, want_table=True)
            got = ctx.run("atomic", R, S, probeLength=plen, buildVariant=4)
            ok = all(got[k] == want[k] for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"))
            tab = np.array_equal(ctx.export_table(2 * n), want["table"])
            bad += not (ok and tab)
            print("ok " if ok and tab else "BAD", f"n=2^{n.bit_length()-1} probeLength {plen} ran {got['buildVariant']} conflicts {got['conflicts']}/{want['conflicts']} "
                  f"conflictSum {got['conflictSum']}/{want['conflictSum']} table {tab} cause {got['compactFallback']}", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
