#!/usr/bin/env python3
"""Why did the CPU baseline (the reference's CAS loops, oracle port, 16 threads) run at a quarter of its 2^27 rate at 2^30
(round-2 VERDICT, weak #8)? Rate by size, with the table first touched by the calling thread and by the worker threads,
plus what the host says about its memory (NUMA nodes, THP). Host-only: no GPU call."""
import glob
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import htm_hashjoin_amd as hj
from oracle import oracle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import effective_cpus

cores = effective_cpus()
threads = min(64, cores)
print("effective cpus", cores, "affinity", sorted(os.sched_getaffinity(0))[:4], "...", len(os.sched_getaffinity(0)), flush=True)
for f in ["/sys/kernel/mm/transparent_hugepage/enabled", "/sys/kernel/mm/transparent_hugepage/defrag"]:
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, e)
for node in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    try:
        mem = [l for l in open(node + "/meminfo") if "MemTotal" in l or "MemFree" in l]
        print(os.path.basename(node), open(node + "/cpulist").read().strip(), " ".join(x.split(":")[1].strip() for x in mem))
    except OSError as e:
        print(node, e)
for log2n in (26, 27, 28, 29, 30):
    n = 1 << log2n
    R = hj.generate_data("uniform", n, n, 16)
    S = np.arange(1, n + 1, dtype=np.uint64)
    for touch in (False, True):
        best = None
        for _ in range(2):
            r = oracle.build_probe_mt(R, S, 4, 64, threads, atomic=True, parallel_touch=touch)
            us = (r["build_us"], r["probe_us"])
            best = us if best is None or sum(us) < sum(best) else best
        print(f"2^{log2n} touch={'workers' if touch else 'caller'}: build {best[0]/1e3:.1f} ms probe {best[1]/1e3:.1f} ms -> {2*n/sum(best)/1e3:.2f} Gtuples/s", flush=True)
    del R, S
