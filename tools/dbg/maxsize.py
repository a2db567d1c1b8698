"""Largest relation the table join takes: rSize = 2^31 (table of 2^32 slots = 32 GiB), local_shuffle W=1024 and S sorted;
unique keys => conflicts 0, totalMatches 2^31, inputSum = N(N+1)/2. ~110 GiB of HBM, ~35 GiB of host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import htm_hashjoin_amd as hj
n = 1 << 31
t0 = time.time()
R = hj.generate_data("local_shuffle", n, n, 1024)
print("R generated", round(time.time() - t0, 1), "s", flush=True)
with hj.HashJoinContext(0) as c:
    dR = c.dev_alloc(n * 8); c.copy_h2d(dR, R); del R
    S = hj.generate_data("sorted", n)
    dS = c.dev_alloc(n * 8); c.copy_h2d(dS, S); del S
    print("copied", round(time.time() - t0, 1), "s", flush=True)
    c.reserve("atomic", n, n)
    for _ in range(2):
        c.build(dR, n); c.probe(dS, n)
    c.checksums()
    r = c.fetch()
    print({k: r[k] for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "buildVariant", "buildDeferred", "build_us", "probe_us")})
    assert r["conflicts"] == 0 and r["totalMatches"] == n and r["inputSum"] == n * (n + 1) // 2 == r["tableSumFull"]
    print("max size OK", 2 * n / (r["build_us"] + r["probe_us"]), "Mtuples/s")
# the radix join at the same size (element indices are 32-bit: sizes up to 2^32 - 2)
with hj.HashJoinContext(0) as c:
    R = hj.generate_data("local_shuffle", n, n, 1024)
    dR = c.dev_alloc(n * 8); c.copy_h2d(dR, R); del R
    S = hj.generate_data("sorted", n)
    dS = c.dev_alloc(n * 8); c.copy_h2d(dS, S); del S
    c.reserve("prj", n, n)
    c.prj_join(dR, n, dS, n)
    r = c.fetch()
    print({k: r[k] for k in ("totalMatches", "prjChecksum", "radixBits", "partition_us", "join_us")})
    assert r["totalMatches"] == n
    print("PRJ max size OK", 2 * n / r["total_us"], "Mtuples/s")
