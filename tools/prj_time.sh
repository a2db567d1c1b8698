python - <<PY
import sys; sys.path.insert(0,'.')
import htm_hashjoin_amd as hj, numpy as np
for log2n in (27, 30):
    n = 1 << log2n
    R = hj.generate_data("local_shuffle", n, n, 1024); S = hj.generate_data("sorted", n)
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n*8); dS = c.dev_alloc(n*8); c.copy_h2d(dR, R); c.copy_h2d(dS, S)
        c.reserve("prj", n, n)
        best=None
        for _ in range(4):
            c.prj_join(dR, n, dS, n); r = c.fetch()
            if best is None or r["total_us"] < best["total_us"]: best = r
        print(log2n, "bits", best["radixBits"], "partition_us", round(best["partition_us"]), "join_us", round(best["join_us"]), "total", round(best["total_us"]), "matches ok", best["totalMatches"] == n, "Gt/s", round(2*n/best["total_us"]/1e3,1))
PY
