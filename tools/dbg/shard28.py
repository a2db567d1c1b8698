import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import htm_hashjoin_amd as hj
from htm_hashjoin_amd.sharded import HipShardEngine
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
R = hj.generate_data("uniform", n, n, 16)
r = torch.from_numpy(R.view("int64")).cuda()
s = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
e = HipShardEngine(hj, torch, 0)
cnt = e.histogram(r, 1); print("count", cnt.tolist())
out = e.scatter(r, 1, cnt, 0, 0)
torch.cuda.synchronize()
idx = out >> 32
print("idx min/max", int(idx.min()), int(idx.max()), "unique idx", int(torch.unique(idx).numel()) if n <= (1 << 26) else "skip")
print("keys equal multiset:", bool((out & 0xFFFFFFFF).sum() == r.sum()))
print("idx sum ok:", int(idx.sum()) == n * (n - 1) // 2)
e.reserve(2 * n, n, n)
e.build(out, 0, 0, 2 * n); e.probe(s)
res = e.finish(); print({k: res[k] for k in ("conflicts", "totalMatches", "inputSum", "buildVariant", "buildDeferred")}, res["conflicts"] + res["totalMatches"] == n)
with hj.HashJoinContext(0) as c:
    c.reserve("atomic", n, n); c.build(r.data_ptr(), n); c.probe(s.data_ptr(), n); c.checksums(); d = c.fetch()
    print("direct", {k: d[k] for k in ("conflicts", "totalMatches", "inputSum", "buildVariant")}, d["conflicts"] + d["totalMatches"] == n)
