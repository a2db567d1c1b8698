"""Per-GPU time of the 8-GPU weak-scaling workload's local join: `uniform` keys squeezed 2:1 into the rank's key range
(tuples outnumber 32-bit keys, every key about twice). usage: python tools/dbg/squeezed.py [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import htm_hashjoin_amd as hj
from htm_hashjoin_amd.sharded import squeeze_into_range
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
for width in (n, n // 2):
    R = squeeze_into_range(hj.generate_data("uniform", n, n, 16), n, 0, width, np)
    S = squeeze_into_range(np.arange(1, n + 1, dtype=np.uint64), n, 0, width, np)
    with hj.HashJoinContext(0) as c:
        dR = c.dev_alloc(n * 8); dS = c.dev_alloc(n * 8)
        c.copy_h2d(dR, R); c.copy_h2d(dS, S)
        c.reserve("atomic", n, n)
        best = None
        for _ in range(4):
            c.build(dR, n); c.probe(dS, n)
            r = c.fetch()
            t = r["build_us"] + r["probe_us"] + r["clear_us"]
            best = t if best is None else min(best, t)
        print(f"width n/{n // width}: build {r['build_us']:.0f} us probe {r['probe_us']:.0f} us, step {best:.0f} us, variant {r['buildVariant']}, "
              f"conflicts {r['conflicts']} ({100.0 * r['conflicts'] / n:.1f} %), deferred {r['buildDeferred']}")
