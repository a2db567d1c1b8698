#!/usr/bin/env python3
"""compact ring build on bare keys with a home shift (what a radix shard runs): why does it hand over?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import htm_hashjoin_amd as hj
from oracle import oracle
for n in (1 << 16, 1 << 20):
    R = oracle.generate_data("uniform", n, n, 16)
    for G in (1, 2, 8):
        shift = G.bit_length() - 1
        for g in range(min(G, 2)):
            keys = R[(R & np.uint64(G - 1)) == g].astype(np.uint32)
            m = keys.size
            r = 1
            while r < m: r *= 2
            table_size = 2 * (n // G)
            with hj.HashJoinContext(0) as c:
                c.reserve("atomic", max(r, table_size // 2), 0, buildVariant=4)
                d = c.dev_alloc((m + 8) * 4)
                c.copy_h2d(d + 4, keys)
                c.build_keys(d + 4, m, shift, table_size)
                c.checksums()
                res = c.fetch()
                want = oracle.build_probe_seq_ts(keys.astype(np.uint64), keys[:1].astype(np.uint64), table_size, shift)
                print(f"n=2^{n.bit_length()-1} G={G} g={g} m={m} ran {res['buildVariant']} cause {res['compactFallback']} conflicts {res['conflicts']}/{want['conflicts']}", flush=True)
                c.dev_free(d)
