#!/usr/bin/env python3
"""When the wavefronts of k_build_wave start and finish (development tool; needs a library built with -DHJ_WV_CLOCKS:
tools/mk_variant.sh clk hj_build_wave "-DHJ_WV_CLOCKS", loaded by path -- the product library is never touched).

    HJ_DEV_LIB_VARIANT=tools/variants/clk.so python tools/wave_clocks.py --log2n 27 [--dist uniform:16]

Prints, in microseconds from the first wavefront's start: percentiles of the start times, of the end times and of the
durations over all chunks, and how many wavefronts are still running at 80 / 90 / 95 / 99 % of the kernel's span."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import htm_hashjoin_amd as hj

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=27)
ap.add_argument("--dist", default="uniform:16")
a = ap.parse_args()
n = 1 << a.log2n
dist, w = a.dist.split(":")
from htm_hashjoin_amd import _lib as _hjlib
lib = ctypes.CDLL(_hjlib.LIB_PATH)            # the variant HJ_DEV_LIB_VARIANT names (it must export hj_debug_wave_clocks)
with hj.HashJoinContext(0) as c:
    dR = c.dev_alloc(n * 8)
    R = hj.generate_data(dist, n, n, int(w)); c.copy_h2d(dR, R); del R
    c.reserve("atomic", n, n, buildVariant=3)
    for _ in range(3):
        c.build(dR, n)
        r = c.fetch()
    m = 32768
    buf = np.zeros(2 * m, dtype=np.uint32)
    rc = lib.hj_debug_wave_clocks(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.c_uint32(m))
    assert rc == 0, rc
    st, en = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
    live = en != 0
    k = int(live.sum())
    st, en = st[live], en[live]
    t0 = st.min()
    st = (st - t0) / 100.0; en = (en - t0) / 100.0          # 100 MHz ticks -> us
    span = en.max()
    pct = lambda v: [round(float(x), 1) for x in np.percentile(v, [0, 1, 10, 50, 90, 99, 100])]
    running = {f"{int(f * 100)}%": int(((st <= f * span) & (en > f * span)).sum()) for f in (0.5, 0.8, 0.9, 0.95, 0.99)}
    dur = en - st
    idx = np.nonzero(live)[0]
    grp = lambda key, m: [round(float(dur[key == g].mean()), 1) for g in range(m)]
    groups = {"by_wave_in_workgroup (c % 4)": grp(idx % 4, 4),
              "by_workgroup % 8 (XCD if workgroups go round robin)": grp((idx // 4) % 8, 8),
              "by_(workgroup // 8) % 4": grp((idx // 32) % 4, 4),
              "by_position_in_R (eighths)": grp(idx * 8 // max(int(idx.max()) + 1, 1), 8),
              "by_workgroup // 256 (dispatch round of 256 workgroups)": grp(np.minimum(idx // 1024, 7), 8) if k > 1024 else None}
    print(json.dumps({"log2n": a.log2n, "mean_duration_us": groups, "dist": a.dist, "chunks": k, "phaseA_us": round(r["buildPhaseA_us"], 1), "span_us": round(float(span), 1),
                      "start_us_pct_0_1_10_50_90_99_100": pct(st), "end_us_pct": pct(en), "duration_us_pct": pct(en - st),
                      "wavefronts_running_at": running}))
    c.dev_free(dR)
