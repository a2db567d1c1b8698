#!/usr/bin/env python3
"""tools/step_timeline.sh's second half: the dispatches of the LAST build + probe step in a rocprofv3 kernel trace."""
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a step starts at the sampler (variant 0) or, with a forced variant, at the first kernel after a k_table_sums
    starts = [i for i, r in enumerate(rows) if "k_sample_locality" in r[2]]
    if not starts:
        ends = [i for i, r in enumerate(rows) if "k_table_sums" in r[2]]
        starts = [e + 1 for e in ends[:-1]]
    first = starts[-1]
    last = max(i for i, r in enumerate(rows) if i >= first)
    t0 = rows[first][0]
    prev_end = None
    busy = 0
    print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  kernel")
    for s, e, name in rows[first:last + 1]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {name[:110]}")
        prev_end = e
        busy += e - s
    print(f"# {last + 1 - first} dispatches, {busy / 1e3:.1f} us inside kernels, {(rows[last][1] - t0) / 1e3:.1f} us from the first start to the last end")


if __name__ == "__main__":
    main()
