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
    # a step = everything between two probes: the dispatches after the previous step's k_probe (and the copy of the counters
    # that follows it) up to the last k_probe. (The locality sample of a context with a preference runs on a stream of its
    # own beside the build: it shows up inside the step, overlapping its neighbours -- negative gaps.)
    probes = [i for i, r in enumerate(rows) if "k_probe" in r[2]]
    last = probes[-1]
    first = probes[-2] + 1
    while "copyBuffer" in rows[first][2]:
        first += 1
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
