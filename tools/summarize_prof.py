#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into small CSV/JSON summaries for profiles/.
usage: summarize_prof.py stats <dir> <out.csv> | pmc <dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def stats(d, out):
    rows = []
    for f in glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            w.writerow([r["Name"].split("(")[0][:80], r["Calls"], f"{float(r['TotalDurationNs']) / 1e3:.1f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", f"{float(r['MinNs']) / 1e3:.2f}",
                        f"{float(r['MaxNs']) / 1e3:.2f}", r["Percentage"]])


def pmc(d, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()}
           for k, cs in acc.items() if k.startswith(("hj::", "void hj::"))}
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
