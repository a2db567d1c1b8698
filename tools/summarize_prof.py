#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into small CSV/JSON summaries for profiles/.
usage: summarize_prof.py stats <dir> <out.csv> | pmc <dir> <out.json> | traffic <pmc_fetch.json> <pmc_write.json> <out.json>"""
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


def traffic(fetch_json, write_json, out):
    """HBM bytes per launch and kernel = 2 * FETCH_SIZE + WRITE_SIZE (both counters are in KiB). FETCH_SIZE is doubled: on
    gfx950 it reports half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM; calibrated in round 1
    on k_table_sums and on k_build_own, whose 512-byte-per-wave-instruction reads of R came out at exactly half of the
    8.59 GB it reads once); WRITE_SIZE is exact for 16-byte-per-lane stores. bench.py reads this file for roofline.traffic."""
    f, w = json.load(open(fetch_json)), json.load(open(write_json))
    short = lambda k: k.replace("void ", "").replace("hj::", "").split("<")[0]          # noqa: E731
    res = {}
    for k in sorted(set(f) | set(w)):
        fb = f.get(k, {}).get("FETCH_SIZE", {}).get("mean", 0.0) * 1024.0
        wb = w.get(k, {}).get("WRITE_SIZE", {}).get("mean", 0.0) * 1024.0
        res[short(k)] = res.get(short(k), 0.0) + 2.0 * fb + wb
        res.setdefault("_raw", {})[k] = {"FETCH_SIZE_bytes_as_reported": fb, "WRITE_SIZE_bytes": wb}
    res["_note"] = ("HBM bytes per launch, |R|=|S|=2^30 uniform, from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                    "`python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline`; per kernel name 2 * FETCH_SIZE + WRITE_SIZE "
                    "(template instances of one kernel summed per launch; see tools/summarize_prof.py: traffic)")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    res["_source_hash"] = kernel_source_hash()        # bench.py reports these bytes only while the kernel sources are these
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "traffic": traffic}[sys.argv[1]](*sys.argv[2:])
