#!/usr/bin/env python3
"""Extracts the machine-independent fields of the reference's own committed run
logs (experiments/**) into tests/golden/reference_logs.json.

These logs are the only golden outputs the reference holds for this path
(SURVEY.md 8c): counts and checksums printed by the authors' runs. Timings are
dropped. The run configuration of each line follows from the sweep script that
produced the log (experiments/probe.sh, AtomicsVsHTMVsNoCC.sh, motivation.sh),
restated below as data.

Run in the build container only (needs /root/reference):
    python tests/golden/make_reference_logs.py
"""
import json
import os
import re

REF = os.environ.get("HJ_REFERENCE", "/root/reference")
EXP = os.path.join(REF, "experiments")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_logs.json")
KEEP = ("algo", "rSize", "probeLength", "transactionSize", "conflicts", "conflictCount", "totalMatches", "inputSum", "outputSum")


def json_lines(path):
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith("{"):
                rows.append(json.loads(line))
    return rows


def keep(row):
    return {k: row[k] for k in KEEP if k in row}


cases = []

# experiments/probe.sh: 28 x nocc, 28 x atomic, 28 x htm(tSize 16); local_shuffle, W = 2^0..2^27
for rep in range(1, 6):
    path = os.path.join(EXP, "new_backup", f"probe_log{rep}")
    rows = json_lines(path)
    assert len(rows) == 84, (path, len(rows))
    for i, row in enumerate(rows):
        c = keep(row)
        c.update(source=f"experiments/new_backup/probe_log{rep}:{i + 1}", script="experiments/probe.sh",
                 dataDistr="local_shuffle", shuffleRange=2 ** (i % 28), probe=1)
        cases.append(c)

# experiments/AtomicsVsHTMVsNoCC.sh: build only; nocc/atomic/htm(tSize 1) x sorted/shuffle
for rep in range(1, 6):
    path = os.path.join(EXP, "new_backup", f"AtomicsVsHTMVsNoCC_log{rep}")
    rows = json_lines(path)
    assert len(rows) == 6, (path, len(rows))
    for i, row in enumerate(rows):
        c = keep(row)
        c.update(source=f"experiments/new_backup/AtomicsVsHTMVsNoCC_log{rep}:{i + 1}",
                 script="experiments/AtomicsVsHTMVsNoCC.sh", dataDistr=("sorted", "shuffle")[i % 2],
                 shuffleRange=16, probe=0)
        cases.append(c)

# experiments/overflow_log1: htm runs on `uniform` at 2^27 -- only inputSum is machine independent
rows = json_lines(os.path.join(EXP, "overflow_log1"))
sums = sorted({(r["rSize"], r["inputSum"]) for r in rows})
uniform_input_sums = [{"rSize": r, "inputSum": s, "dataDistr": "uniform", "shuffleRange": 16,
                       "source": "experiments/overflow_log1"} for r, s in sums]

# experiments/old/uniform_log: nocc / atomic / htm on `uniform` at several sizes. Counts there are from parallel
# (order dependent) runs; inputSum is a function of DataGen + glibc rand() only.
rows = json_lines(os.path.join(EXP, "old", "uniform_log"))
for r_, s_ in sorted({(r["rSize"], r["inputSum"]) for r in rows}):
    uniform_input_sums.append({"rSize": r_, "inputSum": s_, "dataDistr": "uniform", "shuffleRange": 16,
                               "source": "experiments/old/uniform_log"})
uniform_parallel_conflicts = sorted({(r["algo"], r["rSize"], r["conflicts"]) for r in rows if "conflicts" in r})

# experiments/motivation.sh: mc PRO "Results" (sum of bucket idx) and npo_probe: NPO match count
mc = []
for name, algo in (("motivation_log1", "PRO"), ("npo_probe_log1", "NPO")):
    path = os.path.join(EXP, "new_backup", name)
    with open(path) as f:
        text = f.read()
    vals = sorted({int(v) for v in re.findall(r"Results = (\d+)\. DONE", text)})
    mc.append({"algo": algo, "rSize": 2 ** 27, "results": vals, "source": f"experiments/new_backup/{name}"})

with open(OUT, "w") as f:
    json.dump({"cases": cases, "uniform_input_sums": uniform_input_sums,
               "uniform_parallel_conflicts_not_pins": [list(t) for t in uniform_parallel_conflicts], "mc": mc}, f, indent=0)
print(f"wrote {OUT}: {len(cases)} json cases, {len(uniform_input_sums)} uniform sums, {len(mc)} mc rows")
