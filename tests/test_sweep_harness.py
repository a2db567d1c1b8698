"""tools/sweep.py -- the reference's shuffle-window sweep (experiments/probe.sh, motivation.sh, runner.sh) on GPU + CPU.
CPU part: the pins the harness asserts are the values the reference's own logs hold, and a committed sweep file, if
present, is well formed. GPU part: a smoke run of the harness itself at 2^20."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("sweep", os.path.join(ROOT, "tools", "sweep.py"))
sweep = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sweep)

ALGOS_PER_W = {"nocc", "atomic", "htm", "prj", "auto"}


def test_pins_equal_the_reference_logs(golden_dir):
    logs = json.load(open(os.path.join(golden_dir, "reference_logs.json")))
    n = 1 << 27
    for c in logs["cases"]:
        if c["script"] != "experiments/probe.sh":
            continue
        want = sweep.expected(c["algo"], n)
        for k, v in want.items():
            assert c[k] == v, (c, k)
    pro = [r for r in logs["mc"] if r["algo"] == "PRO"][0]
    assert sweep.expected("prj", n, 14)["results"] == pro["results"][0]        # motivation_log1:8
    with pytest.raises(AssertionError):
        sweep.check({"conflicts": 1, "totalMatches": n, "inputSum": 0, "outputSum": 0}, sweep.expected("atomic", n))


def test_summary_is_the_median_of_the_repeats():
    s = sweep.summarise([5.0, 1.0, 9.0, 3.0, 7.0])
    assert s == {"hashBuildTimeInMicroseconds": 5, "repeats": 5, "runs_us": [5, 1, 9, 3, 7]}


def _validate(lines, n, repeats, windows, with_cpu):
    by_w = {}
    for l in lines:
        assert l["rSize"] == n and l["repeats"] == repeats and len(l["runs_us"]) == repeats
        assert l["hashBuildTimeInMicroseconds"] == int(sorted(l["runs_us"])[repeats // 2]) or repeats % 2 == 0
        assert l["totalMatches"] == n and l["dataDistr"] == "local_shuffle"
        by_w.setdefault(l["shuffleRange"], []).append((l["algo"], l["device"]))
    assert sorted(by_w) == windows
    for w, got in by_w.items():
        gpu = {a for a, d in got if d == "hip"}
        assert gpu == {"atomic", "htm", "prj", "auto"}, (w, got)
        if with_cpu:
            assert {a for a, d in got if d == "cpu"} == {"nocc", "atomic"}, (w, got)   # the CPU leg at EVERY W


def test_committed_sweep_file_is_well_formed():
    path = os.path.join(ROOT, "profiles", "r02_sweep.jsonl")
    if not os.path.exists(path):
        pytest.skip("no committed sweep yet")
    lines = [json.loads(l) for l in open(path) if l.startswith("{")]
    n = lines[0]["rSize"]
    _validate(lines, n, 5, [1 << e for e in range(n.bit_length())], with_cpu=True)


@pytest.mark.gpu
def test_sweep_harness_smoke_on_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep.py"), "--log2n", "20", "--repeats", "3", "--max-log2w", "11"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    _validate(lines, 1 << 20, 3, [1 << e for e in range(12)], with_cpu=True)
    first = next(l for l in lines if l["algo"] == "atomic" and l["device"] == "hip")
    assert list(first)[:9] == ["algo", "rSize", "probeLength", "hashBuildTimeInMicroseconds", "repeats", "runs_us", "conflicts",
                               "totalMatches", "inputSum"]
    auto = {l["shuffleRange"]: l["algoUsed"] for l in lines if l["algo"] == "auto"}
    assert auto[1] == "atomic" and auto[16] == "atomic"                          # locality: the table join


def test_motivation_pins_equal_the_reference_logs(golden_dir):
    """--protocol motivation asserts, at 2^27: PRO's "Results" = the value of every block of motivation_log* (the closed
    form at NUM_RADIX_BITS 14), and for the build-only runs the sums AtomicsVsHTMVsNoCC_log1:1-6 hold."""
    logs = json.load(open(os.path.join(golden_dir, "reference_logs.json")))
    n = 1 << 27
    tri = n * (n + 1) // 2
    assert sweep.expected("prj", n, 14)["results"] == 549688705024
    seen = 0
    for c in logs["cases"]:
        if c["script"] != "experiments/AtomicsVsHTMVsNoCC.sh":
            continue
        seen += 1
        assert c["inputSum"] == tri and c.get("conflicts", c.get("conflictCount")) == 0
        assert c["outputSum"] == (tri - n if c["algo"] == "nocc" else tri)
    assert seen >= 6


def test_mc_pro_leg_runs_the_reference_binary():
    """The CPU radix leg of the motivation protocol: oracle/_ref/mchashjoins (built from the reference's mc/src where it
    lies) with motivation.sh's own flags, at a small size; its "Results" equals the closed form the harness asserts."""
    if not os.path.exists(sweep.MC):
        pytest.skip("oracle/_ref/mchashjoins not built (the reference's sources are not on this machine)")
    n = 1 << 18
    rows = sweep.mc_pro_lines(n, 16, 2, 2)
    assert len(rows) == 2 and all(r["us"] > 0 for r in rows)
    assert all(r["results"] == sweep.expected("prj", n, 14)["results"] for r in rows)


@pytest.mark.gpu
def test_motivation_protocol_smoke_on_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep.py"), "--protocol", "motivation", "--log2n", "20",
                        "--repeats", "3", "--max-log2w", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    mot = [l for l in lines if l["protocol"] == "motivation.sh"]
    avh = [l for l in lines if l["protocol"] == "AtomicsVsHTMVsNoCC.sh"]
    for w in (1 << e for e in range(6)):
        got = {(l["algo"], l["device"]) for l in mot if l["shuffleRange"] == w}
        want = {("atomic", "hip"), ("htm", "hip"), ("prj", "hip"), ("nocc", "cpu"), ("atomic", "cpu")}
        if os.path.exists(sweep.MC):
            want.add(("PRO", "cpu"))
        assert got == want, (w, got)
    assert {(l["dataDistr"], l["algo"], l["device"]) for l in avh} >= {(d, a, "hip") for d in ("sorted", "shuffle") for a in ("atomic", "htm", "prj")}
    assert all("totalMatches" not in l for l in lines)               # build only, as the reference's ENABLE_PROBE 0 binaries print
