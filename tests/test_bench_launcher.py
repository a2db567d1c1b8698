"""`bench.py --gpus N` must give an N-rank run (round-2 VERDICT, row e'): without a launcher in the environment it starts
the N ranks itself -- fresh processes, before the parent touches the GPU -- and relays rank 0's one JSON line. Driven here
at N = 2 over gloo with the checker-backed stand-in engine (HJ_BENCH_TEST_ENGINE, tests only): the launcher, the
rendezvous, the pre-flight exchange self-check and the line's contract are the product's own code."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    env.update(extra)
    return env


def test_gpus_2_starts_two_ranks_and_relays_one_line():
    from oracle import oracle
    n = 1 << 12
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--log2n", "12", "--steps", "2", "--warmup", "1"],
                       env=_env(HJ_BENCH_TEST_ENGINE="oracle_shard_engine:OracleShardEngine"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                # ONE JSON line on stdout, nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_world"] == 2 and line["launcher"] == "self-spawned"
    assert line["backend"] == "gloo" and "TEST STAND-IN" in line["engine"]
    assert line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    ex = line["exchange"]
    assert ex["split"] == "low key bits" and ex["sent_r"] + ex["sent_s"] > 0
    # the pre-flight exchange ran both forms at both sizes, every element checked, and the choice follows from it
    sc = line["a2a_selfcheck"]
    assert set(sc) == {"1MiB_per_peer", "step_size"} and sc["step_size"]["keys_per_peer"] == n // 2
    for size in sc.values():
        for form in ("a2a", "p2p"):
            assert size[form]["ok"] is True and size[form]["ms"] > 0
    assert line["exchange_chosen"] in ("a2a", "p2p")
    assert ex["form"] == {"a2a": "all_to_all_single", "p2p": "batch_isend_irecv"}[line["exchange_chosen"]]
    # totals = the sharded reference over the two ranks' pieces
    from htm_hashjoin_amd.sharded import rank_key_range, squeeze_into_range
    Rs, Ss = [], []
    for g in range(2):
        lo, width = rank_key_range(2, n, g)
        Rs.append(squeeze_into_range(oracle.generate_data("uniform", n, n, 16), n, lo, width, np))
        Ss.append(squeeze_into_range(np.arange(1, n + 1, dtype=np.uint64), n, lo, width, np))
    want = oracle.sharded_reference(np.concatenate(Rs), np.concatenate(Ss), 2)
    for k in ("conflicts", "totalMatches", "inputSum"):
        assert line["result"][k] == want[k], k


def test_a_failing_rank_fails_the_launch():
    """A rank that dies takes the launch down with a non-zero exit and no JSON line (here: an engine that does not exist)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--log2n", "10", "--steps", "1", "--warmup", "0"],
                       env=_env(HJ_BENCH_TEST_ENGINE="oracle_shard_engine:NoSuchEngine"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank" in r.stderr and "exited with" in r.stderr


def test_gpus_disagreeing_with_world_size_is_an_error():
    """Under an external launcher `--gpus` must equal WORLD_SIZE: never a silent n_gpus = 1 line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0", "--log2n", "10"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "disagrees with WORLD_SIZE" in r.stderr


def test_more_gpus_than_devices_is_refused():
    import torch
    have = torch.cuda.device_count()
    if have >= 64:
        pytest.skip("a box with 64 devices")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0", "--log2n", "10"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "HIP device(s) visible" in r.stderr
