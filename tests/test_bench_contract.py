"""bench.py's contract pieces that can be checked without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_gpu():
    import htm_hashjoin_amd as hj
    if hj.device_count() > 0:
        import pytest
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                        "--log2n", "10"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "no CPU fallback" in r.stderr


def test_bench_accepts_the_driver_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


def test_sharded_helpers():
    from htm_hashjoin_amd.sharded import _log2
    import pytest
    assert [_log2(n) for n in (1, 2, 4, 8)] == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        _log2(6)
