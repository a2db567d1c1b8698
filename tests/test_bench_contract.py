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


def test_sharded_bench_key_ranges_match_the_range_split():
    """bench.py --gpus N: every rank's keys must lie in its own cut of the key space, also when tuples outnumber 32-bit
    keys (8 x 2^30), and the range split's digit of every such key must be the rank -- the in-place join depends on it."""
    import numpy as np
    from htm_hashjoin_amd.sharded import rank_key_range, squeeze_into_range
    for world, log2n in ((2, 30), (4, 30), (8, 30), (8, 27), (4, 29), (64, 30)):
        n = 1 << log2n
        total = min(world * n, (1 << 32) - 1)
        digit = (total - 1).bit_length() - (world.bit_length() - 1)
        prev_hi = 0
        for rank in range(world):
            lo, width = rank_key_range(world, n, rank)
            assert lo == prev_hi and 1 <= width <= n
            prev_hi = lo + width
            v = np.array([1, 2, n // 2, n - 1, n], dtype=np.uint64)
            k = squeeze_into_range(v, n, lo, width, np)
            assert int(k.min()) == lo + 1 and int(k.max()) == lo + width and int(k.max()) <= (1 << 32) - 1
            assert np.all(np.diff(k.astype(np.int64)) >= 0)
            assert np.all(((k - np.uint64(1)) >> np.uint64(digit)) == rank), (world, log2n, rank)
        assert prev_hi == total
