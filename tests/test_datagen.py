"""The product's DataGen (hj_generate_data: own restatement of glibc's rand()) against
the oracle's, which calls libc srand(0)/rand() exactly as include/DataGen.hpp does."""
import numpy as np
import pytest

import htm_hashjoin_amd as hj
from oracle import oracle

CASES = [("uniform", 16), ("uniform", 3), ("random", 16), ("sorted", 16), ("shuffle", 16),
         ("local_shuffle", 1), ("local_shuffle", 16), ("local_shuffle", 1024)]


@pytest.mark.parametrize("dist,window", CASES)
@pytest.mark.parametrize("n", [1, 2, 64, 1000, 1 << 12, 1 << 17])
def test_matches_libc_stream(dist, window, n):
    distinct = n if dist != "uniform" else 1 << max(n - 1, 1).bit_length()
    a = hj.generate_data(dist, n, distinct, window)
    b = oracle.generate_data(dist, n, distinct, window)
    assert np.array_equal(a, b)


def test_large_uniform_uses_threaded_sort():
    n = 1 << 21
    assert np.array_equal(hj.generate_data("uniform", n, n, 16), oracle.generate_data("uniform", n, n, 16))


def test_zipf_matches_genzipf_restatement():
    # extension: the reference's DataGen zipf branch is an empty stub; both sides follow mc/src/genzipf.c
    n, alpha = 50000, 4096
    a = hj.generate_data("zipf", n, alpha, 16, zipf_theta=0.9)
    b = oracle.generate_zipf(n, alpha, 0.9, 0)
    assert np.array_equal(a, b)
    assert a.min() >= 1 and a.max() <= alpha
    # skew sanity: the most frequent key carries far more than 1/alpha of the mass
    assert np.bincount(a.astype(np.int64)).max() > 20 * n / alpha


def test_unknown_distribution_is_an_error():
    with pytest.raises(hj.HashJoinError):
        hj.generate_data("zipfian", 16)
