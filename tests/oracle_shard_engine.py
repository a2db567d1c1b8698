"""CPU stand-in for HipShardEngine (TEST ONLY): same interface, numpy + the CPU oracle.

Used by tests/test_sharded_gloo.py (the exchange logic of htm_hashjoin_amd/sharded.py under gloo) and, through
bench.py's HJ_BENCH_TEST_ENGINE hook, by tests/test_bench_launcher.py (bench.py --gpus N starting its own ranks).
Nothing in the product imports this file."""
import numpy as np
import torch

from oracle import oracle


class OracleShardEngine:
    """CPU stand-in for HipShardEngine: same interface, numpy + the CPU oracle (TEST ONLY)."""

    def __init__(self):
        self._res = None

    def empty_keys(self, n):
        return torch.empty(int(n), dtype=torch.int32)

    @staticmethod
    def _dest(k, n_shards, mode):
        """hj_shard_histogram_dev: ((key - b) >> d) & (nShards - 1), d = mode & 0xFF, b = 1 if HJ_SHARD_ONE_BASED"""
        k32 = k.astype(np.uint32) - np.uint32(1 if mode & 0x100 else 0)
        return ((k32 >> np.uint32(mode & 0xFF)) & np.uint32(n_shards - 1)).astype(np.int64)

    def histogram(self, t, n_shards, mode=0):
        k = t.numpy().view(np.uint64)
        return torch.from_numpy(np.bincount(self._dest(k, n_shards, mode), minlength=n_shards)).to(torch.int64)

    def scatter(self, t, n_shards, counts, mode=0):
        """grouped by destination, input order kept inside each (what hj_shard_scatter_dev guarantees), keys only"""
        k = t.numpy().view(np.uint64)
        order = np.argsort(self._dest(k, n_shards, mode), kind="stable")
        return torch.from_numpy(k[order].astype(np.uint32).view(np.int32).copy())

    def max_key(self, t):
        return int(t.max().item()) if t.numel() else 0

    def reserve(self, table_size, max_r, max_s):
        pass

    def build(self, r_keys, home_shift, table_size):
        # position in the receive buffer = insertion order
        self._built = (r_keys.numpy().view(np.uint32).astype(np.uint64), home_shift, table_size)

    _check = None

    def set_check(self, n_shards, mode=0, shard_id=0):
        self._check = (n_shards, mode, shard_id) if n_shards else None

    def _count_foreign(self, k):
        if self._check is None:
            return 0
        n_shards, mode, shard_id = self._check
        return int((self._dest(k, n_shards, mode) != shard_id).sum())

    def build_tuples(self, r_tuples):
        k = r_tuples.numpy().view(np.uint64).copy()
        self._foreign = self._count_foreign(k)
        self._built = (k, 0, 2 * r_tuples.numel())

    def probe_tuples(self, s_tuples):
        keys, home_shift, table_size = self._built
        k = s_tuples.numpy().view(np.uint64).copy()
        self._foreign += self._count_foreign(k)
        self._res = oracle.build_probe_seq_ts(keys, k, table_size, home_shift)

    def foreign(self):
        return self._foreign

    def probe(self, s_keys):
        keys, home_shift, table_size = self._built
        s = s_keys.numpy().view(np.uint32).astype(np.uint64)
        self._res = oracle.build_probe_seq_ts(keys, s, table_size, home_shift)

    def finish(self):
        r = dict(self._res)
        r.update(buildVariant=0, buildDeferred=0, build_us=0.0, probe_us=0.0, clear_us=0.0, buildPhaseA_us=0.0)
        return r

    def sync(self):
        pass
