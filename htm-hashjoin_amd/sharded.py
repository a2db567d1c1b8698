"""Multi-GPU radix-sharded join (placeholder until the exchange path lands)."""


def bench_sharded(*args, **kwargs):
    raise NotImplementedError("sharded bench path not implemented yet")
