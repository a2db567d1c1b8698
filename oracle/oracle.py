"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY. Nothing under htm-hashjoin_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
REF_MCHASHJOINS = os.path.join(_HERE, "_ref", "mchashjoins")


def build():
    subprocess.check_call(["make", "-C", _HERE, "--no-print-directory"], stdout=subprocess.DEVNULL)


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.orc_generate_data.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    lib.orc_generate_zipf.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_uint, C.c_void_p]
    lib.orc_generate_relation.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_uint, C.c_void_p]
    lib.orc_build_probe_seq.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.POINTER(OrcResult), C.c_void_p]
    lib.orc_build_probe_seq_ts.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64,
                                           C.c_uint32, C.POINTER(OrcResult), C.c_void_p]
    lib.orc_build_probe_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_int, C.c_int, C.POINTER(OrcResult)]
    lib.orc_build_probe_mt_ex.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                          C.c_int, C.c_int, C.c_int, C.POINTER(OrcResult)]
    lib.orc_prj_join.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                 C.POINTER(OrcPrjResult)]
    lib.orc_htm_build_probe_seq.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                            C.POINTER(OrcHtmResult), C.c_void_p, C.c_void_p]
    lib.orc_true_cardinality.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    lib.orc_true_cardinality.restype = C.c_uint64
    return lib


class OrcResult(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rSize", "sSize", "tableSize", "conflicts", "totalMatches", "inputSum", "tableSumHalf",
        "tableSumFull", "conflictSum", "outputSumNocc", "outputSumAtomic")] + [
        ("build_us", C.c_double), ("probe_us", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OrcHtmResult(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rSize", "sSize", "numBuckets", "conflictCount", "conflictSum", "overflowBuckets", "totalMatches", "inputSum",
        "bucketSum", "overflowSum", "outputSum", "outputSumAsWritten")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# Bucket, HTMHashBuild.hpp:41-45 (32 bytes)
BUCKET_DTYPE = np.dtype([("tuples", np.uint64, 3), ("count", np.uint32), ("nextIndex", np.uint32)])


class OrcPrjResult(C.Structure):
    _fields_ = [("matches", C.c_uint64), ("checksum", C.c_uint64), ("partitions", C.c_uint64),
                ("part_us", C.c_double), ("join_us", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = _load()


def generate_data(dist, n, distinct=None, window=16):
    out = np.empty(n, dtype=np.uint64)
    rc = _lib.orc_generate_data(dist.encode(), n, n if distinct is None else distinct, window, out.ctypes.data)
    if rc != 0:
        raise ValueError(f"unknown distribution {dist}")
    return out


def generate_zipf(n, alphabet, theta, seed):
    out = np.empty(n, dtype=np.uint64)
    _lib.orc_generate_zipf(n, alphabet, theta, seed, out.ctypes.data)
    return out


def generate_relation(kind, n, maxid=None, window=0, theta=0.0, seed=12345):
    """mc/src/generator.c relations: kind in pk, pk_lshuffle, fk, nonunique, zipf (orc_generate_relation)."""
    out = np.empty(n, dtype=np.uint64)
    rc = _lib.orc_generate_relation(kind.encode(), n, n if maxid is None else maxid, window, theta, seed, out.ctypes.data)
    if rc != 0:
        raise ValueError(f"unknown relation kind {kind}")
    return out


def relS_for(dist, relR, n=None):
    """main.cpp:91-97: S is 'sorted' 1..N unless dist == 'random' (then a copy of R)."""
    n = relR.size if n is None else n
    return relR.copy() if dist == "random" else generate_data("sorted", n)


def build_probe_seq(relR, relS=None, probe_length=4, want_table=False):
    relR = np.ascontiguousarray(relR, dtype=np.uint64)
    res = OrcResult()
    table = np.empty(2 * relR.size, dtype=np.uint64) if want_table else None
    s_ptr, s_n = (None, 0)
    if relS is not None:
        relS = np.ascontiguousarray(relS, dtype=np.uint64)
        s_ptr, s_n = relS.ctypes.data, relS.size
    rc = _lib.orc_build_probe_seq(relR.ctypes.data, relR.size, s_ptr, s_n, probe_length, C.byref(res),
                                  table.ctypes.data if want_table else None)
    assert rc == 0
    d = res.as_dict()
    if want_table:
        d["table"] = table
    return d


def build_probe_seq_ts(relR, relS, table_size, home_shift=0, probe_length=4, want_table=False):
    """Sequential build+probe into a table of table_size slots (sharded semantics)."""
    relR = np.ascontiguousarray(relR, dtype=np.uint64)
    res = OrcResult()
    table = np.empty(table_size, dtype=np.uint64) if want_table else None
    s_ptr, s_n = (None, 0)
    if relS is not None:
        relS = np.ascontiguousarray(relS, dtype=np.uint64)
        s_ptr, s_n = relS.ctypes.data, relS.size
    rc = _lib.orc_build_probe_seq_ts(relR.ctypes.data, relR.size, s_ptr, s_n, probe_length, table_size, home_shift,
                                     C.byref(res), table.ctypes.data if want_table else None)
    assert rc == 0
    d = res.as_dict()
    if want_table:
        d["table"] = table
    return d


def sharded_reference(relR, relS, n_shards, probe_length=4, digit_shift=0, one_based=False):
    """What the radix-sharded join computes: shard g takes the tuples with ((key - b) >> digit_shift) & (G-1) == g
    (b = 1 if one_based) in global input order and runs the sequential build+probe into a table of 2*|R|/G slots;
    counters are summed over shards. digit_shift = 0 (low key bits): home slot (key >> log2 G) & mask, the shard bits
    carry no information inside a shard. digit_shift > 0 (high bits, a range split): home slot key & mask, as in the
    single table."""
    strip = (n_shards - 1).bit_length() if digit_shift == 0 else 0
    b = np.uint64(1 if one_based else 0)
    table_size = 2 * relR.size // n_shards
    tot = {"conflicts": 0, "totalMatches": 0, "inputSum": 0, "tableSumFull": 0, "conflictSum": 0}
    for g in range(n_shards):
        Rg = relR[(((relR - b) >> np.uint64(digit_shift)) & np.uint64(n_shards - 1)) == g]
        Sg = relS[(((relS - b) >> np.uint64(digit_shift)) & np.uint64(n_shards - 1)) == g]
        r = build_probe_seq_ts(Rg, Sg, table_size, strip, probe_length)
        for k in tot:
            tot[k] += r[k]
    return tot


def htm_build_probe_seq(relR, relS=None, num_partitions=64, want_buckets=False):
    """HTMHashBuild in sequential order (orc_htm_build_probe_seq). With want_buckets the dict also carries the primary
    buckets and the overflow buckets (index 0 unused) as BUCKET_DTYPE arrays."""
    relR = np.ascontiguousarray(relR, dtype=np.uint64)
    res = OrcHtmResult()
    s_ptr, s_n = (None, 0)
    if relS is not None:
        relS = np.ascontiguousarray(relS, dtype=np.uint64)
        s_ptr, s_n = relS.ctypes.data, relS.size
    nb = 1
    while nb < relR.size // 3 + 1:
        nb *= 2
    buckets = np.zeros(nb, dtype=BUCKET_DTYPE) if want_buckets else None
    overflows = np.zeros(relR.size + 1, dtype=BUCKET_DTYPE) if want_buckets else None
    rc = _lib.orc_htm_build_probe_seq(relR.ctypes.data, relR.size, s_ptr, s_n, num_partitions, C.byref(res),
                                      buckets.ctypes.data if want_buckets else None,
                                      overflows.ctypes.data if want_buckets else None)
    assert rc == 0
    d = res.as_dict()
    assert d["numBuckets"] == nb
    if want_buckets:
        d["buckets"] = buckets
        d["overflows"] = overflows[: d["overflowBuckets"] + 1]
    return d


def htm_chains(buckets, overflows):
    """Logical view of a bucket table: per primary bucket the tuples it holds, then those of its overflow chain in walk
    order (head = newest overflow bucket). Physical overflow indices may differ between two builds of the same table;
    this view may not. Returns (counts, flat tuple array, offsets)."""
    out, off = [], [0]
    for b in range(buckets.size):
        cur = buckets[b]
        while True:
            out.extend(int(x) for x in cur["tuples"][: cur["count"]])
            if cur["nextIndex"] == 0:
                break
            cur = overflows[cur["nextIndex"]]
        off.append(len(out))
    return np.array(out, dtype=np.uint64), np.array(off, dtype=np.int64)


def build_probe_mt(relR, relS, probe_length=4, num_partitions=64, nthreads=1, atomic=False, parallel_touch=False):
    res = OrcResult()
    rc = _lib.orc_build_probe_mt_ex(relR.ctypes.data, relR.size, relS.ctypes.data if relS is not None else None,
                                    relS.size if relS is not None else 0, probe_length, num_partitions, nthreads,
                                    1 if atomic else 0, 1 if parallel_touch else 0, C.byref(res))
    assert rc == 0
    return res.as_dict()


def prj_join(relR, relS=None, radix_bits=14):
    relR = np.ascontiguousarray(relR, dtype=np.uint64)
    res = OrcPrjResult()
    s_ptr, s_n = (None, 0)
    if relS is not None:
        relS = np.ascontiguousarray(relS, dtype=np.uint64)
        s_ptr, s_n = relS.ctypes.data, relS.size
    rc = _lib.orc_prj_join(relR.ctypes.data, relR.size, s_ptr, s_n, radix_bits, C.byref(res))
    assert rc == 0
    return res.as_dict()


def true_cardinality(relR, relS):
    relR = np.ascontiguousarray(relR, dtype=np.uint64)
    relS = np.ascontiguousarray(relS, dtype=np.uint64)
    return _lib.orc_true_cardinality(relR.ctypes.data, relR.size, relS.ctypes.data, relS.size)
