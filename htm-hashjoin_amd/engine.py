"""Host-side mirror of the reference's operator interface.

    reference (C++)                                   here
    ------------------------------------------------  ---------------------------------
    generate_data(dist, n, distinct, window)          generate_data(...)      DataGen.hpp:26
    NoCCHashBuild(relR,rSize,relS,sSize,scale,P,pl)    NoCCHashBuild(...)      NoCCHashBuild.hpp:13
    AtomicHashBuild(... same ...)                      AtomicHashBuild(...)    AtomicHashBuild.hpp:14
    HTMHashBuild(relR,rSize,relS,sSize,tSize,...)      HTMHashBuild(...)       HTMHashBuild.hpp:54
    PRO(relR, relS, nthreads)                          PRO(...)                mc/src/parallel_radix_join.c:1305

The reference functions return void and print one JSON line; these return the
same fields as a dict. Every operator runs on the GPU through the C ABI; without
a gfx950 device they raise HashJoinError(HJ_ERR_NO_DEVICE).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, hj_params, hj_result


# struct Bucket, HTMHashBuild.hpp:41-45 (32 bytes)
BUCKET_DTYPE = np.dtype([("tuples", np.uint64, 3), ("count", np.uint32), ("nextIndex", np.uint32)])


class HashJoinError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib.hj_strerror(status).decode()
        super().__init__(f"{msg} [{status}]" + (f": {detail}" if detail else ""))


def device_count():
    n = C.c_int(0)
    lib.hj_device_count(C.byref(n))
    return n.value


def generate_data(dist, size_in_tuples, distinct_keys=None, local_shuffle_range=16, zipf_theta=0.9):
    """include/DataGen.hpp:26 -- returns a numpy uint64 array of size_in_tuples tuples."""
    if distinct_keys is None:
        distinct_keys = size_in_tuples
    out = np.empty(size_in_tuples, dtype=np.uint64)
    rc = lib.hj_generate_data(dist.encode(), size_in_tuples, distinct_keys, int(local_shuffle_range),
                              float(zipf_theta), out.ctypes.data)
    if rc != _lib.HJ_OK:
        raise HashJoinError(rc, f"Unknown distribution {dist!r}")
    return out


def generate_relation(kind, num_tuples, maxid=None, local_shuffle_range=0, zipf_param=0.0, seed=12345):
    """mc/src/generator.c: kind = pk | pk_lshuffle | fk | nonunique | zipf (create_relation_*), srand(seed) first
    (mc seeds R with 12345 and S with 54321, mc/src/main.c:337-338)."""
    out = np.empty(num_tuples, dtype=np.uint64)
    rc = lib.hj_generate_relation(kind.encode(), num_tuples, num_tuples if maxid is None else maxid,
                                  int(local_shuffle_range), float(zipf_param), int(seed), out.ctypes.data)
    if rc != _lib.HJ_OK:
        raise HashJoinError(rc, f"hj_generate_relation({kind!r})")
    return out


def _params(algo, scaleOutput=2, numPartitions=64, probeLength=4, transactionSize=16, radixBits=0,
            buildVariant=0, prjMode=0):
    p = hj_params()
    p.algo = _lib.ALGO_IDS[algo]
    p.scaleOutput, p.numPartitions, p.probeLength = scaleOutput, numPartitions, probeLength
    p.transactionSize, p.radixBits, p.buildVariant = transactionSize, radixBits, buildVariant
    p.prjMode = prjMode
    return p


SHARD_ONE_BASED = 0x100


class HashJoinContext:
    """One engine context bound to one GPU (hj_ctx). ``stream`` is a raw hipStream_t
    handle (e.g. torch.cuda.current_stream().cuda_stream); None = private stream."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        if stream is None:
            rc = lib.hj_create(device, C.byref(self._h))
        else:
            rc = lib.hj_create_on_stream(device, C.c_void_p(stream), C.byref(self._h))
        if rc != _lib.HJ_OK:
            self._h = None
            raise HashJoinError(rc)
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib.hj_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != _lib.HJ_OK:
            raise HashJoinError(rc, lib.hj_last_error(self._h).decode())

    # ---- one-shot, host buffers -------------------------------------------
    def run(self, algo, relR, relS=None, **kw):
        relR = np.ascontiguousarray(relR, dtype=np.uint64)
        if relS is not None:
            relS = np.ascontiguousarray(relS, dtype=np.uint64)
        p = _params(algo, **kw)
        r = hj_result()
        self._check(lib.hj_run(self._h, C.byref(p), relR.ctypes.data, relR.size,
                               relS.ctypes.data if relS is not None else None,
                               relS.size if relS is not None else 0, C.byref(r)))
        return r.as_dict()

    # ---- split API, device pointers ---------------------------------------
    def reserve(self, algo, rSize, sSize, **kw):
        p = _params(algo, **kw)
        self._check(lib.hj_reserve(self._h, C.byref(p), rSize, sSize))

    def build(self, dR_ptr, rSize, idx_base=0):
        self._check(lib.hj_build_dev(self._h, C.c_void_p(dR_ptr), rSize, idx_base))

    def probe(self, dS_ptr, sSize):
        self._check(lib.hj_probe_dev(self._h, C.c_void_p(dS_ptr), sSize))

    def prj_join(self, dR_ptr, rSize, dS_ptr, sSize):
        self._check(lib.hj_prj_join_dev(self._h, C.c_void_p(dR_ptr), rSize,
                                        C.c_void_p(dS_ptr) if dS_ptr else None, sSize))

    def join(self, dR_ptr, rSize, dS_ptr, sSize):
        """Build + probe by the reserved algo; "auto" samples R for locality and picks atomic or prj."""
        self._check(lib.hj_join_dev(self._h, C.c_void_p(dR_ptr), rSize,
                                    C.c_void_p(dS_ptr) if dS_ptr else None, sSize))

    def checksums(self):
        self._check(lib.hj_checksums_dev(self._h))

    def fetch(self):
        r = hj_result()
        self._check(lib.hj_fetch_result(self._h, C.byref(r)))
        return r.as_dict()

    def synchronize(self):
        self._check(lib.hj_synchronize(self._h))

    def export_table(self, tableSize):
        out = np.empty(tableSize, dtype=np.uint64)
        self._check(lib.hj_export_table(self._h, out.ctypes.data, tableSize))
        return out

    def export_buckets(self, numBuckets):
        """HTM table as the reference's Bucket structs: (buckets[numBuckets], overflows[1 + used]); overflows[0] is unused,
        nextIndex is 1-based (HTMHashBuild.hpp:41-45, 231-279)."""
        used = C.c_uint64(0)
        buckets = np.zeros(numBuckets, dtype=BUCKET_DTYPE)
        # first call learns the number of overflow buckets (no overflow buffer handed over), second one copies them
        rc = lib.hj_export_buckets(self._h, buckets.ctypes.data, numBuckets, None, 0, C.byref(used))
        if rc != _lib.HJ_OK and used.value == 0:
            self._check(rc)
        overflows = np.zeros(used.value + 1, dtype=BUCKET_DTYPE)
        self._check(lib.hj_export_buckets(self._h, buckets.ctypes.data, numBuckets, overflows.ctypes.data,
                                          overflows.size, C.byref(used)))
        return buckets, overflows

    # ---- raw device memory (hosts without a HIP runtime of their own) -------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._check(lib.hj_dev_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        self._check(lib.hj_dev_free(self._h, C.c_void_p(ptr)))

    def copy_h2d(self, dst_ptr, src_np):
        src_np = np.ascontiguousarray(src_np)
        self._check(lib.hj_copy_h2d(self._h, C.c_void_p(dst_ptr), src_np.ctypes.data, src_np.nbytes))

    def copy_d2h(self, dst_np, src_ptr):
        self._check(lib.hj_copy_d2h(self._h, dst_np.ctypes.data, C.c_void_p(src_ptr), dst_np.nbytes))

    # ---- streaming Zipf generator (probe sides that do not fit one host buffer) -------------------------------
    def zipf_open(self, alphabet_size, theta, seed=0):
        self._check(lib.hj_zipf_open(self._h, alphabet_size, float(theta), int(seed)))

    def zipf_next(self, n, d_out):
        """the next n draws of the stream, as 8-byte tuples at device pointer d_out"""
        self._check(lib.hj_zipf_next_dev(self._h, n, C.c_void_p(d_out)))

    def zipf_close(self):
        self._check(lib.hj_zipf_close(self._h))

    def shard_histogram(self, d_in, n, n_shards, d_counts, mode=0):
        """mode = bit position of the radix digit (0 = low key bits), | SHARD_ONE_BASED for (key - 1)"""
        self._check(lib.hj_shard_histogram_dev(self._h, C.c_void_p(d_in), n, n_shards, mode, C.c_void_p(d_counts)))

    def shard_scatter(self, d_in, n, n_shards, d_counts, d_out_keys, mode=0):
        """tuples in, bare 32-bit keys out, grouped by destination in input order"""
        self._check(lib.hj_shard_scatter_dev(self._h, C.c_void_p(d_in), n, n_shards, mode, C.c_void_p(d_counts),
                                             C.c_void_p(d_out_keys)))

    def build_keys(self, d_keys, n, home_shift, table_size):
        self._check(lib.hj_build_keys_dev(self._h, C.c_void_p(d_keys), n, home_shift, table_size))

    def probe_keys(self, d_keys, n):
        self._check(lib.hj_probe_keys_dev(self._h, C.c_void_p(d_keys), n))

    def set_shard_check(self, n_shards, mode=0, shard_id=0):
        """later builds/probes also count tuples whose destination is another shard (result["foreignTuples"]); 0 = off"""
        self._check(lib.hj_set_shard_check(self._h, n_shards, mode, shard_id))


def _operator(algo, relR, rSize, relS, sSize, device, **kw):
    relR = np.asarray(relR, dtype=np.uint64)[:rSize]
    if relS is not None:
        relS = np.asarray(relS, dtype=np.uint64)[:sSize]
    with HashJoinContext(device) as ctx:
        r = ctx.run(algo, relR, relS, **kw)
    out = {"algo": algo, "rSize": r["rSize"], "probeLength": kw.get("probeLength", 4),
           "hashBuildTimeInMicroseconds": int(r["total_us"]), "conflicts": r["conflicts"]}
    if relS is not None:
        out["totalMatches"] = r["totalMatches"]
    out["inputSum"] = r["inputSum"]
    out["outputSum"] = r["outputSum"]
    out["detail"] = r
    return out


def NoCCHashBuild(relR, rSize, relS=None, sSize=0, scaleOutput=2, numPartitions=64, probeLength=4, device=0):
    """NoCCHashBuild.hpp:13-19. outputSum keeps the [0, rSize) quirk of :94."""
    return _operator("nocc", relR, rSize, relS, sSize, device, scaleOutput=scaleOutput,
                     numPartitions=numPartitions, probeLength=probeLength)


def AtomicHashBuild(relR, rSize, relS=None, sSize=0, scaleOutput=2, numPartitions=64, probeLength=4, device=0):
    """AtomicHashBuild.hpp:14-20."""
    return _operator("atomic", relR, rSize, relS, sSize, device, scaleOutput=scaleOutput,
                     numPartitions=numPartitions, probeLength=probeLength)


def HTMHashBuild(relR, rSize, relS=None, sSize=0, transactionSize=16, scaleOutput=2, numPartitions=64,
                 probeLength=4, device=0):
    """HTMHashBuild.hpp:54-60: the bucketised table (three tuples per 32-byte bucket, bucket = (key/3) & mask, conflicts
    chained into overflow buckets). The TSX transaction groups are replaced outright by the index-priority fill, so
    there are no aborted transactions to report (failedTransactions = 0); transactionSize is accepted and echoed. Returns
    the reference's JSON fields (:417-452) in its order."""
    relR = np.asarray(relR, dtype=np.uint64)[:rSize]
    if relS is not None:
        relS = np.asarray(relS, dtype=np.uint64)[:sSize]
    with HashJoinContext(device) as ctx:
        r = ctx.run("htm", relR, relS, scaleOutput=scaleOutput, numPartitions=numPartitions, probeLength=probeLength,
                    transactionSize=transactionSize)
    out = {"algo": "htm", "rSize": r["rSize"], "transactionSize": transactionSize, "probeLength": probeLength,
           "hashBuildTimeInMicroseconds": int(r["total_us"]), "firstRoundTime": 0, "firstRoundFailureFraction": 0.0,
           "conflictCount": r["conflicts"], "failedTransactions": 0, "failedTransactionPercentage": 0.0,
           "totalFailedPercentage": r["conflicts"] / max(r["rSize"], 1)}
    if relS is not None:
        out["totalMatches"] = r["totalMatches"]
    out["inputSum"] = r["inputSum"]
    out["outputSum"] = r["outputSum"]
    out["detail"] = r
    return out


def PRO(relR, relS=None, nthreads=0, radixBits=0, device=0):
    """mc/src/parallel_radix_join.c:1305 (algos[] entry, mc/src/main.c:292-301).
    Returns the join cardinality and the fork's printed checksum ("Results")."""
    relR = np.asarray(relR, dtype=np.uint64)
    if relS is not None:
        relS = np.asarray(relS, dtype=np.uint64)
    with HashJoinContext(device) as ctx:
        r = ctx.run("prj", relR, relS, radixBits=radixBits)
    return {"algo": "PRO", "matches": r["totalMatches"], "results": r["prjChecksum"],
            "radixBits": r["radixBits"], "detail": r}
