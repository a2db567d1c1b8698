"""ctypes binding of libhtmjoin_hip.so (C ABI: include/htm_hashjoin.h).

The library is the product; this module only declares its signatures. If the
shared object has not been built the import fails loudly -- there is no Python
or CPU fallback for any operator.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libhtmjoin_hip.so")
# Development tools (tools/run_variants.sh and friends) time A/B builds of the library made by tools/mk_variant.sh: they
# name the variant here instead of copying it over the product library (round-2 ADVICE: an interrupted script left a
# variant in place). Nothing but those tools sets it; the path is printed so that a measurement can never pass a
# variant off as the product.
_DEV = os.environ.get("HJ_DEV_LIB_VARIANT")
if _DEV:
    LIB_PATH = os.path.abspath(_DEV)
    import sys as _sys
    print(f"[htm_hashjoin_amd] DEVELOPMENT VARIANT LIBRARY: {LIB_PATH}", file=_sys.stderr)

HJ_OK = 0
HJ_ERR_INVALID = -1
HJ_ERR_NO_DEVICE = -2
HJ_ERR_HIP = -3
HJ_ERR_OOM = -4
HJ_ERR_KEY_RANGE = -5
HJ_ERR_UNKNOWN_ALGO = -6
HJ_ERR_STATE = -7

HJ_ALGO_NOCC, HJ_ALGO_ATOMIC, HJ_ALGO_HTM, HJ_ALGO_PRJ, HJ_ALGO_AUTO = 0, 1, 2, 3, 4
ALGO_IDS = {"nocc": HJ_ALGO_NOCC, "atomic": HJ_ALGO_ATOMIC, "htm": HJ_ALGO_HTM, "prj": HJ_ALGO_PRJ,
            "auto": HJ_ALGO_AUTO}
ALGO_NAMES = {v: k for k, v in ALGO_IDS.items()}


class hj_params(C.Structure):
    _fields_ = [
        ("algo", C.c_uint32),
        ("scaleOutput", C.c_uint32),
        ("numPartitions", C.c_uint32),
        ("probeLength", C.c_uint32),
        ("transactionSize", C.c_uint32),
        ("radixBits", C.c_uint32),
        ("buildVariant", C.c_uint32),
        ("prjMode", C.c_uint32),
        ("reserved", C.c_uint32 * 4),
    ]


class hj_result(C.Structure):
    _fields_ = (
        [(n, C.c_uint64) for n in (
            "rSize", "sSize", "tableSize", "conflicts", "totalMatches", "inputSum",
            "tableSumHalf", "tableSumFull", "conflictSum", "outputSum", "prjChecksum",
            "prjPartitions")]
        + [("radixBits", C.c_uint32), ("buildVariant", C.c_uint32)]
        + [(n, C.c_double) for n in (
            "clear_us", "build_us", "probe_us", "partition_us", "join_us", "total_us", "h2d_us")]
        + [("buildDeferred", C.c_uint64), ("buildPhaseA_us", C.c_double), ("algoUsed", C.c_uint32),
           ("prjPath", C.c_uint32), ("foreignTuples", C.c_uint64), ("prjScatterPass1R_us", C.c_double),
           ("htmBuckets", C.c_uint64), ("htmOverflowBuckets", C.c_uint64), ("htmOverflowSum", C.c_uint64),
           ("compactFallback", C.c_uint64)]
    )

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}
        d["algoUsed"] = ALGO_NAMES.get(d["algoUsed"], d["algoUsed"])
        return d


def _declare(lib):
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    P = C.POINTER
    sig = {
        "hj_abi_version": ([], i32),
        "hj_device_count": ([P(i32)], i32),
        "hj_create": ([i32, P(vp)], i32),
        "hj_create_on_stream": ([i32, vp, P(vp)], i32),
        "hj_destroy": ([vp], None),
        "hj_strerror": ([i32], C.c_char_p),
        "hj_last_error": ([vp], C.c_char_p),
        "hj_synchronize": ([vp], i32),
        "hj_run": ([vp, P(hj_params), vp, u64, vp, u64, P(hj_result)], i32),
        "hj_reserve": ([vp, P(hj_params), u64, u64], i32),
        "hj_build_dev": ([vp, vp, u64, u64], i32),
        "hj_probe_dev": ([vp, vp, u64], i32),
        "hj_prj_join_dev": ([vp, vp, u64, vp, u64], i32),
        "hj_join_dev": ([vp, vp, u64, vp, u64], i32),
        "hj_checksums_dev": ([vp], i32),
        "hj_fetch_result": ([vp, P(hj_result)], i32),
        "hj_export_table": ([vp, vp, u64], i32),
        "hj_export_buckets": ([vp, vp, u64, vp, u64, P(u64)], i32),
        "hj_shard_histogram_dev": ([vp, vp, u64, u32, u32, vp], i32),
        "hj_shard_scatter_dev": ([vp, vp, u64, u32, u32, vp, vp], i32),
        "hj_build_keys_dev": ([vp, vp, u64, u32, u64], i32),
        "hj_probe_keys_dev": ([vp, vp, u64], i32),
        "hj_set_shard_check": ([vp, u32, u32, u32], i32),
        "hj_dev_alloc": ([vp, u64, P(vp)], i32),
        "hj_dev_free": ([vp, vp], i32),
        "hj_copy_h2d": ([vp, vp, vp, u64], i32),
        "hj_copy_d2h": ([vp, vp, vp, u64], i32),
        "hj_prj_workspace_info": ([u64, u64, u32, P(u64)], i32),
        "hj_prj_fragment_info": ([u64, u64, u32, u32, P(u64)], i32),
        "hj_zipf_open": ([vp, u64, C.c_double, C.c_uint], i32),
        "hj_zipf_next_dev": ([vp, u64, vp], i32),
        "hj_zipf_close": ([vp], i32),
        "hj_generate_relation": ([C.c_char_p, u64, u64, i32, C.c_double, C.c_uint, vp], i32),
        "hj_generate_data": ([C.c_char_p, u64, u64, i32, C.c_double, vp], i32),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.argtypes = args
        fn.restype = res
    return sig


def load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C htm-hashjoin_amd/csrc`. There is no fallback implementation."
        )
    lib = C.CDLL(LIB_PATH)
    lib._hj_signatures = _declare(lib)
    return lib


lib = load()
