"""libhtmjoin_sharded.so -- the radix-sharded join behind the C ABI (include/htm_hashjoin_sharded.h): one process, one host
thread per GPU, RCCL linked directly. CPU part: the library loads, exports exactly what its header declares, and its
host-side plan arithmetic (send / receive layout, destination mode) equals what htm_hashjoin_amd/sharded.py computes --
whose exchange logic tests/test_sharded_gloo.py pins against the sharded reference. GPU part: at world 1 (all a one-GPU
box can run: RCCL refuses two ranks on one device) the sharded join is the single-GPU operator, through the library
and through `main --gpus 1 --split low`."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import htm_hashjoin_amd as hj
from htm_hashjoin_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "htm-hashjoin_amd", "lib", "libhtmjoin_sharded.so")
MAIN = os.path.join(ROOT, "htm-hashjoin_amd", "bin", "main")


class Stats(C.Structure):
    _fields_ = [("nRanks", C.c_uint32), ("mode", C.c_uint32), ("homeShift", C.c_uint32), ("reserved", C.c_uint32),
                ("keysMovedR", C.c_uint64), ("keysMovedS", C.c_uint64), ("maxMessageKeys", C.c_uint64), ("tableSizePerRank", C.c_uint64)]


def _lib_sharded():
    C.CDLL(_lib.LIB_PATH, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB)
    u64p = C.POINTER(C.c_uint64)
    lib.hj_sharded_plan.argtypes = [C.c_uint32, u64p, u64p, u64p, u64p, u64p, u64p]
    lib.hj_sharded_mode.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32)]
    lib.hj_sharded_mode.restype = C.c_uint32
    lib.hj_sharded_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    lib.hj_sharded_destroy.argtypes = [C.c_void_p]
    lib.hj_sharded_destroy.restype = None
    lib.hj_sharded_last_error.argtypes = [C.c_void_p]
    lib.hj_sharded_last_error.restype = C.c_char_p
    lib.hj_sharded_alloc.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.hj_sharded_free.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.hj_sharded_copy_h2d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.hj_sharded_join.argtypes = [C.c_void_p, C.POINTER(_lib.hj_params), C.c_uint32, C.c_uint64, C.c_uint64,
                                    C.POINTER(C.c_void_p), u64p, C.POINTER(C.c_void_p), u64p,
                                    C.POINTER(_lib.hj_result), C.POINTER(Stats)]
    return lib


def test_library_loads_and_exports_what_its_header_declares():
    text = open(os.path.join(ROOT, "include", "htm_hashjoin_sharded.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(hj_sharded_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 9
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("hj_")}
    assert exported == declared
    lib = _lib_sharded()
    for s in declared:
        assert hasattr(lib, s)
    # RCCL is a dependency of THIS library only: the single-GPU library does not pull it in
    assert "rccl" in subprocess.run(["ldd", LIB], capture_output=True, text=True).stdout
    assert "rccl" not in subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout


@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_plan_layout_is_sharded_py_layout(G):
    """send offsets = prefix of my counts by destination; receive offsets = prefix of the counts sent TO me by source rank
    (pieces in source-rank order: position in the receive buffer = global input order) -- ShardedJoin._exchange_async."""
    lib = _lib_sharded()
    rng = np.random.default_rng(G)
    counts = rng.integers(0, 1000, size=(G, G)).astype(np.uint64)
    counts[rng.integers(0, G), rng.integers(0, G)] = 0
    flat = np.ascontiguousarray(counts.reshape(-1))
    so = np.zeros(G * (G + 1), dtype=np.uint64); ro = np.zeros(G * (G + 1), dtype=np.uint64)
    tot = np.zeros(G, dtype=np.uint64)
    mx = C.c_uint64(); mv = C.c_uint64()
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))          # noqa: E731
    assert lib.hj_sharded_plan(G, p(flat), p(so), p(ro), p(tot), C.byref(mx), C.byref(mv)) == 0
    so, ro = so.reshape(G, G + 1), ro.reshape(G, G + 1)
    for g in range(G):
        assert so[g].tolist() == [0] + np.cumsum(counts[g]).tolist()
        assert ro[g].tolist() == [0] + np.cumsum(counts[:, g]).tolist()
        assert tot[g] == counts[:, g].sum()
    off_diag = counts.copy(); np.fill_diagonal(off_diag, 0)
    assert mv.value == off_diag.sum() and mx.value == off_diag.max()


def test_destination_mode_is_sharded_py_mode():
    """ShardedJoin._decide_split: low = digit 0 with the shard bits shifted out of the home slot; high = the top log2 G
    bits of (key - 1) over [1, maxKey], one-based, home shift 0; a key domain too small for a digit falls back to low."""
    lib = _lib_sharded()
    hs = C.c_uint32()
    for G in (1, 2, 4, 8, 64):
        gbits = G.bit_length() - 1
        assert lib.hj_sharded_mode(G, 0, 1 << 30, C.byref(hs)) == 0 and hs.value == gbits
        for mk in (1, 2, 1000, 1 << 14, (1 << 30), (1 << 32) - 1):
            digit = max((max(mk, 1) - 1).bit_length() - gbits, 0)
            want = (digit | 0x100, 0) if (digit > 0 and G > 1) else (0, gbits)
            assert (lib.hj_sharded_mode(G, 1, mk, C.byref(hs)), hs.value) == want, (G, mk)
    assert lib.hj_sharded_mode(4, 1, 1 << 14, C.byref(hs)) == (12 | 0x100)       # the gloo tests' case: keys 1..2^14 over 4 ranks


def test_create_fails_loudly_without_a_gpu():
    if hj.device_count() > 0:
        pytest.skip("GPU present")
    lib = _lib_sharded()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert lib.hj_sharded_create(devs, 2, C.byref(h)) == _lib.HJ_ERR_NO_DEVICE and not h.value
    assert lib.hj_sharded_create(devs, 3, C.byref(h)) == _lib.HJ_ERR_INVALID      # a power of two
    r = subprocess.run([MAIN, "--algo", "atomic", "--rSize", "1024", "--dataDistr", "uniform", "--gpus", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 2 and r.stdout == "" and "no gfx950" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("dist,window", [("uniform", 16), ("local_shuffle", 1024), ("shuffle", 16)])
def test_world_1_is_the_single_gpu_operator(dist, window):
    from oracle import oracle
    lib = _lib_sharded()
    n = 1 << 18
    R = oracle.generate_data(dist, n, n, window)
    S = oracle.relS_for(dist, R)
    want = oracle.build_probe_seq(R, S, 4)
    h = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert lib.hj_sharded_create(devs, 1, C.byref(h)) == 0
    try:
        dR, dS = C.c_void_p(), C.c_void_p()
        assert lib.hj_sharded_alloc(h, 0, (n + 2) * 8, C.byref(dR)) == 0 and lib.hj_sharded_alloc(h, 0, (S.size + 2) * 8, C.byref(dS)) == 0
        assert lib.hj_sharded_copy_h2d(h, 0, dR, R.ctypes.data, n * 8) == 0
        assert lib.hj_sharded_copy_h2d(h, 0, dS, S.ctypes.data, S.size * 8) == 0
        params = _lib.hj_params(algo=_lib.HJ_ALGO_ATOMIC, probeLength=4)
        for split in (0, 1):
            res, st = _lib.hj_result(), Stats()
            pr = (C.c_void_p * 1)(dR.value); ps = (C.c_void_p * 1)(dS.value)
            nr = (C.c_uint64 * 1)(n); ns = (C.c_uint64 * 1)(S.size)
            rc = lib.hj_sharded_join(h, C.byref(params), split, n, 0, pr, nr, ps, ns, C.byref(res), C.byref(st))
            assert rc == 0, lib.hj_sharded_last_error(h)
            got = res.as_dict()
            for k in ("conflicts", "totalMatches", "inputSum", "tableSumFull", "conflictSum"):
                assert got[k] == want[k], (split, k)
            assert got["outputSum"] == want["outputSumAtomic"]
            assert (st.nRanks, st.mode, st.homeShift, st.keysMovedR, st.keysMovedS, st.tableSizePerRank) == (1, 0, 0, 0, 0, 2 * n)
        lib.hj_sharded_free(h, 0, dR); lib.hj_sharded_free(h, 0, dS)
    finally:
        lib.hj_sharded_destroy(h)


@pytest.mark.gpu
def test_main_gpus_1_split_low_prints_the_reference_line():
    r = subprocess.run([MAIN, "--algo", "atomic", "--rSize", "1048576", "--dataDistr", "uniform", "--gpus", "1", "--split", "low"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert list(j)[:8] == ["algo", "rSize", "probeLength", "hashBuildTimeInMicroseconds", "conflicts", "totalMatches", "inputSum", "outputSum"]
    assert (j["conflicts"], j["totalMatches"], j["inputSum"]) == (176864, 871712, 549507039110)
    assert j["n_gpus"] == 1 and j["split"] == "low" and j["keysMovedR"] == 0
