"""The C-ABI library loads and exports every symbol include/htm_hashjoin.h declares.
No compute calls here (no GPU in this container)."""
import ctypes
import os
import re
import subprocess

import pytest

import htm_hashjoin_amd as hj
from htm_hashjoin_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "htm_hashjoin.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hj_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    syms = _declared_symbols()
    assert len(syms) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/htm_hashjoin.h but not exported"
        assert s in hj.lib._hj_signatures, f"{s} has no ctypes signature in _lib.py"


def test_no_undeclared_exports():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("hj_")}
    assert exported == set(_declared_symbols())


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.hj_params) == 12 * 4
    assert ctypes.sizeof(_lib.hj_result) == 12 * 8 + 2 * 4 + 7 * 8 + 4 * 8 + 8 + 3 * 8 + 8     # + compactFallback (ABI 4)


def test_abi_version_and_strerror():
    assert hj.lib.hj_abi_version() == 4
    assert b"no gfx950" in hj.lib.hj_strerror(_lib.HJ_ERR_NO_DEVICE)
    assert hj.lib.hj_strerror(0) == b"ok"


@pytest.mark.skipif(hj.device_count() > 0, reason="checks the no-GPU failure mode")
def test_operators_fail_loudly_without_a_gpu():
    with pytest.raises(hj.HashJoinError) as e:
        hj.HashJoinContext(0)
    assert e.value.status == _lib.HJ_ERR_NO_DEVICE
    R = hj.generate_data("sorted", 64)
    for op in (hj.NoCCHashBuild, hj.AtomicHashBuild, hj.HTMHashBuild):
        with pytest.raises(hj.HashJoinError):
            op(R, 64, R, 64)
    with pytest.raises(hj.HashJoinError):
        hj.PRO(R, R)


def test_product_does_not_reference_the_oracle():
    """The product tree must not import, include or link anything under oracle/."""
    pkg = os.path.join(ROOT, "htm-hashjoin_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_prj_workspace_covers_both_relations_layouts():
    """hj_reserve sizes the PRJ histogram region for the larger NEED of the two relations, not for the larger
    relation: the chunk length doubles at the 8192 -> 16384 step, so |R| = 40e6 has fewer chunks than |S| = 2^25
    (round-1 ADVICE: a 700 KB out-of-bounds device write). Host arithmetic only."""
    out = (ctypes.c_uint64 * 4)()
    sizes = [1, 1000, 8191, 8192, 1 << 20, (1 << 25) - 1, 1 << 25, 33_560_000, 40_000_000, 1 << 26, 100_000_000,
             (1 << 27) + 12345, 1 << 28, 268_500_000, 1 << 30, (1 << 31) + 7, (1 << 32) - 2]
    worst = 0.0
    for bits in (0, 1, 8, 9, 14, 16):
        for a in sizes:
            for b in sizes + [0]:
                assert hj.lib.hj_prj_workspace_info(a, b, bits, out) == 0
                ws, planned, need_r, need_s = (int(x) for x in out)
                assert planned >= need_r and planned >= need_s, (a, b, bits, planned, need_r, need_s)
                assert ws >= 4 * planned
                if b and a > b and need_s > need_r:
                    worst = max(worst, need_s / need_r)
    assert worst > 1.5          # the non-monotone case exists (and is what the region is now sized for)
    assert hj.lib.hj_prj_workspace_info(40_000_000, 1 << 25, 14, out) == 0 and out[3] > out[2]


def test_prj_fragment_geometry_invariants():
    """The histogram-free radix passes (hj_params.prjMode 0 / 2) lay every relation out as fixed-capacity fragments. Host
    arithmetic only: whatever sizes come in, a geometry that is handed to the kernels keeps every position below 2^32 and
    inside the 8-bytes-per-tuple buffers, cuts pass 2 along whole pass-1 fragments, starts every fragment on a 128-byte
    line, leaves at least the expected number of keys + 7 sigma per fragment and never plans a partition larger than the
    join's LDS table."""
    out = (ctypes.c_uint64 * 13)()
    sizes = [1, 4097, 1 << 18, 1 << 20, 3_000_001, 1 << 22, 1 << 24, 19_999_999, (1 << 25) - 1, 1 << 25, 40_000_000, 1 << 26,
             100_000_000, 1 << 27, (1 << 27) + 12345, 1 << 28, 1 << 30, (1 << 31) - 5, 1 << 31, (1 << 31) + 1, (1 << 32) - 2]
    taken = 0
    for mode in (0, 1, 2):
        for bits in (0, 4, 8, 9, 12, 14, 15, 16):
            for nR in sizes:
                for nS in (0, nR, sizes[(sizes.index(nR) * 7 + 3) % len(sizes)]):
                    assert hj.lib.hj_prj_fragment_info(nR, nS, bits, mode, out) == 0
                    v = [int(x) for x in out]
                    b1, b2 = v[11], v[12]
                    rel = [(nR, v[1:6]), (nS, v[6:11])]
                    if mode == 1 or b2 == 0:
                        assert v[0] == 0 and v[1] == 0 and v[6] == 0
                    assert v[0] == int(v[1] != 0 and (nS == 0 or v[6] != 0))
                    for n, (C1, cap1, chunk1, C2, cap2) in rel:
                        if C1 == 0:
                            continue
                        taken += 1
                        F1, F2 = 1 << b1, 1 << b2
                        assert n <= 1 << 31 and (mode == 2 or n >= 1 << 25)
                        assert C1 & (C1 - 1) == 0 and C2 & (C2 - 1) == 0 and C1 % C2 == 0 and C2 <= 16 and C1 // C2 <= 1024
                        assert chunk1 % 8192 == 0 and C1 * chunk1 >= n                    # the chunks cover the relation
                        assert cap1 % 32 == 0 and cap2 % 32 == 0
                        mean1 = chunk1 / F1
                        mean2 = (C1 // C2) * mean1 / F2
                        assert cap1 >= mean1 + 7 * mean1 ** 0.5 and cap2 >= mean2 + 7 * mean2 ** 0.5
                        assert F1 * C1 * cap1 <= 2 * n and F1 * F2 * C2 * cap2 <= 2 * n       # inside the key buffers
                        assert F1 * C1 * cap1 < 1 << 32 and F1 * F2 * C2 * cap2 < 1 << 32
                        assert C2 * cap2 <= (65535 if b1 + b2 >= 16 else 24576)
    assert taken > 100
    # the sizes the bench and the sweep run
    assert hj.lib.hj_prj_fragment_info(1 << 30, 1 << 30, 0, 0, out) == 0
    assert [int(x) for x in out][:6] == [1, 1024, 4576, 1 << 20, 4, 4576]
    assert hj.lib.hj_prj_fragment_info(1 << 27, 1 << 27, 0, 0, out) == 0 and out[0] == 1 and out[11] == 7 and out[12] == 7
    assert hj.lib.hj_prj_fragment_info(1 << 20, 1 << 20, 0, 0, out) == 0 and out[0] == 0      # small: exact passes
