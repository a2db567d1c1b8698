"""CPU-side AddressSanitizer + UBSan pass over the oracle and the product's host DataGen (GPU sanitizers are
not available on the GPU pool). The reference has no sanitizer coverage; ASan would have caught its
out-of-bounds conflictSum read (AtomicHashBuild.hpp:111-114)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_datagen_are_clean_under_asan_ubsan():
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "sanitize.sh")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle sanitizer pass ok" in r.stdout and "datagen sanitizer pass ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
