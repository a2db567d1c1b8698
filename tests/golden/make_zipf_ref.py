#!/usr/bin/env python3
"""Regenerates tests/golden/zipf_ref.json from the REFERENCE's own Zipf generator: oracle/_ref/genzipf_ref =
oracle/zipf_ref_driver.c (ours, a main()) + /root/reference/mc/src/genzipf.c compiled where it lies (oracle/Makefile).
Run in the container that has the reference checkout:  make -C oracle && python tests/golden/make_zipf_ref.py

Each row: gen_zipf(stream_size, alphabet_size, theta) after srand(seed) -- the first 32 keys, the sum of all keys and
an FNV-1a-64 hash over all keys. Seeds: 0 (DataGen.hpp:27, what hj_generate_data uses), 12345 and 54321
(mc/src/main.c:337-338)."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "..", "..", "oracle", "_ref", "genzipf_ref")

CASES = [  # (stream_size, alphabet_size, theta, seed)
    (1000, 100, 0.9, 0), (1000, 100, 0.9, 12345), (65536, 65536, 0.9, 0), (65536, 1024, 0.5, 54321),
    (1 << 20, 1 << 18, 0.9, 0), (1 << 20, 1 << 20, 1.0, 0), (3 * (1 << 18) + 7, 1 << 16, 0.9, 1),
    (1 << 21, 1 << 18, 0.9, 0),      # tests/test_gpu_parity.py::test_skew_probe_side_zipf draws 8 * 2^18 + 3 of these
    (5, 1, 0.9, 0), (64, 2, 0.0, 0), (1 << 16, 1 << 22, 1.5, 12345),
]


def main():
    rows = [json.loads(subprocess.run([EXE, str(n), str(a), repr(t), str(s), "32"], capture_output=True, text=True,
                                      check=True).stdout) for n, a, t, s in CASES]
    doc = {"source": "oracle/_ref/genzipf_ref: oracle/zipf_ref_driver.c + /root/reference/mc/src/genzipf.c (gen_zipf, "
                     "genzipf.c:95-158), glibc rand()", "rows": rows}
    with open(os.path.join(HERE, "zipf_ref.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(f"wrote {len(rows)} rows")


if __name__ == "__main__":
    main()
