#!/usr/bin/env python3
"""Adds the radix join's dominant kernel to profiles/pmc_traffic.json (bench.py: other_workloads.prj_local_shuffle_1024.roofline.traffic):
HBM bytes of the pass-1 scatter of R = 2 * FETCH_SIZE + WRITE_SIZE of the launch on R -- the maximum over the launches (the
launch on the sorted S writes less) -- from separate --pmc passes of `python3 tools/time_prj.py --log2n 30 --mode 0 --reps 1`:

    bash tools/pmc_cmd.sh prjfetch FETCH_SIZE python3 $GRAFT_REPO_ROOT/tools/time_prj.py --log2n 30 --mode 0 --reps 1
    bash tools/pmc_cmd.sh prjwrite WRITE_SIZE python3 $GRAFT_REPO_ROOT/tools/time_prj.py --log2n 30 --mode 0 --reps 1
    python3 tools/merge_prj_traffic.py gpurun_out/pmc_prjfetch/pmc.json gpurun_out/pmc_prjwrite/pmc.json profiles/pmc_traffic.json
"""
import json
import sys


def main():
    f, w, out = (json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3])
    pick = lambda d, c: next(v[c] for k, v in d.items() if "k_radix_scatter_frag<false" in k)          # noqa: E731
    fb, wb = pick(f, "FETCH_SIZE"), pick(w, "WRITE_SIZE")
    res = json.load(open(out))
    res["k_radix_scatter_frag_pass1_R_local_shuffle_1024"] = 2.0 * fb["max"] * 1024.0 + wb["max"] * 1024.0
    res["_note_prj"] = ("k_radix_scatter_frag_pass1_R_local_shuffle_1024: HBM bytes of the pass-1 scatter of R (the radix join's dominant kernel; "
                        "tiles of 32768 keys), |R|=2^30 local_shuffle W=1024, 2 * FETCH_SIZE + WRITE_SIZE of the launch on R (the maximum over the "
                        f"launches: FETCH {fb['max'] * 1024 / 1e9:.2f} GB as reported, WRITE {wb['max'] * 1024 / 1e9:.2f} GB; the launch on the sorted S: "
                        f"WRITE {wb['min'] * 1024 / 1e9:.2f} GB), from separate --pmc passes of `python3 tools/time_prj.py --log2n 30 --mode 0 --reps 1` "
                        "(tools/pmc_cmd.sh, tools/merge_prj_traffic.py)")
    json.dump(res, open(out, "w"), indent=1)
    print(res["k_radix_scatter_frag_pass1_R_local_shuffle_1024"])


if __name__ == "__main__":
    main()
