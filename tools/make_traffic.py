#!/usr/bin/env python3
"""profiles/kernel_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` at one grid size.

    python3 tools/make_traffic.py <pmc summary json of tools/pmc_summary.py> <cells_per_gpu> <run label> > profiles/kernel_traffic.json

The file carries the sha256 of the kernel sources the passes were made with (mpp_amd/csrc/*): bench.py quotes the moved
bytes of a kernel class only when that hash equals the one of the sources it runs (a stale file is refused, not replayed)."""
from __future__ import annotations

import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel class of bench.py's `kernels` object -> the kernels that make up a launch of that class (substring of the demangled name)
# kernel class of bench.py's `kernels` object -> how a launch of that class is made up: ("sum", ...) = every launch of the class runs
# all of these kernels (substrings of the demangled name); ("mean", ...) = launches of the class alternate between them
CLASSES = {
    "spmv": ("first", ["k_spmv_cols_dots", "k_spmv_c<3, false, true>"]),   # the launches bench.py times in the SpMV class (the line search's
                                                                          # pass): index-free on soil columns (round 4), else the compact index
    "spmv_plain": ("sum", ["k_spmv_c<3, false, false>"]),
    # first kernel of a solve / second product; round 5: with the line search's F . (J p), F . (J s) reduced on the way (ls_fold), which
    # is what every launch of the timed region is (a substring match would also catch the <.., false> instantiations of a solve that
    # goes on: they do not run in the benchmark)
    "spmv_pc": ("mean", ["k_tri_solve_lds<2, 4, 3, 4, 2, true>", "k_tri_solve_lds<2, 5, 2, 4, 3, true>"]),
    "residual": ("sum", ["k_assemble_cols<0,", "k_bc<0>"]),
    "jacobian": ("sum", ["k_assemble_cols<2,", "k_bc<2>"]),
    "pc_apply": ("sum", ["k_tri_back_dots"]),
    "krylov_update": ("sum", ["k_bcgs_xr<true, true>"]),
    "vec": ("sum", ["k_waxpy"]),
}


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mpp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    pmc = json.load(open(sys.argv[1]))
    cells = int(sys.argv[2])
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    out = {"cells_per_gpu": cells, "kernel_source_sha16": source_hash(), "run": label,
           "correction": "reads = %.5f x FETCH_SIZE (gfx950: 128-byte requests tallied at 64 B), calibrated on k_copy of the same run"
                         % pmc["read_factor"],
           "classes": {}}
    for cls, (mode, subs) in CLASSES.items():
        ks = {}
        for sub in subs:
            for name, v in pmc["kernels"].items():
                short = re.sub(r"\(.*$", "", name.replace("mpp::(anonymous namespace)::", "").replace("void ", ""))
                if sub in short:
                    ks[short] = v["hbm_bytes_per_launch"]
            if mode == "first" and ks:     # the first of the alternatives that ran
                mode = "sum"
                break
        if mode == "first":
            mode = "sum"
        if ks:
            total = sum(ks.values())
            out["classes"][cls] = {"kernels": ks, "hbm_bytes_per_launch": total if mode == "sum" else total / len(ks),
                                   "launch_is": "all of these kernels" if mode == "sum" else "one of these kernels in turn (mean)"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
