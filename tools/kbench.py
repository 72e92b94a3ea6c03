#!/usr/bin/env python3
"""Device-resident micro-benchmark of every kernel class / variant on a celia-extruded grid.
usage: python tools/kbench.py [--nx 1024 --ny 1024 --nz 100] [--reps 20]
Prints one line per kernel: avg ms, algorithmic GB/s, fraction of the 8 TB/s HBM peak."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from mpp_amd import workloads as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--nz", type=int, default=100)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ncols = args.nx * args.ny
    prob = bench.build_problem(ncols, args.nz, 0, ncols)
    mpp = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    soe = mpp.soe
    conv, reason = soe.StepDT(prob.dt, 1)   # populate J, F, factors with a real state
    rows = []
    plan = [("copy", 0), ("dot", 0), ("waxpy", 0), ("dots3", 0), ("bcgs_xr", 0),
            ("spmv", 0), ("spmv", 1), ("spmv", 2), ("spmv", 3), ("spmv", 4), ("spmv", 5), ("spmv", 8), ("spmv", 9), ("spmv", 17), ("spmv", 18), ("spmv", 10), ("spmv", 20), ("spmv", 21),
            ("pc_setup", 0), ("pc_apply", 0), ("residual", 0), ("jacobian", 0), ("assemble_both", 0)]
    for name, var in plan:
        ms, by = soe.bench_kernel(name, var, args.reps)
        gbs = by / (ms * 1e-3) / 1e9
        rows.append({"kernel": name, "variant": var, "avg_ms": ms, "alg_bytes": by, "GBps": gbs, "frac_of_8TBps": gbs / 8000.0})
        print("%-14s v%d  %9.4f ms  %9.1f GB/s  %5.1f%%" % (name, var, ms, gbs, gbs / 80.0), flush=True)
    if args.out:
        json.dump({"grid": [args.nx, args.ny, args.nz], "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
