#!/usr/bin/env python3
"""Device-resident micro-benchmark of every kernel class / variant on a celia-extruded grid.
usage: python tools/kbench.py [--nx 1024 --ny 1024 --nz 100] [--reps 20] [--only NAME[:VARIANT[:KEY=VAL,...]] ...]
Prints one line per kernel: avg ms, algorithmic GB/s, fraction of the 8 TB/s HBM peak."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from mpp_amd import workloads as P  # noqa: E402

DEFAULT_PLAN = ["copy", "dot", "waxpy", "dots3", "bcgs_xr", "spmv:0", "spmv:1", "spmv:17", "spmv:20", "spmv:21",
                "pc_setup", "pc_apply", "pc_apply_dots:0", "pc_apply_dots:1:tri_group=8", "pc_apply_dots:1:tri_group=16",
                "pc_apply_dots:1:tri_group=24", "residual", "jacobian", "assemble_both"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--nz", type=int, default=100)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", nargs="*", default=None, help="entries NAME[:VARIANT[:KEY=VAL,...]] instead of the default plan")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="mppgpu_set_option pairs before setup")
    args = ap.parse_args()
    ncols = args.nx * args.ny
    prob = bench.build_problem(ncols, args.nz, 0, ncols)
    opts = {"ksp_type": "bcgs", "pc_type": "bjacobi"}
    opts.update(kv.split("=", 1) for kv in args.opt)
    mpp = P.to_gpu(prob, opts)
    soe = mpp.soe
    conv, reason = soe.StepDT(prob.dt, 1)   # populate J, F, factors with a real state
    rows = []
    for entry in (args.only if args.only is not None else DEFAULT_PLAN):
        parts = entry.split(":")
        name, var = parts[0], int(parts[1]) if len(parts) > 1 and parts[1] != "" else 0
        kv = dict(p.split("=", 1) for p in parts[2].split(",")) if len(parts) > 2 else {}
        for k, v in kv.items():
            soe.SetOption(k, v)
        ms, by = soe.bench_kernel(name, var, args.reps)
        gbs = by / (ms * 1e-3) / 1e9
        rows.append({"kernel": name, "variant": var, "options": kv, "avg_ms": ms, "alg_bytes": by, "GBps": gbs,
                     "frac_of_8TBps": gbs / 8000.0})
        print("%-14s v%-2d %-18s %9.4f ms  %9.1f GB/s  %5.1f%%" % (name, var, ",".join("%s=%s" % i for i in kv.items()), ms, gbs, gbs / 80.0),
              flush=True)
    if args.out:
        json.dump({"grid": [args.nx, args.ny, args.nz], "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
