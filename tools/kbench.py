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
    ap.add_argument("--satfunc", choices=["vg", "vg_m", "bc", "sbc_bz2", "sbc_bz3", "mixed"], default="vg",
                    help="saturation function of every cell: van Genuchten m = 1/2 (the celia soil), van Genuchten m = 0.4 (general "
                         "powers), Brooks-Corey, smoothed Brooks-Corey bz2 / bz3 (ELM's default), or vg / bc / sbc_bz3 mixed per column")
    ap.add_argument("--density", choices=["constant", "tgdpb01", "ifc67"], default="tgdpb01")
    ap.add_argument("--per-cell-soils", action="store_true",
                    help="every cell its own (alpha, lambda, residual saturation): no soil classes, ELM-style soil maps")
    ap.add_argument("--soil-classes", type=int, default=0, help="this many distinct soil tuples, assigned per cell at random")
    args = ap.parse_args()
    ncols = args.nx * args.ny
    prob = bench.build_problem(ncols, args.nz, 0, ncols)
    import numpy as np
    from mpp_amd import constants as C
    n = prob.alpha.size
    sf = {"vg": C.SAT_FUNC_VAN_GENUCHTEN, "vg_m": C.SAT_FUNC_VAN_GENUCHTEN, "bc": C.SAT_FUNC_BROOKS_COREY,
          "sbc_bz2": C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2, "sbc_bz3": C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3}
    if args.satfunc == "mixed":
        kinds = np.array([C.SAT_FUNC_VAN_GENUCHTEN, C.SAT_FUNC_BROOKS_COREY, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3], dtype=np.int32)
        prob.satfunc_type = np.repeat(kinds[np.arange(ncols) % 3], args.nz).astype(np.int32)
    else:
        prob.satfunc_type = np.full(n, sf[args.satfunc], dtype=np.int32)
    if args.satfunc == "vg_m":
        prob.lam = np.full(n, 0.4)
    prob.density_type = {"constant": C.DENSITY_CONSTANT, "tgdpb01": C.DENSITY_TGDPB01, "ifc67": C.DENSITY_IFC67}[args.density]
    rng = np.random.default_rng(2024)
    if args.per_cell_soils:
        prob.alpha = prob.alpha * (1.0 + 0.05 * rng.random(n))
        prob.sat_res = prob.sat_res * (1.0 + 0.05 * rng.random(n))
        if args.satfunc != "vg":
            prob.lam = prob.lam * (1.0 + 0.05 * rng.random(n))
    elif args.soil_classes > 1:
        k = args.soil_classes
        ta, tr = 1.0 + 0.05 * rng.random(k), 1.0 + 0.05 * rng.random(k)
        cls = rng.integers(0, k, n)
        prob.alpha = prob.alpha * ta[cls]
        prob.sat_res = prob.sat_res * tr[cls]
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
        json.dump({"grid": [args.nx, args.ny, args.nz], "satfunc": args.satfunc, "density": args.density,
                   "per_cell_soils": args.per_cell_soils, "soil_classes": args.soil_classes, "converged_first_step": bool(conv),
                   "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
