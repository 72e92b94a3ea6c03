#!/usr/bin/env python3
"""VSFM on the 3-D (7-point) synthetic grid of SURVEY 8(d): the v-channel soil and geometry (dx = dy = 10 m,
dz = 0.5 m, hydrostatic initial state, no boundary conditions), nx x ny x nz cells, lateral + vertical connections.
General (any connection list) assembly, CSR SpMV on 7-entry rows, BiCGStab + per-column block-Jacobi or
GMRES(30) + level-scheduled ILU(0).  Prints Newton iterations/s and per-kernel-class statistics on one GPU."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpp_amd import workloads as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=128)
    ap.add_argument("--ny", type=int, default=128)
    ap.add_argument("--nz", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--ksp", default="bcgs")
    ap.add_argument("--pc", default="bjacobi")
    ap.add_argument("--out", default="")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra mppgpu_set_option pairs")
    a = ap.parse_args()
    t0 = time.perf_counter()
    prob = P.vchannel(nx=a.nx, ny=a.ny, nz=a.nz)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    opts = {"ksp_type": a.ksp, "pc_type": a.pc}
    opts.update(kv.split("=", 1) for kv in a.opt)
    g = P.to_gpu(prob, opts)
    t_setup = time.perf_counter() - t0
    conv, reason = g.soe.StepDT(prob.dt, 1)          # warm-up (ILU analysis, allocations)
    c0 = g.soe.counters()
    g.soe.SetOption("profile", 1)
    g.soe.reset_stats()
    g.soe.synchronize()
    t0 = time.perf_counter()
    for s in range(a.steps):
        conv, reason = g.soe.StepDT(prob.dt, s + 2)
        if not conv:
            break
    g.soe.synchronize()
    t_steps = time.perf_counter() - t0
    c1 = g.soe.counters()
    n = a.nx * a.ny * a.nz
    stats = {k: g.soe.kernel_stats(k) for k in ("spmv", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce")}
    newton = c1["newton"] - c0["newton"]
    out = {"grid": [a.nx, a.ny, a.nz], "cells": n, "ksp": a.ksp, "pc": a.pc, "converged": bool(conv), "reason": reason,
           "steps": a.steps, "newton_iterations": newton, "linear_iterations": c1["linear"] - c0["linear"],
           "time_cuts": c1["time_cuts"] - c0["time_cuts"], "newton_its_per_s": newton / t_steps, "steps_s": t_steps,
           "setup_s": t_setup, "build_s": t_build,
           "kernels": {k: {"launches": s["launches"], "total_ms": round(s["total_ms"], 3),
                           "avg_ms": round(s["total_ms"] / s["launches"], 4),
                           "GBps": (round(s["bytes_per_launch"] * s["launches"] / (s["total_ms"] * 1e-3) / 1e9, 1)
                                    if s["total_ms"] > 0 and s["bytes_per_launch"] > 0 else None)}
                       for k, s in stats.items() if s["launches"]}}
    if "spmv" in out["kernels"]:
        sp = stats["spmv"]
        gbs = sp["bytes_per_launch"] * sp["launches"] / (sp["total_ms"] * 1e-3) / 1e9
        out["roofline"] = {"kernel": "csr_spmv_fp64, 7-point rows", "bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                           "achieved": round(gbs, 1), "frac": round(gbs / 8000.0, 4), "bytes_per_launch": sp["bytes_per_launch"]}
    print(json.dumps(out), flush=True)
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
