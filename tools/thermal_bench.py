#!/usr/bin/env python3
"""Thermal KSP step (SURVEY 8 f-1 / BASELINE config 4) on one GPU: 3-D heat-diffusion MMS problem, n^3 cells,
GMRES(30) + ILU(0) (PETSc defaults) or another ksp/pc.  Prints timings and per-kernel-class statistics."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mpp_amd import workloads as P  # noqa: E402


def cpu_baseline(n, n_full, rtol):
    """the CPU oracle ("port": thermal_oracle.c + the restated GMRES(30)+ILU(0), one core) on a smaller grid of the same
    problem; reported as cell-iterations per second so that it can be set against the GPU's"""
    from oracle import adapters as A   # the CPU oracle: baseline leg only
    tp = P.thermal_mms(3, n, n, n, replicate_reference_soil_indexing=False)
    o = A.to_thermal_oracle(tp, ksp_rtol=rtol)
    t0 = time.perf_counter()
    o.step_dt(tp.dt)
    dt = time.perf_counter() - t0
    its = o.linear_iterations()
    return {"value": its * n ** 3 / dt, "unit": "cell-iterations/s", "cores": 1, "kind": "port",
            "sample": "%d^3 grid (GPU run: %d^3), %d GMRES iterations in %.2f s" % (n, n_full, its, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--ksp", default="gmres")
    ap.add_argument("--pc", default="ilu")
    ap.add_argument("--rtol", type=float, default=1e-5)
    ap.add_argument("--out", default="")
    ap.add_argument("--cpu-n", type=int, default=64, help="grid size of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra mppgpu_set_option pairs")
    ap.add_argument("--rtol-sweep", default="", help="comma-separated ksp_rtol values solved one after the other on the same problem "
                                                     "(time, iterations and error against the manufactured solution for each)")
    ap.add_argument("--then", action="append", default=[], metavar="KEY=VALUE[,KEY=VALUE]",
                    help="after the timed step: set these options and time the step again (repeatable; A/B on one set-up)")
    ap.add_argument("--no-events", action="store_true", help="no HIP-event timing of the kernel classes")
    ap.add_argument("--dump-maps", default="", help="write /proc/self/maps there before the timed step (to attribute stack frames)")
    a = ap.parse_args()
    t0 = time.perf_counter()
    tp = P.thermal_mms(3, a.n, a.n, a.n, replicate_reference_soil_indexing=False)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    opts = {"ksp_type": a.ksp, "pc_type": a.pc, "ksp_rtol": a.rtol}
    opts.update(kv.split("=", 1) for kv in a.opt)
    g = P.to_thermal_gpu(tp, opts)
    t_setup = time.perf_counter() - t0
    g.soe.StepDT(tp.dt, 1)          # warm-up (ILU analysis, allocations)
    g.soe.PreStepDT()
    g.soe.SetOption("profile", 0 if a.no_events else 1)
    g.soe.reset_stats()
    if a.dump_maps:
        open(a.dump_maps, "w").write(open("/proc/self/maps").read())
    g.soe.synchronize()
    t0 = time.perf_counter()
    conv, _ = g.soe.StepDT(tp.dt, 1)
    g.soe.synchronize()
    t_step = time.perf_counter() - t0
    err = float(np.abs(g.soe.GetSoln() - tp.exact).max())
    stats = {k: g.soe.kernel_stats(k) for k in ("spmv", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce")}
    out = {"grid": a.n, "cells": a.n ** 3, "ksp": a.ksp, "pc": a.pc, "converged": conv,
           "linear_iterations": g.soe.last_linear_iterations, "step_s": t_step, "setup_s": t_setup, "build_s": t_build,
           "max_error_vs_manufactured_solution": err, "cell_iterations_per_s": a.n ** 3 * g.soe.last_linear_iterations / t_step,
           "kernels": {k: {"launches": s["launches"], "total_ms": round(s["total_ms"], 3),
                           "GBps": (round(s["bytes_per_launch"] * s["launches"] / (s["total_ms"] * 1e-3) / 1e9, 1)
                                    if s["total_ms"] > 0 and s["bytes_per_launch"] > 0 else None)}
                       for k, s in stats.items() if s["launches"]}}
    # roofline of the dominant kernel class (ILU(0) apply): algorithmic bytes 12 nnz + 24 N
    N = a.n ** 3
    nnz = 7 * N - 6 * a.n * a.n
    pa = stats["pc_apply"]
    if pa["launches"]:
        ms = pa["total_ms"] / pa["launches"]
        by = 12.0 * nnz + 24.0 * N
        info = g.soe.pc_info() if a.pc == "ilu" else {"path": "none", "launches": 0, "box_extents": (0, 0, 0)}
        what = {"levels": "level-ordered ELL, one launch per dependency level", "chained": "level-ordered ELL, tiles chained by flags",
                "boxes": ("box wavefronts of %d x %d x %d cells, " % info["box_extents"]) +
                         ("two streams of box pencils + two layout passes per application" if info["launches"] == 4
                          else "one launch per box level")}.get(info["path"], a.pc)
        out["roofline"] = {"kernel": "ILU(0) triangular solves (%s)" % what, "bound": "hbm", "unit": "GB/s",
                           "peak": 8000.0, "achieved": round(by / (ms * 1e-3) / 1e9, 1), "frac": round(by / (ms * 1e-3) / 1e9 / 8000.0, 4),
                           "bytes_per_launch": by, "avg_apply_ms": round(ms, 4), "dependency_levels": 2 * (3 * a.n - 2),
                           "launches_per_application": info["launches"],
                           "note": "algorithmic bytes 12 nnz + 24 N (SURVEY 8d) over the average application (layout passes included); the solves are chains of dependent steps"}
    # further timed steps of the same problem with options changed in between (A/B pairs on one box, one set-up)
    variants = []
    for spec in a.then:
        for kv in spec.split(","):
            k, v = kv.split("=", 1)
            g.soe.SetOption(k, v)
        g.soe.PreStepDT()
        g.soe.reset_stats()
        g.soe.synchronize()
        t0 = time.perf_counter()
        conv_v, _ = g.soe.StepDT(tp.dt, 1)
        g.soe.synchronize()
        dt_v = time.perf_counter() - t0
        pa_v = g.soe.kernel_stats("pc_apply")
        variants.append({"options": spec, "converged": conv_v, "linear_iterations": g.soe.last_linear_iterations, "step_s": round(dt_v, 4),
                         "pc_apply_avg_ms": round(pa_v["total_ms"] / max(pa_v["launches"], 1), 4), "pc_apply_launches_timed": pa_v["launches"],
                         "chain_fallbacks": g.soe.comm_stats()["chain_fallbacks"],
                         "max_error_vs_manufactured_solution": float(np.abs(g.soe.GetSoln() - tp.exact).max())})
        print("then %s: %d its, %.3f s, ILU(0) application %.3f ms" % (spec, variants[-1]["linear_iterations"], dt_v, variants[-1]["pc_apply_avg_ms"]),
              file=sys.stderr, flush=True)
    if variants:
        out["then"] = variants
    if a.rtol_sweep:
        # time to a given accuracy: PETSc's default rtol 1e-5 (on the preconditioned residual) stops far above the discretisation
        # error on fine grids, so every timing is reported next to the error it bought
        sweep = []
        for r in [float(x) for x in a.rtol_sweep.split(",")]:
            g.soe.SetOption("ksp_rtol", r)
            g.soe.PreStepDT()
            g.soe.synchronize()
            t0 = time.perf_counter()
            conv_r, _ = g.soe.StepDT(tp.dt, 1)
            g.soe.synchronize()
            dt_r = time.perf_counter() - t0
            sweep.append({"ksp_rtol": r, "converged": conv_r, "linear_iterations": g.soe.last_linear_iterations, "step_s": round(dt_r, 4),
                          "max_error_vs_manufactured_solution": float(np.abs(g.soe.GetSoln() - tp.exact).max())})
            print("rtol %.1e: %d its, %.3f s, error %.4g K" % (r, sweep[-1]["linear_iterations"], dt_r,
                                                                sweep[-1]["max_error_vs_manufactured_solution"]), file=sys.stderr, flush=True)
        out["rtol_sweep"] = sweep
    if a.cpu_n > 0:
        out["cpu_baseline"] = cpu_baseline(a.cpu_n, a.n, a.rtol)
    print(json.dumps(out), flush=True)
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
