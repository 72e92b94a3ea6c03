#!/usr/bin/env python3
"""Coupled thermal-hydrology steps (SURVEY 8 f-2 / BASELINE config 5 shape) on one GPU: nx*ny soil columns x nz layers,
Richards + enthalpy heat equation, 2N unknowns [P; T], Newton + GMRES(30)/BiCGStab + ILU(0).  Prints Newton
iterations/s and per-kernel-class statistics."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mpp_amd import workloads as P  # noqa: E402


def cpu_baseline(cols, nz, ncols_full):
    """the CPU oracle ("port": th_oracle.c, GMRES(30)+ILU(0) as the reference's PETSc defaults, one core -- the reference is
    serial) on a bounded sample of the same workload, scaled to full-grid Newton iterations/s"""
    from oracle import adapters as A   # the CPU oracle: baseline leg only
    mult = np.exp(0.5 * np.random.default_rng(12345).standard_normal(ncols_full))[:cols]
    tp = P.th_columns(ncols=cols, nz=nz, perm_mult=mult)
    o = A.to_th_oracle(tp)
    o.step_dt(tp.dt)
    c0 = o.counters()["newton"]
    t0 = time.perf_counter()
    for s in range(2):
        A.th_set_conditions(tp, o)
        o.step_dt(tp.dt)
    dt = time.perf_counter() - t0
    its = o.counters()["newton"] - c0
    return {"value": its * cols / ncols_full / dt, "unit": "Newton iters/s", "cores": 1, "kind": "port",
            "sample": "%d of %d columns x %d layers, 2 timed steps after 1 warm-up, GMRES(30)+ILU(0); %d Newton iterations in %.2f s"
                      % (cols, ncols_full, nz, its, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=128)
    ap.add_argument("--ny", type=int, default=128)
    ap.add_argument("--nz", type=int, default=80)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--ksp", default="gmres")
    ap.add_argument("--pc", default="ilu")
    ap.add_argument("--out", default="")
    ap.add_argument("--cpu-cols", type=int, default=256, help="columns of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra mppgpu_set_option pairs")
    ap.add_argument("--no-events", action="store_true", help="no HIP-event timing of the kernel classes (runs under rocprofv3)")
    a = ap.parse_args()
    # one process per GPU under torch.distributed.run (BASELINE config 5 is an 8-GPU run): the columns shard across the
    # ranks, RCCL all-reduces the Krylov / Newton scalars -- the same set-up as bench.py
    from mpp_amd import parallel as PAR
    from mpp_amd import vsfm as V
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    group = PAR.HostGroup() if launched else PAR.HostGroup(rank=0, world=1)
    ncols_global = a.nx * a.ny
    col0, ncols = PAR.shard_columns(ncols_global, world, rank)
    t0 = time.perf_counter()
    mult = np.exp(0.5 * np.random.default_rng(12345).standard_normal(ncols_global))[col0:col0 + ncols]
    tp = P.th_columns(ncols=ncols, nz=a.nz, perm_mult=mult)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    opts = {"ksp_type": a.ksp, "pc_type": a.pc}
    opts.update(kv.split("=", 1) for kv in a.opt)
    g = P.to_th_gpu(tp, opts, device_id=int(os.environ.get("LOCAL_RANK", "0")))
    if launched:
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)            # RCCL's banner goes to stderr, stdout keeps the one JSON line
        try:
            uid = group.broadcast_bytes(V.comm_get_unique_id() if rank == 0 else b"")
            g.soe.comm_init(world, rank, uid)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    t_setup = time.perf_counter() - t0
    conv, reason = g.soe.StepDT(tp.dt, 1)          # warm-up (ILU analysis, allocations)
    c0 = g.soe.counters()
    g.soe.SetOption("profile", 0 if a.no_events else 1)
    g.soe.reset_stats()
    g.soe.synchronize()
    group.barrier()
    t0 = time.perf_counter()
    for s in range(a.steps):
        P.th_gpu_set_conditions(tp, g)
        conv, reason = g.soe.StepDT(tp.dt, s + 2)
        if not conv:
            break
    g.soe.synchronize()
    group.barrier()
    t_steps = group.allreduce([time.perf_counter() - t0], op="max")[0]
    c1 = g.soe.counters()
    stats = {k: g.soe.kernel_stats(k) for k in ("spmv", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce")}
    newton = c1["newton"] - c0["newton"]
    if rank != 0:
        return
    out = {"n_gpus": world, "columns": ncols_global, "columns_per_gpu": ncols, "nz": a.nz, "cells": ncols_global * a.nz,
           "unknowns": 2 * ncols_global * a.nz, "ksp": a.ksp, "pc": a.pc,
           "converged": conv, "reason": reason, "steps": a.steps, "newton_iterations": newton,
           "linear_iterations": c1["linear"] - c0["linear"], "time_cuts": c1["time_cuts"] - c0["time_cuts"],
           "newton_its_per_s": newton / t_steps, "steps_s": t_steps, "setup_s": t_setup, "build_s": t_build,
           "kernels": {k: {"launches": s["launches"], "total_ms": round(s["total_ms"], 3),
                           "avg_ms": round(s["total_ms"] / s["launches"], 4),
                           "GBps": (round(s["bytes_per_launch"] * s["launches"] / (s["total_ms"] * 1e-3) / 1e9, 1)
                                    if s["total_ms"] > 0 and s["bytes_per_launch"] > 0 else None)}
                       for k, s in stats.items() if s["launches"]}}
    # roofline of the dominant kernel (fused column assembly): algorithmic bytes per cell = x 16 + soil/heat parameters 80
    # + accum_prev 16 + connection geometry 56 read, F 16 + 12 Jacobian entries 96 written = 280 B
    if "jacobian" in out["kernels"]:
        jk = out["kernels"]["jacobian"]
        by = 280.0 * ncols * a.nz
        out["roofline"] = {"kernel": "k_th_cols<2> (residual + 2x2-block Jacobian)", "bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                           "achieved": round(by / (jk["avg_ms"] * 1e-3) / 1e9, 1), "frac": round(by / (jk["avg_ms"] * 1e-3) / 1e9 / 8000.0, 4),
                           "bytes_per_launch": by, "note": "pow/div-bound at occupancy 1 (256 VGPRs), not bandwidth-bound"}
    if a.cpu_cols > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(a.cpu_cols, a.nz, ncols_global)
    print(json.dumps(out), flush=True)
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
