#!/usr/bin/env python3
"""bench.py -- Newton iterations/s and SpMV GB/s of the VSFM solve path on MI355X.

Workload (BASELINE.json `metric`): "celia-extruded" VSFM grid, 1024 x 1024 soil columns x 100
layers (104,857,600 cells, vertical-only connectivity), each column the reference's celia1990 column
(src/driver/standalone/vsfm/vsfm_celia1990_problem.F90) with a per-column permeability multiplier
exp(0.5 z), z ~ N(0,1), numpy default_rng(12345); Dirichlet top/bottom; dt = 3600 s; BiCGStab + per-column
block-Jacobi, ksp_rtol 1e-5, Newton rtol 1e-8 / stol 1e-10.  One "step" = one soe%StepDT call.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched by torch.distributed.run, one rank per GPU; the columns are sharded across ranks (strong
scaling: the global grid is fixed) and the only inter-GPU traffic is the RCCL all-reduce of the Newton/Krylov
norms and dot products inside libmppgpu.so.  Host coordination (unique-id broadcast, barrier, max-over-ranks) is a
small TCP group built from the RANK / WORLD_SIZE / MASTER_* variables the launcher exports.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_problem(ncols_global, nz, col0, ncols_local, hetero=True):
    import problems as P
    mult = None
    if hetero:
        mult = np.exp(0.5 * np.random.default_rng(12345).standard_normal(ncols_global))[col0:col0 + ncols_local]
    return P.celia1990(ncols=ncols_local, nz=nz, perm_mult=mult)


def cpu_baseline(ncols_global, nz, nsteps, warmup, cores, cols_per_core, budget_s=30.0):
    """The CPU oracle ("port") on a bounded sample of the same workload: `cores` independent shards (the way
    MPP+ELM scales on CPUs: every MPI rank solves its own columns with its own serial SNES), GMRES+ILU(0) as
    the reference's PETSc defaults; reported as full-grid-equivalent Newton iterations/s."""
    import problems as P
    from oracle import vsfm_oracle as VO
    VO.lib()
    rng_cols = np.random.default_rng(99).choice(ncols_global, size=cores * cols_per_core, replace=False)
    mult_all = np.exp(0.5 * np.random.default_rng(12345).standard_normal(ncols_global))
    shards = []
    for c in range(cores):
        cols = np.sort(rng_cols[c * cols_per_core:(c + 1) * cols_per_core])
        prob = P.celia1990(ncols=cols_per_core, nz=nz, perm_mult=mult_all[cols])
        shards.append((prob, P.to_oracle(prob, with_blocks=False)))
    work = [0.0] * cores

    def run(i, steps, count):
        prob, o = shards[i]
        for s in range(steps):
            before = o.counters()["newton"]
            o.step_dt(prob.dt, s + 1)
            if count:
                work[i] += (o.counters()["newton"] - before) * prob.mesh.ncells_local

    def run_all(steps, count):
        th = [threading.Thread(target=run, args=(i, steps, count)) for i in range(cores)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    run_all(warmup, False)
    dt = run_all(nsteps, True)
    cell_its_per_s = sum(work) / dt
    n_full = ncols_global * nz
    # the reference itself is serial (PETSC_COMM_SELF, one rank): the same measure on ONE core, one more step of shard 0
    work[0] = 0.0
    t0 = time.perf_counter()
    run(0, 1, True)
    one_core = work[0] / (time.perf_counter() - t0) / n_full
    return {
        "value": cell_its_per_s / n_full, "unit": "Newton iters/s", "cores": cores, "kind": "port",
        "value_one_core": one_core,
        "sample": "%d of %d columns (%d threads x %d columns x %d layers), %d timed steps after %d warm-up, "
                  "GMRES(30)+ILU(0); %.3g cell-Newton-updates/s scaled by 1/%d cells" %
                  (cores * cols_per_core, ncols_global, cores, cols_per_core, nz, nsteps, warmup, cell_its_per_s, n_full),
        "seconds": dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--nz", type=int, default=100)
    ap.add_argument("--ksp", default="bcgs")
    ap.add_argument("--pc", default="bjacobi")
    ap.add_argument("--spmv-variant", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0)
    ap.add_argument("--homogeneous", action="store_true")
    ap.add_argument("--profile", choices=["auto", "all", "spmv"], default="auto",
                    help="kernel classes timed with HIP events inside the timed region: every class, or the roofline kernel "
                         "(CSR SpMV) only -- the events of every class cost 5-6 %% at 13 M cells per GPU, under 1 %% at 105 M; "
                         "auto = all on one GPU, spmv on several")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra mppgpu_set_option pairs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" %
                             (args.gpus, args.gpus))
        args.gpus = world

    # the HIP extension first (fails loudly if missing), torch only for multi-rank host coordination
    from mpp_amd import constants as C
    from mpp_amd import vsfm as V
    import problems as P

    # Launched by torch.distributed.run: host-side rendezvous over plain TCP (mpp_amd.parallel.HostGroup) from the
    # RANK / WORLD_SIZE / MASTER_* variables it exports.  torch itself is not imported: a rank process then loads
    # exactly one ROCm runtime and one RCCL (those of the ROCm installation), the configuration the GPU tests run.
    from mpp_amd import parallel as PAR
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    group = PAR.HostGroup() if launched else PAR.HostGroup(rank=0, world=1)

    ncols_global = args.nx * args.ny
    col0, ncols_local = PAR.shard_columns(ncols_global, world, rank)
    t_setup = time.perf_counter()
    prob = build_problem(ncols_global, args.nz, col0, ncols_local, hetero=not args.homogeneous)
    opts = {"ksp_type": args.ksp, "pc_type": args.pc, "spmv_variant": args.spmv_variant}
    opts.update(kv.split("=", 1) for kv in args.opt)
    mpp = P.to_gpu(prob, opts, device_id=local_rank)
    soe = mpp.soe
    if launched:
        # RCCL prints a version banner on stdout at communicator creation: keep stdout for the ONE JSON line
        import ctypes
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            uid = group.broadcast_bytes(V.comm_get_unique_id() if rank == 0 else b"")
            soe.comm_init(world, rank, uid)
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        soe.synchronize()
        group.barrier()

    def step(i):
        for b, (_t, _cs, vals) in enumerate(prob.bcs):
            soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, b + 1, vals)
        conv, reason = soe.StepDT(prob.dt, i + 1)
        if not conv:
            raise RuntimeError("StepDT did not converge (reason %d)" % reason)

    for i in range(args.warmup):
        step(i)
    soe.SetOption("profile", 1 if (args.profile == "all" or (args.profile == "auto" and world == 1)) else 2)
    soe.reset_stats()
    c0 = soe.counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = group.allreduce([elapsed], op="max")[0]
    c1 = soe.counters()
    newton = c1["newton"] - c0["newton"]
    linear = c1["linear"] - c0["linear"]

    stats = {k: soe.kernel_stats(k) for k in ("spmv", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce")}
    soe.SetOption("profile", 0)

    if rank == 0:
        n_global = ncols_global * args.nz
        sp = stats["spmv"]
        avg_ms = sp["total_ms"] / max(sp["launches"], 1)
        achieved = sp["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "spmv_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("cells_per_gpu") == ncols_local * args.nz:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kernels = {}
        for k, s in stats.items():
            if s["launches"]:
                ms = s["total_ms"] / s["launches"]
                kernels[k] = {"launches": s["launches"], "avg_ms": round(ms, 5),
                              "share_of_step_time": round(s["total_ms"] * 1e-3 / elapsed, 4)}
                if s["bytes_per_launch"] > 0:
                    kernels[k]["GBps"] = round(s["bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
        out = {
            "metric": "Newton iters/sec (VSFM %dx%dx%d)" % (args.nx, args.ny, args.nz),
            "value": newton / elapsed, "unit": "Newton iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "VSFM celia-extruded %dx%d columns x %d layers, %s+%s, per-column perm multiplier"
                                   % (args.nx, args.ny, args.nz, args.ksp, args.pc),
                       "cells": n_global, "cells_per_gpu": ncols_local * args.nz, "dt_s": prob.dt,
                       "parallelism": "columns sharded over %d GPU(s), RCCL all-reduce of Krylov/Newton scalars" % world},
            "newton_iterations": newton, "linear_iterations": linear, "setup_s": round(t_setup, 2),
            "roofline": {"kernel": "csr_spmv_fp64 (k_spmv_pipe_c: CSR values + compact 16-bit index, see DESIGN.md section 4)"
                                   if ncols_local * args.nz >= (1 << 23) and args.spmv_variant < 0 else "csr_spmv_fp64",
                         "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "bytes_per_launch": sp["bytes_per_launch"],
                         "avg_launch_ms": round(avg_ms, 5), "launches": sp["launches"]},
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported at N = 1 only
            cores = args.cpu_cores or min(16, os.cpu_count() or 1)
            try:
                # ~10-30 s of CPU work: 16 x 8192 columns x 100 layers = 13 M cells (1/8 of the grid), 3+1 steps
                out["cpu_baseline"] = cpu_baseline(ncols_global, args.nz, 3, 1, cores,
                                                   cols_per_core=max(64, min(819200 // args.nz, ncols_global // cores)))
            except Exception as e:  # the baseline must never hide the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "Newton iters/s", "cores": cores, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


if __name__ == "__main__":
    main()
