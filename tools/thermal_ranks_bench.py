#!/usr/bin/env python3
"""BASELINE config 4 (thermal n^3, GMRES(30) + ILU(0), 4 GPUs) as far as ONE GPU can show it.

Three measurements on one box:
  (1) the single handle on the whole n^3 grid: iterations, time per iteration, error against the manufactured solution;
  (2) `world` in-process ranks (one handle + one host thread each, mpp_amd.vsfm.comm_init_local): every rank builds its own z-slab
      (workloads.thermal_mms_rank: no global mesh), the handles form ONE global GMRES(30) + rank-local ILU(0) solve with halo exchanges
      and all-reduced dots -- what `bench.py --workload thermal --gpus 4` runs, on one device.  This gives the ITERATION COUNT of the
      split preconditioner, the halo bytes per exchange and the number of exchanges / all-reduces per iteration; its wall time is
      NOT a 4-GPU time (the four handles share one GPU);
  (3) one rank's share alone: a single handle on n x n x (n / world) cells -- the kernels a rank runs, at the size it runs them,
      with the GPU to itself: time per GMRES iteration of a rank.

PROJECTION (labelled as such in the output): iterations of (2) x [time per iteration of (3) + exchanges per iteration x halo time +
all-reduces per iteration x all-reduce latency], with the halo time = halo bytes / 50 GB/s per xGMI link direction (one link between a
pair of GPUs: ~64 GB/s raw each way, MI355X_MICROARCH.md) + 10 us, and 20 us per 4-rank all-reduce of a few doubles -- assumptions, not
measurements: RCCL at N > 1 has never run on hardware available to this build."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpp_amd import vsfm as V  # noqa: E402
from mpp_amd import workloads as P  # noqa: E402


def run_ranks(fns):
    out, errors = [None] * len(fns), []

    def run(r):
        try:
            out[r] = fns[r]()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise RuntimeError("rank failure: %r" % errors)
    return out


def timed_step(soe, dt):
    soe.PreStepDT()
    soe.synchronize()
    t0 = time.perf_counter()
    conv, _ = soe.StepDT(dt, 1)
    soe.synchronize()
    return conv, soe.last_linear_iterations, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--rtol", type=float, default=1e-5)
    ap.add_argument("--skip-single", action="store_true", help="skip measurement (1) (the whole grid on one handle)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    n, world = a.n, a.world
    opts = {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": a.rtol}
    out = {"grid": n, "cells": n ** 3, "world": world, "ksp_rtol": a.rtol}

    if not a.skip_single:
        tp = P.thermal_mms(3, n, n, n, replicate_reference_soil_indexing=False)
        g = P.to_thermal_gpu(tp, opts)
        g.soe.StepDT(tp.dt, 1)
        conv, its, dt = timed_step(g.soe, tp.dt)
        out["single_handle"] = {"converged": conv, "linear_iterations": its, "step_s": round(dt, 4), "ms_per_iteration": round(dt / max(its, 1) * 1e3, 4),
                                "max_error_K": float(np.abs(g.soe.GetSoln() - tp.exact).max())}
        print("single handle: %d its, %.3f s" % (its, dt), file=sys.stderr, flush=True)
        g.soe.close()
        del g, tp

    # (2) the four ranks in this process
    ranks = [P.thermal_mms_rank(n, n, n, world, r) for r in range(world)]
    gs = [P.to_thermal_gpu(lp, opts) for _, lp in ranks]
    V.comm_init_local([g.soe for g in gs])
    dt0 = ranks[0][1].dt
    run_ranks([lambda g=g: g.soe.StepDT(dt0, 1) for g in gs])            # warm-up
    for g in gs:
        g.soe.reset_stats()
    res = run_ranks([lambda g=g: timed_step(g.soe, dt0) for g in gs])
    its_split = res[0][1]
    comm = gs[0].soe.comm_stats()
    err = max(float(np.abs(g.soe.GetSoln()[:part.nowned] - lp.exact[:part.nowned]).max()) for (part, lp), g in zip(ranks, gs))
    halo_cells = [int(sum(part.recv_count)) for part, _ in ranks]
    out["in_process_ranks"] = {"converged": all(r[0] for r in res), "linear_iterations": its_split, "same_on_every_rank": len({r[1] for r in res}) == 1,
                               "wall_s_sharing_one_gpu": round(max(r[2] for r in res), 4), "max_error_K": err,
                               "ghost_cells_per_rank": halo_cells, "halo_bytes_per_exchange_per_rank": [8 * h for h in halo_cells],
                               "halo_bytes_per_neighbour": int(ranks[0][0].recv_count[0]) * 8 if len(ranks[0][0].recv_count) else 0,
                               "halo_exchanges_rank0": comm["halo_exchanges"], "allreduces_rank0": comm["allreduces"],
                               "exchanges_per_iteration": round(comm["halo_exchanges"] / max(its_split, 1), 3),
                               "allreduces_per_iteration": round(comm["allreduces"] / max(its_split, 1), 3),
                               "ilu_path": gs[0].soe.pc_info()["path"]}
    print("in-process ranks: %d its" % its_split, file=sys.stderr, flush=True)
    for g in gs:
        g.soe.close()
    del gs, ranks

    # (3) one rank's share with the GPU to itself
    nzr = n // world
    tp = P.thermal_mms(3, n, n, nzr, replicate_reference_soil_indexing=False)
    g = P.to_thermal_gpu(tp, opts)
    g.soe.StepDT(tp.dt, 1)
    g.soe.SetOption("profile", 1)
    g.soe.reset_stats()
    conv, its, dt = timed_step(g.soe, tp.dt)
    stats = {k: g.soe.kernel_stats(k) for k in ("spmv", "pc_apply", "vec", "reduce")}
    out["one_rank_share_alone"] = {"cells": n * n * nzr, "linear_iterations": its, "step_s": round(dt, 4), "ms_per_iteration": round(dt / max(its, 1) * 1e3, 4),
                                   "kernels_avg_ms": {k: round(s["total_ms"] / s["launches"], 4) for k, s in stats.items() if s["launches"]}}
    g.soe.close()

    ipr, share = out["in_process_ranks"], out["one_rank_share_alone"]
    halo_s = ipr["halo_bytes_per_neighbour"] / 50e9 + 10e-6
    per_it = share["ms_per_iteration"] * 1e-3 + ipr["exchanges_per_iteration"] * halo_s + ipr["allreduces_per_iteration"] * 20e-6
    out["projection_4_gpus"] = {
        "LABEL": "PROJECTION, not a measurement: RCCL at N > 1 has never run on hardware available to this build",
        "iterations": ipr["linear_iterations"], "seconds_per_iteration": round(per_it, 6), "solve_s": round(per_it * ipr["linear_iterations"], 4),
        "assumptions": {"halo_link_GBps": 50.0, "halo_latency_us": 10.0, "allreduce_latency_us": 20.0},
        "single_gpu_solve_s": out.get("single_handle", {}).get("step_s"),
        "speedup_over_one_gpu": (round(out["single_handle"]["step_s"] / (per_it * ipr["linear_iterations"]), 2) if "single_handle" in out else None)}
    print(json.dumps(out), flush=True)
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
