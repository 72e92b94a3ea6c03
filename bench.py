#!/usr/bin/env python3
"""bench.py -- Newton iterations/s and SpMV GB/s of the VSFM solve path on MI355X.

Workload (BASELINE.json `metric`): "celia-extruded" VSFM grid, 1024 x 1024 soil columns x 100
layers (104,857,600 cells, vertical-only connectivity), each column the reference's celia1990 column
(src/driver/standalone/vsfm/vsfm_celia1990_problem.F90) with a per-column permeability multiplier
exp(0.5 z), z ~ N(0,1), numpy default_rng(12345); Dirichlet top/bottom; dt = 3600 s; BiCGStab + per-column
block-Jacobi, ksp_rtol 1e-5, Newton rtol 1e-8 / stol 1e-10.  One "step" = one soe%StepDT call.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank process per GPU.  Either a launcher starts them (python -m torch.distributed.run --nproc-per-node N ...:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when bench.py is invoked directly, it starts the N rank
processes itself -- plain child processes, before anything in this process has touched the GPU -- waits for them with a
bound, and exits non-zero with the failing rank's stderr if any of them fails.  The columns are sharded across ranks
(strong scaling: the global grid is fixed); the only inter-GPU traffic is the RCCL all-reduce of the Newton/Krylov norms
and dot products inside libmppgpu.so.  Host coordination (unique-id broadcast, barrier, max-over-ranks) is a small TCP
group (mpp_amd.parallel.HostGroup).

--workload thermal / th: the other two BASELINE configurations through the same launcher, rank set-up and output line (see
run_thermal / run_th below): thermal n^3 heat diffusion (config 4; GMRES(30) + ILU(0), the mesh split into z-slabs with ghost
planes and an RCCL halo exchange) and the coupled thermal-hydrology columns (config 5; column-sharded like the VSFM grid).
--workload vsfm3d / --workload th --mesh 3d: the 7-point grids (SURVEY 8d's 3-D variant: v-channel soil and geometry); N > 1: ranks own
x-y blocks of WHOLE soil columns, the neighbours' edge columns as ghosts (mpp_amd.workloads.vchannel_rank builds a rank's block
without the global mesh).  The default workload, and its line, are the VSFM headline; at N = 1 its extra_configs carry a short run of the
7-point grid whose `roofline` is the CSR SpMV proper inside that run's own timed region.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
EVENT_STRIDE = 5       # every 5th launch of a kernel class is bracketed by HIP events (a pair costs ~10 us of queue time); 5 is
                       # co-prime with the periods (2, 3, 4) at which different kernels alternate inside a class


def kernel_source_hash():
    """sha256 (16 hex digits) of the kernel sources this run was built from: profiles/kernel_traffic.json is quoted only when
    its PMC passes were made with the same sources (tools/make_traffic.py)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mpp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def perm_multipliers(ncols_global):
    return np.exp(0.5 * np.random.default_rng(12345).standard_normal(ncols_global))


def build_problem(ncols_global, nz, col0, ncols_local, hetero=True):
    from mpp_amd import workloads as P
    mult = perm_multipliers(ncols_global)[col0:col0 + ncols_local] if hetero else None
    return P.celia1990(ncols=ncols_local, nz=nz, perm_mult=mult)


# ------------------------------------------------------------------------------------------------------------
# CPU legs (rank 0, N = 1): the oracle as baseline and as checker of the benchmark's own answer
# ------------------------------------------------------------------------------------------------------------
def usable_cores():
    """cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box exposes every core of
    the host but grants a share of them)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:   # cgroup v2
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]))))
    except (OSError, ValueError, IndexError):
        pass
    try:   # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ncols_global, nz, nsteps, warmup, cores, cols_per_core, ksp="gmres", pc="ilu"):
    """The CPU oracle ("port") on a bounded sample of the same workload: `cores` independent shards (the way
    MPP+ELM scales on CPUs: every MPI rank solves its own columns with its own serial SNES); reported as
    full-grid-equivalent Newton iterations/s.  ksp/pc: gmres+ilu = the reference's PETSc defaults, bcgs+bjacobi = the
    configuration the GPU run uses."""
    from mpp_amd import constants as C
    from mpp_amd import workloads as P
    from oracle import adapters as A
    from oracle import vsfm_oracle as VO
    VO.lib()
    rng_cols = np.random.default_rng(99).choice(ncols_global, size=cores * cols_per_core, replace=False)
    mult_all = perm_multipliers(ncols_global)
    shards = []
    for c in range(cores):
        cols = np.sort(rng_cols[c * cols_per_core:(c + 1) * cols_per_core])
        prob = P.celia1990(ncols=cols_per_core, nz=nz, perm_mult=mult_all[cols])
        o = A.to_oracle(prob, with_blocks=(pc == "bjacobi"))
        o.set_solver(C.KSP_BCGS if ksp == "bcgs" else C.KSP_GMRES, C.PC_BJACOBI if pc == "bjacobi" else C.PC_ILU0)
        shards.append((prob, o))
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
    return {"value": cell_its_per_s / n_full, "cores": cores, "solver": "%s+%s" % (ksp, pc), "seconds": round(dt, 2),
            "sample": "%d of %d columns (%d threads x %d columns x %d layers), %d timed steps after %d warm-up; "
                      "%.3g cell-Newton-updates/s scaled by 1/%d cells" %
                      (cores * cols_per_core, ncols_global, cores, cols_per_core, nz, nsteps, warmup, cell_its_per_s, n_full),
            "_shards": shards, "_work0": work}


def physical_cores():
    """physical cores of the host (sockets x cores per socket from /proc/cpuinfo; falls back to os.cpu_count() // 2)"""
    try:
        ids = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip() and phys is not None and core is not None:
                ids.add((phys, core)); phys = core = None
        if ids:
            return len(ids)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def extrapolate_cpu(value_threads, threads, value_one_core, gpu_value):
    """The north-star's ">= 10x all host cores" is judged against the whole host, but this process may only use a share of it (cgroup
    quota): the measured rate is scaled LINEARLY to every physical core -- generous to the CPU (memory bandwidth does not scale
    linearly) -- from the multi-thread measurement, and from the one-core measurement for comparison; arithmetic spelled out."""
    pc = physical_cores()
    per_thread = value_threads / max(threads, 1)
    a, b = per_thread * pc, value_one_core * pc
    return {"physical_cores": pc, "from_threads_measured": round(a, 4), "from_one_core": round(b, 4),
            "arithmetic": "%.4g it/s on %d threads / %d x %d physical cores = %.4g; one core %.4g x %d = %.4g (linear scaling: an upper bound "
                          "for a bandwidth-bound solve)" % (value_threads, threads, threads, pc, a, value_one_core, pc, b),
            "gpu_over_cpu": round(gpu_value / max(a, b), 1) if max(a, b) > 0 else None}


def cpu_one_core(shards, n_full):
    """the reference itself is serial (PETSC_COMM_SELF, one rank): the same measure on ONE core, two more steps of shard 0"""
    prob, o = shards[0]
    before = o.counters()["newton"]
    t0 = time.perf_counter()
    for s in range(2):
        o.step_dt(prob.dt, s + 1)
    dt = time.perf_counter() - t0
    return (o.counters()["newton"] - before) * prob.mesh.ncells_local / dt / n_full


def verify_answer(pressure, ncols_global, nz, col0, ncols_local, substeps, hetero, nsample=64):
    """The benchmark's own answer against the oracle: `nsample` columns of this rank's shard, each solved alone by the CPU
    oracle at tight Newton tolerances over the same sequence of (sub-)steps the GPU run accepted -- `substeps` is the list,
    per StepDT call, of the step sizes mppgpu_get_step_history reported: a global solve that cuts its time step applies the
    halved step to every column (SystemOfEquationsBaseType.F90:486-506), so the backward-Euler discretisation to compare
    with is that one.  Columns are independent, so a column's converged answer does not depend on its neighbours: what is
    left between the two is the Newton truncation of the GPU's global solve (rtol 1e-8 on the global norm).  Returns the
    largest pressure difference, absolute and relative to the largest pressure."""
    from mpp_amd import constants as C
    from mpp_amd import workloads as P
    from oracle import adapters as A
    mult = perm_multipliers(ncols_global) if hetero else np.ones(ncols_global)
    pick = np.sort(np.random.default_rng(7).choice(ncols_local, size=min(nsample, ncols_local), replace=False))
    worst, scale, oracle_cuts = 0.0, 0.0, 0
    # one oracle problem per column: each column converges by itself, as tightly as the arithmetic allows
    for c in pick:
        one = P.celia1990(ncols=1, nz=nz, perm_mult=mult[col0 + c:col0 + c + 1])
        o = A.to_oracle(one, with_blocks=False)
        o.set_solver(C.KSP_GMRES, C.PC_ILU0, ksp_rtol=1e-12, snes_rtol=1e-13, snes_stol=1e-14)
        k = 0
        for subs in substeps:
            for dt_sub in subs:
                k += 1
                o.step_dt(dt_sub, k)
        oracle_cuts += o.counters()["time_cuts"]
        po = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
        pg = pressure[c * nz:(c + 1) * nz]
        worst = max(worst, float(np.abs(pg - po).max()))
        scale = max(scale, float(np.abs(po).max()))
    return {"columns_checked": int(pick.size), "max_abs_err_Pa": worst, "max_rel_err": worst / scale,
            "oracle_time_step_cuts": int(oracle_cuts),
            "against": "CPU oracle, one column at a time, snes_rtol 1e-13 / stol 1e-14, the %d (sub-)steps the GPU run accepted"
                       % sum(len(x) for x in substeps)}


# ------------------------------------------------------------------------------------------------------------
# rank launcher: `python bench.py --gpus N` invoked directly
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(nranks, argv, timeout_s):
    """Start `nranks` copies of this script as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* exported: the
    contract of torch.distributed.run), pass rank 0's stdout through, and return the exit code: 0 when every rank exits 0;
    otherwise a failing rank's code, after the stderr of every failed rank has been shown and the ranks still waiting for
    them have been stopped (124 when the time limit ends the run).  This process never initialises the GPU."""
    port = _free_port()
    procs, logs = [], []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MPPGPU_BENCH_RANK_PROCESS="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env.get("NCCL_DEBUG") and not env.get("NCCL_DEBUG_FILE"):
            env["NCCL_DEBUG_FILE"] = "/dev/stderr"   # RCCL logs to stdout by default: keep rank 0's stdout for the ONE JSON line
        err = tempfile.TemporaryFile(mode="w+")
        logs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL), stderr=err))
    deadline = time.time() + timeout_s
    first_bad, timed_out = None, False
    while True:
        codes = [p.poll() for p in procs]
        if first_bad is None and any(c not in (None, 0) for c in codes):
            first_bad = time.time()
        if all(c is not None for c in codes):
            break
        # a rank failed: the ranks that depend on it usually fail by themselves within moments (closed sockets); give them
        # that long, so that what is reported below are exit codes of their own making
        if first_bad is not None and time.time() > first_bad + 2.0:
            break
        if time.time() > deadline:
            timed_out = True
            break
        time.sleep(0.1)
    codes = [p.poll() for p in procs]
    stopped = [r for r, c in enumerate(codes) if c is None]
    if stopped:
        # blocked in a collective a failed rank will never join (or over the time limit): stop exactly the processes started here
        for r in stopped:
            procs[r].terminate()
        grace = time.time() + 5.0
        for r in stopped:
            while procs[r].poll() is None and time.time() < grace:
                time.sleep(0.1)
            if procs[r].poll() is None:
                procs[r].kill()
                procs[r].wait()
    failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
    rc = 0
    if timed_out:
        rc = 124
        sys.stderr.write("bench.py: rank(s) %s still running after %.0f s (launcher time limit); stopped\n" % (stopped, timeout_s))
    elif failed:
        rc = codes[failed[0]] if codes[failed[0]] > 0 else 1
    for r, f in enumerate(logs):
        f.seek(0)
        text = f.read()
        f.close()
        if r in failed:
            sys.stderr.write("bench.py: rank %d exited with code %d; its stderr:\n%s\n" % (r, codes[r], text[-4000:]))
        elif (timed_out and r in stopped) or (rc == 0 and text.strip()):
            sys.stderr.write("".join("[rank %d] %s\n" % (r, l) for l in text.splitlines()[-20:]))
    if failed and stopped:
        sys.stderr.write("bench.py: stopped rank(s) %s, which were waiting for the failed one(s)\n" % stopped)
    return rc


class Watchdog:
    """ends THIS rank process when a phase that involves the other ranks (rendezvous, ncclCommInitRank, the first solve
    with its first all-reduces) does not finish in time -- a rank that never arrives must not leave the others waiting
    forever; the launcher (or torch.distributed.run) then stops the remaining ranks"""

    def __init__(self, seconds, what, rank):
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.what, self.rank, self.seconds = what, rank, seconds
        self.t.start()

    def _fire(self):
        sys.stderr.write("bench.py: rank %d: %s did not finish within %.0f s; giving up\n" % (self.rank, self.what, self.seconds))
        sys.stderr.flush()
        os._exit(3)

    def done(self):
        self.t.cancel()


def dry_run_rank(args, group, rank, world):
    """the multi-rank host path without any GPU call (CPU tests): rendezvous, shard, id broadcast, barrier, reductions"""
    from mpp_amd import parallel as PAR
    if args.workload == "thermal":   # cells in contiguous id ranges = z-slabs; ghost planes next to the neighbours
        ncols_global = args.n ** 3
        col0, ncols_local = PAR.shard_columns(ncols_global, world, rank)
    else:
        ncols_global = args.nx * args.ny
        col0, ncols_local = PAR.shard_columns(ncols_global, world, rank)
    if args.fail_rank == rank:
        raise SystemExit("bench.py: rank %d: injected failure (--fail-rank)" % rank)
    if args.fail_rank == -2 and rank == world - 1:   # a rank that never joins the first collective
        time.sleep(3600)
    extra = {}
    if args.workload == "thermal":   # the rank's slab as the GPU run builds it (numpy only): owned cells, ghost planes, halo plan
        from mpp_amd import workloads as P
        part, tp = P.thermal_mms_rank(args.n, args.n, args.n, world, rank)
        sent = float(part.send_ptr[-1]) if len(part.send_ptr) else 0.0
        tot = group.allreduce([float(part.nowned), float(part.nghost), sent, float(tp.bc.num_connections)], op="sum")
        extra = {"owned_cells": tot[0], "ghost_cells": tot[1], "halo_values_sent": tot[2], "boundary_faces": tot[3]}
    elif args.workload == "vsfm3d" or (args.workload == "th" and args.mesh == "3d"):
        # 7-point grids: a rank owns an x-y block of WHOLE soil columns (rows of the grid), its ghosts are the neighbours' edge columns
        from mpp_amd import workloads as P
        build = P.vchannel_rank if args.workload == "vsfm3d" else P.th_vchannel_rank
        part, lp = build(args.nx, args.ny, args.nz, world, rank)
        sent = float(part.send_ptr[-1]) if len(part.send_ptr) else 0.0
        fields = 2.0 if args.workload == "th" else 1.0
        tot = group.allreduce([float(part.nowned), float(part.nghost), sent, float(part.ncols_owned), float(part.nghost) / args.nz], op="sum")
        first_col = group.allreduce([float(part.l2g[0] % (args.nx * args.ny)) if r == rank else 0.0 for r in range(world)], op="sum")
        per_nb = group.allreduce([float(max(part.recv_count)) if len(part.recv_count) else 0.0], op="max")[0]
        extra = {"cells": tot[0], "owned_cells": tot[0], "ghost_cells": tot[1], "halo_values_sent": tot[2], "columns": tot[3],
                 "ghost_columns": tot[4], "first_column_of_rank": first_col, "mesh": "3d", "partition": getattr(part, "mode", "?"),
                 "halo_bytes_per_neighbour_per_exchange": per_nb * 8.0 * fields,
                 "unknowns": fields * tot[0]}
        ncols_local = part.ncols_owned
    elif args.workload == "th":
        from mpp_amd import workloads as P
        tp = P.th_columns(ncols=ncols_local, nz=args.nz, perm_mult=perm_multipliers(ncols_global)[col0:col0 + ncols_local])
        tot = group.allreduce([float(tp.mesh.ncells_local)], op="sum")
        extra = {"cells": tot[0], "unknowns": 2 * tot[0]}
    uid = group.broadcast_bytes(bytes(range(128)) if rank == 0 else b"")
    group.barrier()
    elapsed = group.allreduce([1.0 + rank], op="max")[0]
    total_cols = group.allreduce([float(ncols_local)], op="sum")[0]
    gathered = group.allreduce([float(col0) if r == rank else 0.0 for r in range(world)], op="sum")
    if rank == 0:
        line = {"dry_run": True, "workload": args.workload, "n_gpus": world, "elapsed_max": elapsed, "columns": total_cols,
                "first_column_of_rank": gathered, "unique_id_ok": uid == bytes(range(128))}
        if args.workload == "thermal":
            line["cells"] = line.pop("columns"); line["first_cell_of_rank"] = line.pop("first_column_of_rank")
        line.update(extra)
        print(json.dumps(line), flush=True)
    group.barrier()
    group.close()


def extra_config(label, nx, ny, nz, elm_default, steps, warmup, opts, device_id):
    """A short timed run of another configuration of the VSFM path, behind the headline's timed region (N = 1): the same grid family
    and solver, reported next to the headline because the headline is the friendliest case of the kernels (one soil class, van
    Genuchten m = 1/2).  elm_default: every cell its own soil parameters (no soil classes) and the smoothed Brooks-Corey function
    `smooth_brooks_corey_bz3`, ELM's default (mpp_varctl) -- what the march costs in E3SM."""
    from mpp_amd import constants as C
    from mpp_amd import workloads as P
    ncols = nx * ny
    prob = build_problem(ncols, nz, 0, ncols)
    if elm_default:
        n = prob.alpha.size
        rng = np.random.default_rng(2024)
        prob.satfunc_type = np.full(n, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3, dtype=np.int32)
        prob.alpha = prob.alpha * (1.0 + 0.05 * rng.random(n))
        prob.sat_res = prob.sat_res * (1.0 + 0.05 * rng.random(n))
        prob.lam = prob.lam * (1.0 + 0.05 * rng.random(n))
    t0 = time.perf_counter()
    mpp = P.to_gpu(prob, opts, device_id=device_id)
    soe = mpp.soe
    t_setup = time.perf_counter() - t0

    def step(i):
        for b, (_t, _cs, vals) in enumerate(prob.bcs):
            soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, b + 1, vals)
        conv, reason = soe.StepDT(prob.dt, i + 1)
        if not conv:
            raise RuntimeError("StepDT did not converge (reason %d)" % reason)

    try:
        for i in range(warmup):
            step(i)
        soe.SetOption("profile_stride", EVENT_STRIDE)
        soe.SetOption("profile", 1)
        soe.reset_stats()
        c0 = soe.counters()
        soe.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        soe.synchronize()
        elapsed = time.perf_counter() - t0
        c1 = soe.counters()
        newton = c1["newton"] - c0["newton"]
        layout = soe.assembly_info()
        kernels = class_stats(soe, elapsed, EVENT_STRIDE)
        for k, v in kernels.items():
            if "GBps_algorithmic" in v:
                v["frac_of_peak_algorithmic"] = round(v["GBps_algorithmic"] / HBM_PEAK_GBS, 4)
        return {"config": label, "cells": ncols * nz, "steps": steps, "warmup": warmup, "newton_iterations": newton,
                "value": newton / elapsed, "unit": "Newton iters/s", "ms_per_newton_iteration": elapsed / max(newton, 1) * 1e3,
                "time_step_cuts": c1["time_cuts"] - c0["time_cuts"], "setup_s": round(t_setup, 2),
                "soil_classes": layout["soil_classes"], "level_geometry": bool(layout["level_geometry"]),
                "complete_row_march": bool(layout["complete_rows"]), "kernels": kernels}
    finally:
        soe.close()


def extra_vsfm3d(nx, ny, nz, steps, warmup, ksp, pc, device_id):
    """The 7-point VSFM grid (SURVEY 8d's 3-D variant: v-channel soil and geometry) behind the headline's timed region (N = 1): the one
    VSFM workload where BiCGStab runs tens of iterations per Newton step, i.e. where the CSR SpMV proper -- values + index + gathered
    x, y written -- is the hot loop; its `roofline` is that kernel inside this run's own timed region (vsfm3d_measure)."""
    from mpp_amd import workloads as P
    t0 = time.perf_counter()
    part, prob = P.vchannel_rank(nx, ny, nz, 1, 0)
    g = P.to_gpu(prob, {"ksp_type": ksp, "pc_type": pc}, device_id=device_id)
    t_setup = time.perf_counter() - t0
    try:
        m = vsfm3d_measure(g.soe, prob, steps, warmup, nx, ny, nz, part.nowned)
        layout = g.soe.assembly_info()
        m.update({"config": "vsfm3d: VSFM v-channel 7-point grid %dx%dx%d, %s+%s" % (nx, ny, nz, ksp, pc), "cells": nx * ny * nz,
                  "setup_s": round(t_setup, 2), "spmv_variant": layout["spmv_variant"],
                  "columns_plus_lateral_assembly": bool(layout.get("columns_plus_lateral_assembly"))})
        return m
    finally:
        g.soe.close()


def pick_device(args, local_rank):
    if not args.share_gpus:
        return local_rank
    import ctypes
    ndev = ctypes.c_int(0)
    ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(ndev))
    return local_rank % max(ndev.value, 1)


def init_comm(soe, group, rank, world):
    """the RCCL communicator of a launched run (unique id from rank 0 over the host group); RCCL prints a version banner on stdout
    at communicator creation: stdout is kept for the ONE JSON line"""
    import ctypes
    from mpp_amd import vsfm as V
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


def class_stats(soe, elapsed, stride):
    out = {}
    for k in ("spmv", "spmv_pc", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce"):
        s = soe.kernel_stats(k)
        if s["launches"]:
            ms = s["total_ms"] / s["launches"]
            out[k] = {"launches_timed": s["launches"], "launches": s["calls"], "avg_ms": round(ms, 5),
                      "share_of_step_time": round(ms * s["calls"] * 1e-3 / elapsed, 4)}
            if s["bytes_per_launch"] > 0:
                out[k]["GBps_algorithmic"] = round(s["bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
    return out


def run_thermal(args, group, rank, world, local_rank, launched, dog):
    """BASELINE config 4: 3-D heat diffusion (the reference's thermal_mms_3d problem: T = 10 sin(pi x) cos(2 pi y) sin(3 pi z) + 270,
    lambda = exp(x + y + z - 1), regression_tests/thermal) on n^3 cells, GMRES(30) + ILU(0) = PETSc's defaults.  One step = one
    soe%StepDT = operator assembly + one KSP solve (SOEBaseStepDT_KSP, SystemOfEquationsBaseType.F90:555-647).  N > 1: the cells
    are split into contiguous id ranges (z-slabs) with one ghost plane per neighbour; halo exchange (grouped RCCL send / recv)
    before every operator assembly and SpMV, rank-local ILU(0) = PETSc's parallel block-Jacobi / ILU(0), all-reduced dots."""
    from mpp_amd import workloads as P
    n = args.n
    t_setup = time.perf_counter()
    part, tp = P.thermal_mms_rank(n, n, n, world, rank)
    opts = {"ksp_type": args.ksp, "pc_type": args.pc, "ksp_rtol": args.ksp_rtol}
    opts.update(kv.split("=", 1) for kv in args.opt)
    g = P.to_thermal_gpu(tp, opts, device_id=pick_device(args, local_rank))
    soe = g.soe
    if launched:
        init_comm(soe, group, rank, world)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        soe.synchronize()
        group.barrier()

    def step():
        soe.PreStepDT()                     # soln <- soln_prev_clm: every step solves the same problem
        conv, _ = soe.StepDT(tp.dt, 1)
        if not conv:
            raise RuntimeError("the KSP solve did not converge")
        return soe.last_linear_iterations

    for _ in range(args.warmup):
        step()
    if dog:
        dog.done()
    soe.SetOption("profile_stride", EVENT_STRIDE)
    soe.SetOption("profile", 1 if args.profile in ("all", "auto") and world == 1 else 0)
    soe.reset_stats()
    barrier()
    t0 = time.perf_counter()
    its = 0
    for _ in range(args.steps):
        its += step()
    barrier()
    elapsed = group.allreduce([time.perf_counter() - t0], op="max")[0]
    err = float(np.abs(soe.GetSoln()[:part.nowned] - tp.exact[:part.nowned]).max())
    err = group.allreduce([err], op="max")[0]
    info = soe.pc_info() if args.pc == "ilu" else {"path": "none", "launches": 0, "box_extents": (0, 0, 0)}
    kernels = class_stats(soe, elapsed, EVENT_STRIDE)
    soe.SetOption("profile", 0)
    if rank == 0:
        N = n ** 3
        nnz = 7 * N - 6 * n * n
        out = {"metric": "KSP iterations/sec (thermal %d^3, %s+%s)" % (n, args.ksp, args.pc), "value": its / elapsed, "unit": "KSP iterations/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "thermal_mms_3d heat diffusion %d^3 cells, %s+%s, ksp_rtol %g" % (n, args.ksp, args.pc, args.ksp_rtol),
                          "cells": N, "cells_per_gpu": part.nowned, "ghost_cells_rank0": part.nghost,
                          "parallelism": "cells in contiguous id ranges (z-slabs) over %d GPU(s), RCCL halo exchange + all-reduce" % world,
                          "ilu_path": info["path"], "ilu_launches_per_application": info["launches"], "box_extents": list(info["box_extents"])},
               "linear_iterations": its, "ms_per_iteration": elapsed / max(its, 1) * 1e3, "setup_s": round(t_setup, 2),
               "max_error_vs_manufactured_solution_K": err, "kernels": kernels}
        pa = kernels.get("pc_apply")
        if pa and world == 1:
            by = 12.0 * nnz + 24.0 * N
            ach = by / (pa["avg_ms"] * 1e-3) / 1e9
            out["roofline"] = {"kernel": "ILU(0) application (both triangular solves by boxes)", "bound": "hbm", "achieved": round(ach, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                               "achieved_is": "algorithmic bytes of an ILU(0) application on CSR factors (12 nnz + 24 N, SURVEY 8d) / HIP-event average",
                               "bytes_per_launch": by, "avg_launch_ms": pa["avg_ms"], "launches_timed": pa["launches_timed"]}
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


def run_th(args, group, rank, world, local_rank, launched, dog):
    """BASELINE config 5: coupled thermal-hydrology (Richards + enthalpy equation, unknowns [P; T], SOETHResidual / SOETHJacobian:
    SystemOfEquationsTHType.F90:736-1003) on nx x ny celia soil columns x nz layers, column-sharded across the GPUs like the VSFM
    grid (no neighbour data; RCCL all-reduce of the Krylov / Newton scalars).  One step = one soe%StepDT."""
    from mpp_amd import parallel as PAR
    from mpp_amd import workloads as P
    ncols_global = args.nx * args.ny
    col0, ncols = PAR.shard_columns(ncols_global, world, rank)
    t_setup = time.perf_counter()
    mult = None if args.homogeneous else perm_multipliers(ncols_global)[col0:col0 + ncols]
    tp = P.th_columns(ncols=ncols, nz=args.nz, perm_mult=mult)
    opts = {"ksp_type": args.ksp, "pc_type": args.pc, "ksp_rtol": args.ksp_rtol}
    opts.update(kv.split("=", 1) for kv in args.opt)
    g = P.to_th_gpu(tp, opts, device_id=pick_device(args, local_rank))
    soe = g.soe
    if launched:
        init_comm(soe, group, rank, world)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        soe.synchronize()
        group.barrier()

    def step(i):
        P.th_gpu_set_conditions(tp, g)
        conv, reason = soe.StepDT(tp.dt, i + 1)
        if not conv:
            raise RuntimeError("StepDT did not converge (reason %d)" % reason)

    for i in range(args.warmup):
        step(i)
    if dog:
        dog.done()
    soe.SetOption("profile_stride", EVENT_STRIDE)
    soe.SetOption("profile", 1 if args.profile in ("all", "auto") and world == 1 else 0)
    soe.reset_stats()
    c0 = soe.counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = group.allreduce([time.perf_counter() - t0], op="max")[0]
    c1 = soe.counters()
    newton = c1["newton"] - c0["newton"]
    kernels = class_stats(soe, elapsed, EVENT_STRIDE)
    soe.SetOption("profile", 0)
    if rank == 0:
        n_global = ncols_global * args.nz
        out = {"metric": "Newton iters/sec (TH %dx%dx%d)" % (args.nx, args.ny, args.nz), "value": newton / elapsed, "unit": "Newton iters/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "coupled thermal-hydrology, %dx%d celia columns x %d layers, 2x2-block Jacobian, %s+%s"
                                      % (args.nx, args.ny, args.nz, args.ksp, args.pc),
                          "cells": n_global, "unknowns": 2 * n_global, "cells_per_gpu": ncols * args.nz,
                          "parallelism": "columns sharded over %d GPU(s), RCCL all-reduce of Krylov/Newton scalars" % world},
               "newton_iterations": newton, "linear_iterations": c1["linear"] - c0["linear"], "time_step_cuts": c1["time_cuts"] - c0["time_cuts"],
               "setup_s": round(t_setup, 2), "kernels": kernels}
        jk = kernels.get("jacobian")
        if jk and world == 1:
            # algorithmic bytes per cell of the fused column assembly: x 16 + soil / heat parameters 80 + accum_prev 16 + connection
            # geometry 56 read, F 16 + 12 Jacobian entries 96 written = 280 B
            by = 280.0 * ncols * args.nz
            ach = by / (jk["avg_ms"] * 1e-3) / 1e9
            out["roofline"] = {"kernel": "coupled column assembly (residual + 2x2-block Jacobian)", "bound": "hbm", "achieved": round(ach, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                               "achieved_is": "280 B per cell (DESIGN.md section 10) / HIP-event average launch time",
                               "bytes_per_launch": by, "avg_launch_ms": jk["avg_ms"], "launches_timed": jk["launches_timed"]}
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


def vsfm3d_measure(soe, prob, steps, warmup, nx, ny, nz, n_owned, group=None, dog=None, profile=1):
    """times `steps` StepDT calls of a 7-point VSFM problem after `warmup`; returns the pieces of a bench line.  The roofline object
    is the CSR SpMV proper (k_spmv_* on 7-entry rows: values + index + x gathered + y written) INSIDE this timed region -- here
    BiCGStab runs tens of iterations per Newton step, so the product is the hot loop (12 nnz + 20 N + 4 bytes per launch)."""
    from mpp_amd import constants as C

    def barrier():
        soe.synchronize()
        if group is not None:
            group.barrier()

    def step(i):
        conv, reason = soe.StepDT(prob.dt, i + 1)
        if not conv:
            raise RuntimeError("StepDT did not converge (reason %d)" % reason)

    mass0 = float(np.sum(soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_MASS, -1)[:n_owned]))
    for i in range(warmup):
        step(i)
    if dog:
        dog.done()
    soe.SetOption("profile_stride", EVENT_STRIDE)
    soe.SetOption("profile", profile)
    soe.reset_stats()
    c0 = soe.counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    if group is not None:
        elapsed = group.allreduce([elapsed], op="max")[0]
    c1 = soe.counters()
    comm = soe.comm_stats()
    kernels = class_stats(soe, elapsed, EVENT_STRIDE)
    sp = soe.kernel_stats("spmv")
    soe.SetOption("profile", 0)
    mass1 = float(np.sum(soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_MASS, -1)[:n_owned]))
    if group is not None:
        mass0, mass1 = group.allreduce([mass0, mass1], op="sum")
    newton, linear = c1["newton"] - c0["newton"], c1["linear"] - c0["linear"]
    out = {"value": newton / elapsed, "unit": "Newton iters/s", "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
           "newton_iterations": newton, "linear_iterations": linear, "linear_per_newton": round(linear / max(newton, 1), 2),
           "ms_per_newton_iteration": elapsed / max(newton, 1) * 1e3, "time_step_cuts": c1["time_cuts"] - c0["time_cuts"],
           "halo_exchanges": comm["halo_exchanges"], "allreduces": comm["allreduces"],
           # no boundary condition, no source: the water in the domain is conserved (a size-independent check of the whole step)
           "answer_check": {"property": "total water mass conserved over the %d steps (no boundary, no source)" % (steps + warmup),
                            "mass_kg_before": mass0, "mass_kg_after": mass1, "relative_drift": abs(mass1 - mass0) / abs(mass0)},
           "kernels": kernels}
    if sp["launches"]:
        ms = sp["total_ms"] / sp["launches"]
        ach = sp["bytes_per_launch"] / (ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "csr_spmv_fp64 on 7-point rows (values + index + gathered x + y written), inside this timed region",
                           "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(min(ach / HBM_PEAK_GBS, 1.0), 4), "traffic": None,
                           "achieved_is": "algorithmic CSR bytes (12 nnz + 20 N + 4, SURVEY 8d) / HIP-event average launch time",
                           "bytes_per_launch": sp["bytes_per_launch"], "avg_launch_ms": round(ms, 5), "launches_timed": sp["launches"],
                           "launches": sp["calls"]}
    return out


def run_vsfm3d(args, group, rank, world, local_rank, launched, dog):
    """SURVEY 8(d)'s 7-point VSFM variant: the v-channel soil and geometry (dx = dy = 10 m, dz = 0.5 m, hydrostatic start, no boundary
    condition) on nx x ny x nz cells.  N > 1: ranks own x-y blocks of WHOLE soil columns (MeshType.F90:293-460), ghosts are
    the neighbours' edge columns, halo exchange (nz x nx cells per neighbour) before every assembly pass and SpMV; the per-column
    block-Jacobi preconditioner is the same operator for every N, so are the iteration counts."""
    from mpp_amd import workloads as P
    t_setup = time.perf_counter()
    part, prob = P.vchannel_rank(args.nx, args.ny, args.nz, world, rank)
    opts = {"ksp_type": args.ksp, "pc_type": args.pc, "ksp_rtol": args.ksp_rtol}
    opts.update(kv.split("=", 1) for kv in args.opt)
    g = P.to_gpu(prob, opts, device_id=pick_device(args, local_rank))
    soe = g.soe
    if launched:
        init_comm(soe, group, rank, world)
    t_setup = time.perf_counter() - t_setup
    m = vsfm3d_measure(soe, prob, args.steps, args.warmup, args.nx, args.ny, args.nz, part.nowned, group=group if launched else None, dog=dog,
                       profile=1 if args.profile in ("all", "auto") and world == 1 else 2)
    layout = soe.assembly_info()
    if rank == 0:
        N = args.nx * args.ny * args.nz
        out = {"metric": "Newton iters/sec (VSFM 7-point %dx%dx%d)" % (args.nx, args.ny, args.nz), "value": m.pop("value"), "unit": m.pop("unit"),
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m.pop("ms_per_step"),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "VSFM v-channel 7-point grid %dx%dx%d, %s+%s" % (args.nx, args.ny, args.nz, args.ksp, args.pc),
                          "cells": N, "cells_per_gpu": part.nowned, "ghost_cells_rank0": part.nghost,
                          "halo_bytes_per_neighbour_per_exchange": int(max(part.recv_count)) * 8 if len(part.recv_count) else 0,
                          "halo_stats_rank0": soe.halo_stats(),
                          "column_segments": layout["column_segments"], "spmv_variant": layout["spmv_variant"],
                          "parallelism": "x-y blocks of whole soil columns over %d GPU(s), RCCL halo exchange + all-reduce" % world},
               "setup_s": round(t_setup, 2)}
        out.update(m)
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


def run_th3d(args, group, rank, world, local_rank, launched, dog):
    """the coupled thermal-hydrology system on the 7-point v-channel grid (workloads.th_vchannel), ranks owning whole soil columns with
    ghost columns; GMRES(30) + rank-local ILU(0) unless --ksp / --pc say otherwise.  One step = one soe%StepDT."""
    from mpp_amd import constants as C
    from mpp_amd import workloads as P
    t_setup = time.perf_counter()
    part, tp = P.th_vchannel_rank(args.nx, args.ny, args.nz, world, rank)
    opts = {"ksp_type": args.ksp, "pc_type": args.pc, "ksp_rtol": args.ksp_rtol}
    opts.update(kv.split("=", 1) for kv in args.opt)
    g = P.to_th_gpu(tp, opts, device_id=pick_device(args, local_rank))
    soe = g.soe
    if launched:
        init_comm(soe, group, rank, world)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        soe.synchronize()
        group.barrier()

    def step(i):
        P.th_gpu_set_conditions(tp, g)
        conv, reason = soe.StepDT(tp.dt, i + 1)
        if not conv:
            raise RuntimeError("StepDT did not converge (reason %d)" % reason)

    for i in range(args.warmup):
        step(i)
    if dog:
        dog.done()
    soe.SetOption("profile_stride", EVENT_STRIDE)
    soe.SetOption("profile", 1 if args.profile in ("all", "auto") and world == 1 else 0)
    soe.reset_stats()
    c0 = soe.counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = group.allreduce([time.perf_counter() - t0], op="max")[0]
    c1 = soe.counters()
    comm = soe.comm_stats()
    kernels = class_stats(soe, elapsed, EVENT_STRIDE)
    soe.SetOption("profile", 0)
    newton = c1["newton"] - c0["newton"]
    if rank == 0:
        N = args.nx * args.ny * args.nz
        out = {"metric": "Newton iters/sec (TH 7-point %dx%dx%d)" % (args.nx, args.ny, args.nz), "value": newton / elapsed, "unit": "Newton iters/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "coupled thermal-hydrology on the v-channel 7-point grid %dx%dx%d, 2x2-block Jacobian, %s+%s"
                                      % (args.nx, args.ny, args.nz, args.ksp, args.pc),
                          "cells": N, "unknowns": 2 * N, "cells_per_gpu": part.nowned, "ghost_cells_rank0": part.nghost,
                          "halo_bytes_per_neighbour_per_exchange": int(max(part.recv_count)) * 16 if len(part.recv_count) else 0,
                          "parallelism": "x-y blocks of whole soil columns over %d GPU(s), RCCL halo exchange + all-reduce" % world},
               "newton_iterations": newton, "linear_iterations": c1["linear"] - c0["linear"], "time_step_cuts": c1["time_cuts"] - c0["time_cuts"],
               "halo_exchanges": comm["halo_exchanges"], "setup_s": round(t_setup, 2), "kernels": kernels}
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["vsfm", "vsfm3d", "thermal", "th"], default="vsfm",
                    help="vsfm: the headline (BASELINE configs 2 / 3); vsfm3d: the 7-point VSFM grid of SURVEY 8(d) (v-channel soil and "
                         "geometry, lateral + vertical connections; ranks own x-y blocks of whole columns); thermal: n^3 heat diffusion, "
                         "GMRES(30) + ILU(0) (config 4); th: coupled thermal-hydrology columns (config 5; --mesh 3d: on the 7-point grid)")
    ap.add_argument("--mesh", choices=["columns", "3d"], default="columns",
                    help="th: independent soil columns (config 5) or the 7-point v-channel grid split by whole columns with ghost columns")
    ap.add_argument("--nx", type=int, default=0, help="columns in x (default 1024; th: 512)")
    ap.add_argument("--ny", type=int, default=0, help="columns in y (default 1024; th: 512)")
    ap.add_argument("--nz", type=int, default=0, help="layers (default 100; th: 80)")
    ap.add_argument("--n", type=int, default=512, help="thermal: the grid is n x n x n")
    ap.add_argument("--ksp", default="", help="default: bcgs (vsfm, th), gmres (thermal)")
    ap.add_argument("--pc", default="", help="default: bjacobi (vsfm, th), ilu (thermal)")
    ap.add_argument("--ksp-rtol", type=float, default=1e-5)
    ap.add_argument("--spmv-variant", type=int, default=-1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of sampled columns with the CPU oracle")
    ap.add_argument("--cpu-cores", type=int, default=0, help="threads of the CPU baseline (default: every core of the box)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short runs of ELM's default constitutive set-up and of 256 x 256 x 50 behind the timed region (N = 1)")
    ap.add_argument("--homogeneous", action="store_true")
    ap.add_argument("--profile", choices=["auto", "all", "spmv"], default="auto",
                    help="kernel classes timed with HIP events inside the timed region: every class, or the roofline kernel "
                         "(CSR SpMV) only -- the events of every class cost 5-6 %% at 13 M cells per GPU, under 1 %% at 105 M; "
                         "auto = all on one GPU, spmv on several")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra mppgpu_set_option pairs")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="direct --gpus N launch: time limit of the rank processes [s]")
    ap.add_argument("--init-timeout", type=float, default=600.0,
                    help="multi-rank: time limit for rendezvous + ncclCommInitRank + the first solve (first all-reduces) [s]")
    ap.add_argument("--share-gpus", action="store_true",
                    help="multi-rank: map LOCAL_RANK onto the visible devices modulo their number (a one-GPU box then reaches "
                         "ncclCommInitRank, which refuses two ranks on one device: the whole path up to RCCL, exercised)")
    ap.add_argument("--dry-run-ranks", action="store_true", help="multi-rank host path only, no GPU call (CPU tests)")
    ap.add_argument("--fail-rank", type=int, default=-1, help="with --dry-run-ranks: this rank exits with an error")
    args = ap.parse_args()
    th_like = args.workload == "th"
    three_d = args.workload == "vsfm3d" or (th_like and args.mesh == "3d")
    args.nx = args.nx or (256 if three_d and th_like else 512 if th_like or three_d else 1024)
    args.ny = args.ny or (256 if three_d and th_like else 512 if th_like or three_d else 1024)
    args.nz = args.nz or (80 if th_like else 100)
    gmres_ilu = args.workload == "thermal" or (th_like and three_d)      # (the coupled column preconditioner needs a vertical-only grid)
    args.ksp = args.ksp or ("gmres" if gmres_ilu else "bcgs")
    args.pc = args.pc or ("ilu" if gmres_ilu else "bjacobi")

    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if not launched and args.gpus > 1:
        # invoked directly: become the launcher.  Nothing above or below this line has touched the GPU in this process.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    # Host-side rendezvous over plain TCP (mpp_amd.parallel.HostGroup) from the RANK / WORLD_SIZE / MASTER_* variables.
    # torch itself is not imported: a rank process then loads exactly one ROCm runtime and one RCCL (those of the ROCm
    # installation), the configuration the GPU tests run.
    from mpp_amd import parallel as PAR
    dog = Watchdog(args.init_timeout, "rendezvous / communicator set-up / first solve", rank) if launched and world > 1 else None
    group = PAR.HostGroup(timeout=args.init_timeout) if launched else PAR.HostGroup(rank=0, world=1)
    if args.dry_run_ranks:
        dry_run_rank(args, group, rank, world)
        if dog:
            dog.done()
        return

    if args.workload != "vsfm":
        run = {"thermal": run_thermal, "th": run_th3d if args.mesh == "3d" else run_th, "vsfm3d": run_vsfm3d}[args.workload]
        run(args, group, rank, world, local_rank, launched, dog)
        return

    # the HIP extension (fails loudly if missing)
    from mpp_amd import constants as C
    from mpp_amd import vsfm as V
    from mpp_amd import workloads as P

    ncols_global = args.nx * args.ny
    col0, ncols_local = PAR.shard_columns(ncols_global, world, rank)
    t_setup = time.perf_counter()
    prob = build_problem(ncols_global, args.nz, col0, ncols_local, hetero=not args.homogeneous)
    opts = {"ksp_type": args.ksp, "pc_type": args.pc, "spmv_variant": args.spmv_variant}
    opts.update(kv.split("=", 1) for kv in args.opt)
    mpp = P.to_gpu(prob, opts, device_id=pick_device(args, local_rank))
    soe = mpp.soe
    if launched:
        init_comm(soe, group, rank, world)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        soe.synchronize()
        group.barrier()

    substeps = []   # per StepDT call: the sub-steps it accepted ([dt] unless the time step was cut); read outside the timed region

    def step(i):
        for b, (_t, _cs, vals) in enumerate(prob.bcs):
            soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, b + 1, vals)
        conv, reason = soe.StepDT(prob.dt, i + 1)
        if not conv:
            raise RuntimeError("StepDT did not converge (reason %d)" % reason)

    for i in range(args.warmup):
        step(i)
        substeps.append(soe.step_history())
    if dog and args.warmup > 0:
        dog.done()
        dog = None
    soe.SetOption("profile_stride", EVENT_STRIDE)
    soe.SetOption("profile", 1 if (args.profile == "all" or (args.profile == "auto" and world == 1)) else 2)
    soe.reset_stats()
    c0 = soe.counters()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
        substeps.append(soe.step_history())   # a host-side copy of a few doubles (no device work)
        if dog:
            dog.done()
            dog = None
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = group.allreduce([elapsed], op="max")[0]
    c1 = soe.counters()
    newton = c1["newton"] - c0["newton"]
    linear = c1["linear"] - c0["linear"]
    comm = soe.comm_stats()
    layout = soe.assembly_info()

    stats = {k: soe.kernel_stats(k) for k in ("spmv", "spmv_pc", "pc_apply", "pc_setup", "residual", "jacobian", "vec", "reduce")}
    soe.SetOption("profile", 0)
    # the plain CSR SpMV kernel (y = A x) by itself, right behind the timed region: where M^{-1} A v runs as one kernel (vertical-only
    # grids) the only CSR SpMV launches left inside a Newton iteration are the line search's (same matrix pass, f in instead of y
    # out, three reductions): the roofline object is measured on those, live; this is the reference point next to it
    plain_spmv = None
    if layout["spmv_variant"] in (20, 21) and world == 1:
        try:
            ms, by = soe.bench_kernel("spmv", layout["spmv_variant"], 20)
            plain_spmv = {"kernel": "k_spmv_c (y = A x), 20 launches behind the timed region", "avg_launch_ms": round(ms, 5),
                          "achieved": round(by / (ms * 1e-3) / 1e9, 1), "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        except Exception as e:
            plain_spmv = {"error": repr(e)}

    # rank-local SpMV of every rank (the roofline kernel): average launch time, gathered on the host
    # (round 5: where every line search took its sums from the Krylov solve's own kernels -- option ls_fold -- no product launch is left
    # in the timed region by itself: the roofline kernel is then the fused product + column solve, kernels.spmv_pc)
    folded = stats["spmv"]["launches"] == 0 and stats["spmv_pc"]["launches"] > 0
    sp = stats["spmv_pc"] if folded else stats["spmv"]
    avg_ms = sp["total_ms"] / max(sp["launches"], 1)
    all_ms = group.allreduce([avg_ms if r == rank else 0.0 for r in range(world)], op="sum")
    all_bytes = group.allreduce([sp["bytes_per_launch"] if r == rank else 0.0 for r in range(world)], op="sum")

    if rank == 0:
        n_global = ncols_global * args.nz
        n_local = ncols_local * args.nz
        achieved = sp["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        compact = n_local >= (1 << 21) and args.spmv_variant < 0
        # HBM bytes the kernels really move: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, kept in
        # profiles/kernel_traffic.json with the hash of the kernel sources they were made with -- quoted only for the same
        # sources at the same size, never replayed for other code
        traffic, traffic_src, moved, stale = None, None, {}, None
        tfile = os.path.join(ROOT, "profiles", "kernel_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("cells_per_gpu") != n_local:
                    stale = "measured at %s cells per GPU, this run has %d" % (tj.get("cells_per_gpu"), n_local)
                elif tj.get("kernel_source_sha16") != kernel_source_hash():
                    stale = "kernel sources changed since the PMC passes (%s then, %s now)" % (tj.get("kernel_source_sha16"), kernel_source_hash())
                else:
                    moved = tj.get("classes", {})
                    traffic = moved.get("spmv_pc" if folded else "spmv", {}).get("hbm_bytes_per_launch")
                    traffic_src = "profiles/kernel_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, %s; kernel sources %s)" % (
                        tj.get("run"), tj.get("kernel_source_sha16"))
            except Exception as e:
                stale = "unreadable: %r" % (e,)
        kernels = {}
        for k, s in stats.items():
            if s["launches"]:
                ms = s["total_ms"] / s["launches"]
                kernels[k] = {"launches_timed": s["launches"], "launches": s["calls"], "avg_ms": round(ms, 5),
                              "share_of_step_time": round(ms * s["calls"] * 1e-3 / elapsed, 4)}   # (calls and stride read back from the library)
                if s["bytes_per_launch"] > 0:
                    # SURVEY 8(d)'s algorithmic byte count / time.  The assembly passes move fewer bytes than SURVEY counts
                    # (soil classes, per-level geometry, computed row slots: DESIGN.md section 3), so this rate can exceed
                    # the HBM peak for them: structure_exploited says so, GBps_moved is the physical rate
                    alg = s["bytes_per_launch"] / (ms * 1e-3) / 1e9
                    kernels[k]["GBps_algorithmic"] = round(alg, 1)
                    mv = moved.get(k)
                    if mv:
                        kernels[k]["bytes_moved"] = round(mv["hbm_bytes_per_launch"])
                        kernels[k]["bytes_algorithmic"] = round(s["bytes_per_launch"])
                        kernels[k]["GBps_moved"] = round(mv["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
                        kernels[k]["moved_kernels"] = sorted(mv["kernels"])
                    if alg > HBM_PEAK_GBS or (mv and mv["hbm_bytes_per_launch"] < 0.9 * s["bytes_per_launch"]):
                        kernels[k]["structure_exploited"] = True
        fused = stats["spmv_pc"]["launches"] > 0
        index_free = bool(layout.get("line_search_product_index_free"))
        if folded:
            kernel_txt = ("csr_spmv_fp64 fused with the column solve, as the timed region runs it: t = M^-1 (A v) in one kernel per product of a "
                          "BiCGStab iteration (k_tri_solve_lds<.., SPMV>: the CSR VALUES of the column tridiagonals read at their computed places, "
                          "row sums in the CSR kernels' order, A v never written; DESIGN.md section 4), with the Krylov dot products and -- "
                          "round 5, ls_fold -- the line search's F.(J p), F.(J s) reduced on the way, so that no product launch of the search's "
                          "own is left in the timed region.  `achieved` / `frac`: that kernel pair's own byte model; plain_spmv_kernel = k_spmv_c "
                          "(values + compact 16-bit index, y written) timed right behind the region; the CSR kernel inside a timed region of "
                          "its own: extra_configs[vsfm3d].roofline")
        elif fused and index_free:
            kernel_txt = ("csr_spmv_fp64 as the timed region runs it: the line search's product J Y with x.x, f.(A x), ||x|| reduced on the way "
                          "and A x not written (the two products of a BiCGStab iteration run fused with the column solve: kernels.spmv_pc, "
                          "DESIGN.md section 4).  On this soil-column grid it is k_spmv_cols_dots: the CSR VALUES read at their computed places, "
                          "no column index and no row pointer; `achieved` / `frac` are computed from THAT kernel's byte model, `traffic` / "
                          "`achieved_physical` from its PMC bytes; plain_spmv_kernel = k_spmv_c (values + compact 16-bit index, y written), "
                          "the kernel of 3-D and multi-iteration solves, timed right behind the region")
        elif fused and compact:
            kernel_txt = ("csr_spmv_fp64 (k_spmv_c: CSR values + compact 16-bit index; the launches timed are the line search's pass "
                          "-- x.x, f.(A x), ||x|| reduced on the way, A x not written -- because the two products of a BiCGStab iteration "
                          "run fused with the column solve: kernels.spmv_pc, DESIGN.md section 4)")
        elif compact:
            kernel_txt = "csr_spmv_fp64 (k_spmv_c: CSR values + compact 16-bit index, see DESIGN.md section 4)"
        else:
            kernel_txt = "csr_spmv_fp64"
        # roofline.frac is a FRACTION OF PEAK of bytes the timed kernel can be said to move: where the timed launches are the
        # index-free product (no column index, no row pointer, A x not written) the byte model is that kernel's own --
        # values 8 nnz + x 8 N + f 8 N, plus X for ||X|| on the first launch of a time step (not counted) -- and SURVEY 8(d)'s CSR
        # count over the same time is reported beside it as achieved_algorithmic (it can exceed the peak: that kernel moves less)
        csr_bytes = float(stats["spmv"]["bytes_per_launch"])           # 12 nnz + 20 N + 4
        nnz_local = (csr_bytes - 20.0 * n_local - 4.0) / 12.0
        if folded:
            # the two kernels of an iteration: values 8 nnz each; first kernel d inv, g in, x out, then F in, g, x again, z out = 56 N;
            # second kernel r, v, F in, s out, g, s again, z out = 56 N  (g = up / pivot: the back substitution in product form,
            # DESIGN.md section 6; until round 5's first half c and 1 / pivot were read in their place: 72 N and 64 N)
            own_bytes = 8.0 * nnz_local + 56.0 * n_local
            csr_bytes = float(sp["bytes_per_launch"])                   # SURVEY's count of the SpMV + PCApply pair it replaces
        else:
            own_bytes = 8.0 * nnz_local + 16.0 * n_local if index_free else csr_bytes
        achieved_alg = achieved
        achieved = own_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roof = {"kernel": kernel_txt,
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(min(achieved / HBM_PEAK_GBS, 1.0), 4), "traffic": traffic,
                "achieved_is": ("the timed kernels' own byte model, averaged over the two fused kernels of a BiCGStab iteration: values 8 nnz + "
                                "56 N (first: d/pivot, g in, x out, F in, g, x again, z out; second: r, v, F in, s out, g, s "
                                "again, z out; g = upper entry / pivot) / HIP-event average launch time of those launches inside the timed region.  achieved_algorithmic = "
                                "SURVEY 8(d)'s count of the CSR SpMV + PCApply pair each launch replaces (12 nnz + 60 N + 4) over the same time")
                               if folded else
                               (("the timed kernel's own byte model (index-free launches: values 8 nnz + x 8 N + f 8 N; no column index, no row "
                                 "pointer, A x not written) / HIP-event average launch time of those launches inside the timed region.  "
                                 "achieved_algorithmic = SURVEY 8(d)'s CSR count (12 nnz + 20 N + 4) over the same time -- a rate the CSR kernel "
                                 "would have had to reach, not bytes this kernel moves.  The CSR kernel proper (values + index + y written) is "
                                 "timed inside its own region on the 7-point grid: extra_configs[...].roofline, and stand-alone here: plain_spmv_kernel")
                                if index_free else
                                ("algorithmic CSR bytes (12 nnz + 20 N + 4, SURVEY 8d) / HIP-event average launch time of the CSR SpMV launches "
                                 "inside the timed region.  Byte model of those launches: values 8 nnz + column index 4 nnz + row pointer 4 N + x 8 N, "
                                 "and f read (8 N) where y = A x would be written (8 N): the same count; the first launch of a time step also reads X "
                                 "for ||X|| (8 N more, not counted).  plain_spmv_kernel = the kernel that does write y, timed right behind the region"))
                               if fused else "algorithmic CSR bytes (12 nnz + 20 N + 4, SURVEY 8d) / HIP-event average launch time",
                "bytes_per_launch": own_bytes, "avg_launch_ms": round(avg_ms, 5), "launches_timed": sp["launches"]}
        if index_free or folded:
            roof["achieved_algorithmic"] = round(achieved_alg, 1)
            roof["bytes_per_launch_csr_count"] = csr_bytes
        if folded:
            roof["roofline_kernel_class"] = "spmv_pc"
            roof["line_search_sums_folded"] = layout.get("line_search_sums_folded")
        if plain_spmv:
            roof["plain_spmv_kernel"] = plain_spmv
        if index_free or folded:
            roof["structure_exploited"] = True
        if stale:
            roof["traffic_refused"] = stale
        if traffic:
            phys = traffic / (avg_ms * 1e-3) / 1e9
            roof["achieved_physical"] = round(phys, 1)        # bytes the kernel really moves (16-bit index) / the same time
            roof["frac_physical"] = round(phys / HBM_PEAK_GBS, 4)
            roof["traffic_source"] = traffic_src
        if world > 1:
            roof["per_rank"] = [{"rank": r, "avg_launch_ms": round(all_ms[r], 5),
                                 "achieved": round(all_bytes[r] * (own_bytes / csr_bytes) / (all_ms[r] * 1e-3) / 1e9, 1) if all_ms[r] > 0 else None}
                                for r in range(world)]
        out = {
            "metric": "Newton iters/sec (VSFM %dx%dx%d)" % (args.nx, args.ny, args.nz),
            "value": newton / elapsed, "unit": "Newton iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "VSFM celia-extruded %dx%d columns x %d layers, %s+%s, per-column perm multiplier"
                                   % (args.nx, args.ny, args.nz, args.ksp, args.pc),
                       "cells": n_global, "cells_per_gpu": n_local, "dt_s": prob.dt,
                       "soil_classes": layout["soil_classes"], "level_geometry": bool(layout["level_geometry"]),
                       "level_porosity_volume": bool(layout["level_porosity_volume"]),
                       "complete_row_march": bool(layout["complete_rows"]),
                       "parallelism": "columns sharded over %d GPU(s), RCCL all-reduce of Krylov/Newton scalars" % world},
            "newton_iterations": newton, "linear_iterations": linear, "setup_s": round(t_setup, 2),
            "time_step_cuts": c1["time_cuts"] - c0["time_cuts"],
            "accepted_substeps_s": substeps[args.warmup:],
            "allreduces_per_newton_iteration": round(comm["allreduces"] / max(newton, 1), 2) if world > 1 else 0.0,
            "host_readbacks_per_newton_iteration": round(comm["host_readbacks"] / max(newton, 1), 2),
            "speculative_trial_points_discarded": layout["speculative_trials_discarded"],
            "event_sampling": "every %d-th launch of a kernel class is timed with HIP events on the solver's stream" % EVENT_STRIDE,
            "roofline": roof,
            "kernels": kernels,
        }
    # ---- outside the timed region: the benchmark's own answer against the oracle (every rank checks its own shard) ----
    check = None
    if not args.no_verify:
        try:
            pg = soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
            check = verify_answer(pg, ncols_global, args.nz, col0, ncols_local, substeps, not args.homogeneous,
                                  nsample=64 if world == 1 else max(8, 64 // world))
            del pg
        except Exception as e:   # the check must never hide the measurement; its failure is reported in the line
            check = {"error": repr(e)}
    if not args.no_verify:
        worst = group.allreduce([check.get("max_rel_err", float("inf")) if check else float("inf")], op="max")[0]
        ncheck = group.allreduce([float(check.get("columns_checked", 0)) if check else 0.0], op="sum")[0]
        if rank == 0:
            out["answer_check"] = dict(check or {}, max_rel_err=worst, columns_checked=int(ncheck))
    if rank == 0 and world == 1 and not args.no_extra_configs:
        # outside the timed region: the ELM-default constitutive set-up at this size and BASELINE config 2 (256 x 256 x 50)
        extras = []
        for label, dims, elm in (("ELM default: per-cell soils + smooth_brooks_corey_bz3, %dx%dx%d" % (args.nx, args.ny, args.nz),
                                  (args.nx, args.ny, args.nz), True),
                                 ("BASELINE config 2: 256x256x50, %s+%s" % (args.ksp, args.pc), (256, 256, 50), False)):
            try:
                sys.stderr.write("bench.py: extra configuration: %s ...\n" % label); sys.stderr.flush()
                big = dims[0] * dims[1] * dims[2] > (1 << 24)
                extras.append(extra_config(label, dims[0], dims[1], dims[2], elm, 3 if big else 8, 1 if big else 2, opts, pick_device(args, local_rank)))
            except Exception as e:   # never hides the headline
                extras.append({"config": label, "error": repr(e)})
        try:
            sys.stderr.write("bench.py: extra configuration: 7-point VSFM grid 512x512x100 ...\n"); sys.stderr.flush()
            extras.append(extra_vsfm3d(512, 512, 100, 2, 1, args.ksp, args.pc, pick_device(args, local_rank)))
        except Exception as e:
            extras.append({"config": "vsfm3d", "error": repr(e)})
        out["extra_configs"] = extras
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported at N = 1 only
            # every core this process may use (affinity and cgroup quota), at most 64 shards of 8192 columns
            cores = args.cpu_cores or min(usable_cores(), 64)
            n_full = ncols_global * args.nz
            cpc = max(64, min(819200 // args.nz, ncols_global // cores))   # 8192 columns x 100 layers per thread
            try:
                # ~10-15 s of CPU work per leg: cores x 8192 columns x 100 layers, 3 timed steps after 1 warm-up
                sys.stderr.write("bench.py: CPU baseline on %d threads ...\n" % cores); sys.stderr.flush()
                ref = cpu_baseline(ncols_global, args.nz, 3, 1, cores, cols_per_core=cpc, ksp="gmres", pc="ilu")
                sys.stderr.write("bench.py: CPU baseline, same solver as the GPU ...\n"); sys.stderr.flush()
                one = cpu_one_core(ref.pop("_shards"), n_full)
                ref.pop("_work0")
                same = cpu_baseline(ncols_global, args.nz, 3, 1, cores, cols_per_core=cpc, ksp=args.ksp,
                                    pc="bjacobi" if args.pc in ("bjacobi", "ilu") and args.ksp == "bcgs" else "ilu")
                same.pop("_shards"); same.pop("_work0")
                out["cpu_baseline"] = {"value": ref["value"], "unit": "Newton iters/s", "cores": cores, "kind": "port",
                                       "sample": ref["sample"] + "; GMRES(30)+ILU(0) = the reference's PETSc defaults",
                                       "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "usable_cores": usable_cores(),
                                       "value_one_core": one, "seconds": ref["seconds"],
                                       "extrapolated_all_physical_cores": extrapolate_cpu(ref["value"], cores, one, out["value"]),
                                       "same_solver_as_gpu": {"value": same["value"], "solver": same["solver"],
                                                              "seconds": same["seconds"]}}
            except Exception as e:  # the baseline must never hide the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "Newton iters/s", "cores": cores, "kind": "port",
                                       "cpu_model": cpu_model(), "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    group.barrier()
    group.close()


if __name__ == "__main__":
    if os.environ.get("MPPGPU_BENCH_RANK_PROCESS") or ("RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1):
        # a rank of a multi-GPU run: an error ends THIS process at once, with its message on stderr and a non-zero code -- no
        # interpreter teardown (a communicator with an unfinished collective would be waited for), never a re-exec; the
        # launcher (or torch.distributed.run) then stops the other ranks
        try:
            main()
        except SystemExit:
            raise
        except BaseException:
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            sys.stdout.flush()
            os._exit(1)
    else:
        main()
