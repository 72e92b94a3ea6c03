"""More GPU tests: the Fortran iso_c_binding host, the RCCL path (one-rank communicator), and known answers at
BASELINE.json's full sizes through size-independent properties."""
import os
import subprocess

import numpy as np
import pytest

import problems as P
from conftest import parse_regression_file
from mpp_amd import constants as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fortran_host_reproduces_celia1990_baseline(golden_dir, tmp_path):
    """fortran/vsfm_celia1990_gpu.F90: the reference driver's call sequence through the iso_c_binding shim."""
    exe = os.path.join(ROOT, "fortran", "_build", "vsfm_celia1990_gpu")
    if not os.path.exists(exe):
        pytest.skip("no Fortran compiler was available at build time")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if run.returncode != 0 and "error while loading shared libraries" in run.stderr:
        pytest.skip("Fortran runtime libraries are not installed on this box: " + run.stderr.strip())
    assert run.returncode == 0, run.stdout + run.stderr
    out = tmp_path / "celia.regression"
    out.write_text(run.stdout)
    got = parse_regression_file(str(out))
    gold = parse_regression_file(os.path.join(golden_dir, "vsfm_celia1990.regression.baseline"))
    for sec in ("liquid_pressure", "liquid_saturation"):
        for k, v in gold[sec].items():
            if k != "category":
                assert got[sec][k] == pytest.approx(v, rel=1e-9), (sec, k)


@pytest.mark.parametrize("sat", [0, 1])
def test_fortran_hosted_vsfm_mms_matches_the_oracle(sat):
    """fortran/vsfm_mms_gpu.F90: the manufactured-solution check driven from Fortran -- per-cell-array setters, a mass-rate
    source on every cell computed with mppgpu_host_density / mppgpu_host_sat_relperm, PreStepDT, one StepDT(1.0)
    (src/driver/standalone/vsfm/vsfm_mms_problem.F90:97-102); the printed pressures against the oracle's solve of the same
    problem, the printed max-norm error against the analytic P(x), at two resolutions"""
    from oracle import adapters as A
    exe = os.path.join(ROOT, "fortran", "_build", "vsfm_mms_gpu")
    if not os.path.exists(exe):
        pytest.skip("no Fortran compiler was available at build time")
    errs = []
    for nx in (20, 40):
        run = subprocess.run([exe, str(nx), str(sat)], capture_output=True, text=True, timeout=300)
        if run.returncode != 0 and "error while loading shared libraries" in run.stderr:
            pytest.skip("Fortran runtime libraries are not installed on this box: " + run.stderr.strip())
        assert run.returncode == 0, run.stdout + run.stderr
        o = A.to_oracle(A.vsfm_mms(nx, bool(sat)), with_blocks=False)
        o.pre_step_dt()
        assert o.step_dt(1.0, 1)[0]
        po = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
        got = {}
        for line in run.stdout.splitlines():
            if "=" in line and not line.startswith("category"):
                k, v = line.split("=")
                got[k.strip()] = float(v)
        assert got["min"] == pytest.approx(po.min(), rel=1e-8) and got["max"] == pytest.approx(po.max(), rel=1e-8)
        assert got["mean"] == pytest.approx(po.sum() / po.size, rel=1e-8)
        for c in range(1, nx + 1, nx // 5):
            assert got["cell %6d" % c] == pytest.approx(po[c - 1], rel=1e-8), c
        exact = P.vsfm_mms(nx, bool(sat)).exact
        assert got["max_error"] == pytest.approx(np.abs(po - exact).max(), rel=1e-4)
        errs.append(got["max_error"])
    assert errs[0] < (40.0 if sat else 1000.0)
    assert (3.9 if sat else 1.9) < errs[0] / errs[1] < (4.1 if sat else 2.1), errs


def test_rccl_single_rank_communicator_gives_identical_answers():
    """the all-reduce path with a one-rank RCCL communicator must not change a single bit"""
    from mpp_amd import vsfm as V
    prob = P.celia1990(ncols=70, nz=30, perm_mult=np.exp(0.5 * np.random.default_rng(1).standard_normal(70)))
    a = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    b = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    b.soe.comm_init(1, 0, V.comm_get_unique_id())
    for s in range(3):
        ra = a.soe.StepDT(prob.dt, s + 1)
        rb = b.soe.StepDT(prob.dt, s + 1)
        assert ra == rb
    pa = a.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    pb = b.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    assert np.array_equal(pa, pb)
    assert a.soe.counters() == b.soe.counters()


def test_config2_256x256x50_known_answer_and_operator_properties():
    """BASELINE config 2 (256x256 columns x 50 layers, BiCGStab + block-Jacobi): every column of the homogeneous
    celia-extruded grid must equal the single-column oracle solve; SpMV is linear; the column block-Jacobi
    preconditioner is an exact inverse for vertical-only connectivity (M^-1 A v = v)."""
    ncols, nz = 256 * 256, 50
    prob = P.celia1990(ncols=ncols, nz=nz)
    g = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    one = P.celia1990(ncols=1, nz=nz)
    o = P.to_oracle(one)
    o.set_solver(C.KSP_GMRES, C.PC_ILU0)
    for s in range(3):
        conv, reason = g.soe.StepDT(prob.dt, s + 1)
        assert conv and reason > 0
        o.step_dt(one.dt, s + 1)
    pr = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1).reshape(ncols, nz)
    ref = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.abs(pr - ref[None, :]).max() <= 1e-9 * np.abs(ref).max()
    n = ncols * nz
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    g.soe.ComputeJacobian(prob.press_ic + rng.uniform(0, 9e4, n))
    Au, Av, Auv = g.soe.MatMult(u), g.soe.MatMult(v), g.soe.MatMult(2.0 * u - 3.0 * v)
    scale = np.abs(Au).max() + np.abs(Av).max()
    assert np.abs(Auv - (2.0 * Au - 3.0 * Av)).max() <= 1e-13 * scale
    z = g.soe.PCApply(Av)
    assert np.abs(z - v).max() <= 1e-8 * np.abs(v).max()


def test_config3_1024x1024x100_every_column_reproduces_the_reference_baseline(golden_dir):
    """BASELINE config 3 at full size on one GPU (104,857,600 cells): 24 x 3600 s on the homogeneous celia-extruded
    grid; min and max over all 1,048,576 columns of every reported statistic must equal the reference's
    vsfm_celia1990 golden output (a checksum of per-column results).  Tolerance 1e-8 relative: the global norms
    are sums over 1M columns in a different order than the single-column run, which moves the Newton stopping
    decisions by rounding; cell 41 sits on the wetting front and is the most sensitive value."""
    gold = parse_regression_file(os.path.join(golden_dir, "vsfm_celia1990.regression.baseline"))["liquid_pressure"]
    ncols, nz = 1024 * 1024, 100
    prob = P.celia1990(ncols=ncols, nz=nz)
    g = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    for s in range(prob.nstep):
        conv, reason = g.soe.StepDT(prob.dt, s + 1)
        assert conv and reason > 0
    pr = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1).reshape(ncols, nz)
    for stat, val in (("min", pr.min(axis=1)), ("max", pr.max(axis=1)), ("mean", pr.sum(axis=1) / nz)):
        assert val.min() == pytest.approx(gold[stat], rel=1e-8) and val.max() == pytest.approx(gold[stat], rel=1e-8), stat
    for cell in (1, 21, 41, 61, 81):
        colv = pr[:, cell - 1]
        assert colv.min() == pytest.approx(gold["cell %d" % cell], rel=1e-8)
        assert colv.max() == pytest.approx(gold["cell %d" % cell], rel=1e-8)
        assert colv.max() - colv.min() <= 1e-12 * abs(colv.max())   # all columns agree with each other


@pytest.mark.parametrize("extra", [{}, {"assembly": "columns", "nseg": 1, "spmv_variant": 20},
                                   {"assembly": "columns", "nseg": 1, "tri_lds": 1}, {"assembly": "columns", "nseg": 1, "tri_lds": 1, "ls_fold": 0}],
                         ids=["default", "fused_rows", "fused_lds_solves_folded_line_search", "fused_lds_solves"])
def test_sharded_global_solve_on_one_gpu_equals_the_unsharded_solve(extra):
    """SURVEY 8(e) on real kernels: the column set split over 3 handles (ranks) of this process, one host thread each,
    forming ONE global Newton-Krylov solve through the in-process communicator (same call sites as the RCCL
    all-reduce) -- every rank must take the same Newton / Krylov decisions and the assembled result must equal the
    single-handle solve.  Heterogeneous columns, ragged shard sizes."""
    import threading
    from mpp_amd import parallel as PAR
    from mpp_amd import vsfm as V
    ncols, nz, nranks = 203, 30, 3
    mult = np.exp(0.5 * np.random.default_rng(11).standard_normal(ncols))
    # fused_rows: every fused pass under a communicator; fused_lds_*: the kernels a rank of the 8-GPU benchmark runs (M^-1 A v as one
    # kernel with the column solve in LDS, and -- round 5 -- the line search's sums all-reduced out of those kernels: ls_fold)
    opts = dict({"ksp_type": "bcgs", "pc_type": "bjacobi"}, **extra)
    ref_prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=mult)
    ref = P.to_gpu(ref_prob, opts)
    shards = []
    for r in range(nranks):
        c0, nloc = PAR.shard_columns(ncols, nranks, r)
        prob = P.celia1990(ncols=nloc, nz=nz, perm_mult=mult[c0:c0 + nloc])
        shards.append((c0, nloc, prob, P.to_gpu(prob, opts)))
    V.comm_init_local([s[3].soe for s in shards])
    nsteps = 3
    results = [None] * nranks
    errors = []

    def run(r):
        try:
            c0, nloc, prob, g = shards[r]
            out = []
            for s in range(nsteps):
                out.append(g.soe.StepDT(prob.dt, s + 1))
            results[r] = out
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    ref_out = [ref.soe.StepDT(ref_prob.dt, s + 1) for s in range(nsteps)]
    for r in range(nranks):
        assert results[r] == ref_out                      # same (converged, reason) at every step on every rank
        assert shards[r][3].soe.counters()["newton"] == ref.soe.counters()["newton"]
        assert shards[r][3].soe.counters()["linear"] == ref.soe.counters()["linear"]
    p_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE).reshape(ncols, nz)
    for c0, nloc, prob, g in shards:
        p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE).reshape(nloc, nz)
        assert np.abs(p - p_ref[c0:c0 + nloc]).max() <= 1e-9 * np.abs(p_ref).max()
    folded = [s[3].soe.assembly_info()["line_search_sums_folded"] for s in shards]
    if extra.get("tri_lds") and extra.get("ls_fold", 1):
        assert min(folded) > 0 and len(set(folded)) == 1      # every rank folded the same line searches
    else:
        assert max(folded) == 0


def _run_ranks(fns):
    import threading
    errors, out = [], [None] * len(fns)

    def run(r):
        try:
            out[r] = fns[r]()
        except Exception as e:   # noqa: BLE001
            errors.append((r, e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in th)
    return out


@pytest.mark.parametrize("nranks", [2, 3])
def test_3d_mesh_split_across_ranks_thermal(nranks):
    """the thermal MMS grid split into z-slabs over `nranks` handles with ghost planes and a halo
    exchange before every operator assembly and SpMV, GMRES + rank-local ILU(0) (PETSc's parallel default).  The
    operator rows of the owned cells equal the single-handle rows; the solution equals the single-handle one to the
    linear tolerance."""
    from mpp_amd import vsfm as V
    tp = P.thermal_mms(3, 7, 6, 9, replicate_reference_soil_indexing=False)
    opts = {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-12}
    ref = P.to_thermal_gpu(tp, opts)
    rp, cj, val = ref.soe.ComputeOperators(tp.dt)
    b_ref = ref.soe.ComputeRHS(tp.dt)
    ref.soe.PreStepDT()
    conv, _ = ref.soe.StepDT(tp.dt)
    assert conv
    t_ref = ref.soe.GetSoln()
    parts = P.split_thermal_problem(tp, nranks)
    gs = [P.to_thermal_gpu(lp, opts) for _, lp in parts]
    V.comm_init_local([g.soe for g in gs])
    ops = _run_ranks([lambda g=g: (g.soe.ComputeOperators(tp.dt), g.soe.ComputeRHS(tp.dt)) for g in gs])
    n = tp.mesh.ncells_local
    A = np.zeros((n, n))
    for i in range(n):
        A[i, cj[rp[i]:rp[i + 1]]] = val[rp[i]:rp[i + 1]]
    for (part, lp), ((lrp, lcj, lval), lb) in zip(parts, ops):
        for li in range(part.nowned):           # owned rows, columns mapped back to global ids
            row = np.zeros(n)
            row[part.l2g[lcj[lrp[li]:lrp[li + 1]]]] = lval[lrp[li]:lrp[li + 1]]
            assert np.abs(row - A[part.l2g[li]]).max() <= 1e-13 * np.abs(A[part.l2g[li]]).max()
        assert np.abs(lb[:part.nowned] - b_ref[part.l2g[:part.nowned]]).max() <= 1e-12 * np.abs(b_ref).max()
        for li in range(part.nowned, lp.mesh.ncells_local):   # ghost rows are identity rows
            assert lval[lrp[li]:lrp[li + 1]].sum() == 1.0 and np.count_nonzero(lval[lrp[li]:lrp[li + 1]]) == 1

    def step(g):
        g.soe.PreStepDT()
        return g.soe.StepDT(tp.dt)

    res = _run_ranks([lambda g=g: step(g) for g in gs])
    assert all(r[0] for r in res)
    for (part, lp), g in zip(parts, gs):
        t = g.soe.GetSoln()
        assert np.abs(t[:part.nowned] - t_ref[part.l2g[:part.nowned]]).max() <= 1e-8


def test_3d_mesh_split_across_ranks_vsfm():
    """the 7-point v-channel problem on 3 ranks that own whole soil columns (x-y blocks; ghost columns, halo exchange before every
    residual / Jacobian / SpMV, GMRES(30) + rank-local ILU(0) = the reference's defaults) against the single-handle solve: residual
    rows of the owned cells at a random state, then three time steps with the single handle's Newton count and its linear
    iterations +- 2 per solve"""
    from mpp_amd import vsfm as V
    prob = P.vchannel(6, 5, 9)
    n = prob.mesh.ncells_local
    opts = {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-10}
    ref = P.to_gpu(prob, opts)
    x = prob.press_ic - np.random.default_rng(5).uniform(0, 2.5e5, n)
    ref.soe.SetDtime(prob.dt); ref.soe.PreSolve()
    f_ref = ref.soe.ComputeResidual(x)
    parts = P.split_problem(prob, 3)
    assert all(part.mode == "columns" and part.ncols_owned == 10 for part, _ in parts)
    assert [part.recv_count.tolist() for part, _ in parts] == [[6 * 9], [6 * 9, 6 * 9], [6 * 9]]   # nz x edge cells per neighbour
    gs = [P.to_gpu(lp, opts) for _, lp in parts]
    V.comm_init_local([g.soe for g in gs])
    # the halo of a 7-point nx x ny x nz grid cut in y is nz x nx cells = nz * nx * 8 bytes per neighbour and exchange
    nzx = 9 * 6
    assert gs[0].soe.halo_stats() == dict(neighbours=1, cells_sent=nzx, cells_received=nzx, bytes_received_per_exchange=8 * nzx,
                                          max_cells_from_one_neighbour=nzx)
    assert gs[1].soe.halo_stats() == dict(neighbours=2, cells_sent=2 * nzx, cells_received=2 * nzx, bytes_received_per_exchange=16 * nzx,
                                          max_cells_from_one_neighbour=nzx)
    assert ref.soe.halo_stats()["bytes_received_per_exchange"] == 0

    def resid(part, g):
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        return g.soe.ComputeResidual(x[part.l2g])

    fs = _run_ranks([lambda part=part, g=g: resid(part, g) for (part, _), g in zip(parts, gs)])
    for (part, lp), f in zip(parts, fs):
        own = part.l2g[:part.nowned]
        assert np.abs(f[:part.nowned] - f_ref[own]).max() <= 1e-11 * np.abs(f_ref).max()
        assert (f[part.nowned:] == 0.0).all()
    for s in range(prob.nstep):
        assert ref.soe.StepDT(prob.dt, s + 1)[0]
    p_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)

    def steps(g):
        return [g.soe.StepDT(prob.dt, s + 1) for s in range(prob.nstep)]

    res = _run_ranks([lambda g=g: steps(g) for g in gs])
    assert all(c for r in res for (c, _) in r)
    c_ref = ref.soe.counters()
    for (part, lp), g in zip(parts, gs):
        c = g.soe.counters()
        assert c["newton"] == c_ref["newton"] and abs(c["linear"] - c_ref["linear"]) <= 2 * c_ref["newton"], (c, c_ref)
        p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
        assert np.abs(p[:part.nowned] - p_ref[part.l2g[:part.nowned]]).max() <= 1e-7 * np.abs(p_ref).max()


def test_two_vertical_connections_between_the_same_pair_of_cells_are_not_taken_for_a_column():
    """a mesh whose internal connection set joins one pair of cells TWICE (each with half the face area) is not a set of
    soil columns -- the column layout stores one link per cell and would drop one of the two.  Column detection must see it and the
    general assembly (which supports repeated pairs: both fluxes land in the same matrix entries) must run; operators against the oracle"""
    from mpp_amd import mesh as M
    from test_gpu_parity import _hetero, _operator_parity
    prob, x = _hetero(ncols=5, nz=8, seed=13)
    cs = prob.mesh.intrn_conn_sets[0]
    k = 10                                           # the connection that is split in two
    area = np.asarray(cs.area, dtype=float).copy()
    area[k] *= 0.5
    dup = lambda a: np.concatenate([np.asarray(a), np.asarray(a)[k:k + 1]])
    prob.mesh.intrn_conn_sets = [M.ConnectionSet(dup(cs.id_up), dup(cs.id_dn), dup(cs.dist_up), dup(cs.dist_dn), dup(area),
                                                 dup(cs.unit_vec), None if cs.itype is None else dup(cs.itype))]
    o, g, _, _ = _operator_parity(prob, x, {})
    info = g.soe.assembly_info()
    assert info["column_segments"] == 0 and not info["complete_rows"]      # not the column march
    _operator_parity(prob, x, {"assembly": "general"})


def test_thermal_workload_on_four_in_process_ranks_equals_the_single_handle_solve():
    """`bench.py --workload thermal --gpus 4` (BASELINE config 4's rank count) on one GPU: every rank builds its own z-slab
    (workloads.thermal_mms_rank: no global mesh), the four handles form one global GMRES(30) + rank-local ILU(0) solve through the
    in-process group (halo exchange before the operator assembly and every SpMV, all-reduced dots); at ksp_rtol 1e-11 the owned
    cells reproduce the single-handle solution, and every rank took the same number of iterations"""
    from mpp_amd import vsfm as V
    n, world = 16, 4
    opts = {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-11}
    tp = P.thermal_mms(3, n, n, n, replicate_reference_soil_indexing=False)
    ref = P.to_thermal_gpu(tp, opts)
    ref.soe.PreStepDT()
    assert ref.soe.StepDT(tp.dt)[0]
    t_ref = ref.soe.GetSoln()
    ranks = [P.thermal_mms_rank(n, n, n, world, r) for r in range(world)]
    gs = [P.to_thermal_gpu(lp, opts) for _, lp in ranks]
    V.comm_init_local([g.soe for g in gs])

    def step(g):
        g.soe.PreStepDT()
        conv, _ = g.soe.StepDT(tp.dt)
        return conv, g.soe.last_linear_iterations

    res = _run_ranks([lambda g=g: step(g) for g in gs])
    assert all(c for c, _ in res) and len({its for _, its in res}) == 1
    # Two different preconditioners (ILU(0) of the whole matrix / of the four slabs), each stopped at 1e-11 of ITS OWN preconditioned
    # residual norm: the two answers need not agree to 1e-11 of anything.  What both must do is solve A t = b -- the TRUE residual
    # ||b - A t|| / ||b|| of each answer, formed on the host from the single handle's operator (a measured 1.07e-8 K
    # difference had been waved through with a looser bound; this is the statement that holds)
    import scipy.sparse as sp
    ref.soe.PreStepDT()        # (the right-hand side of the step that was solved: from the state before it)
    rp, cj, val = ref.soe.ComputeOperators(tp.dt)
    b = ref.soe.ComputeRHS(tp.dt)
    A = sp.csr_matrix((val, cj, rp), shape=(b.size, b.size))
    t_split = np.empty_like(t_ref)
    for (part, lp), g in zip(ranks, gs):
        t = g.soe.GetSoln()
        t_split[part.l2g[:part.nowned]] = t[:part.nowned]
        assert np.abs(t[:part.nowned] - lp.exact[:part.nowned]).max() < 0.5      # the manufactured solution to the discretisation error
    rr_ref = np.linalg.norm(b - A @ t_ref) / np.linalg.norm(b)
    rr_split = np.linalg.norm(b - A @ t_split) / np.linalg.norm(b)
    print("true relative residuals: single handle %.3e, four ranks %.3e; max |difference| %.3e K" % (rr_ref, rr_split, np.abs(t_split - t_ref).max()))
    assert rr_ref <= 1e-9 and rr_split <= 1e-9, (rr_ref, rr_split)
    # ... and then they differ by no more than the residuals allow: |dt| <= ||A^-1|| (|r_ref| + |r_split|); measured 1.1e-8 K on 270 K
    assert np.abs(t_split - t_ref).max() <= 1e-6
    assert gs[0].soe.pc_info()["path"] == "boxes"                                # whole z-slabs are structured grids: solved by boxes


@pytest.mark.parametrize("restart", [45, 70])
def test_gmres_restart_lengths_above_31_follow_the_oracle(restart):
    """-ksp_gmres_restart beyond what one Gram-Schmidt launch takes (32 vectors): the basis is walked in chunks -- all dot products
    of the unmodified vector first, then the updates in ascending order -- so the iterates are those of the oracle's GMRES(m) with
    classical Gram-Schmidt: same iteration counts over two steps of the 3-D v-channel problem with a preconditioner weak enough
    (the column blocks only) that a solve takes 48-63 iterations -- past 32 in both cases, past the restart length with 45 -- and
    the same pressures"""
    prob = P.vchannel(nx=9, ny=7, nz=8)
    g = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "bjacobi", "ksp_gmres_restart": restart, "ksp_rtol": 1e-10})
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_GMRES, C.PC_BJACOBI, ksp_rtol=1e-10)
    o.set_gmres_restart(restart)
    for s in range(2):
        assert g.soe.StepDT(prob.dt, s + 1) == tuple(o.step_dt(prob.dt, s + 1))
    cg, co = g.soe.counters(), o.counters()
    assert cg["newton"] == co["newton"] and cg["linear"] == co["linear"]
    assert cg["linear"] > 40 * cg["newton"]      # solves of more than 32 iterations were really run (oracle: 254 and 191 over 4)
    pg, po = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1), o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.abs(pg - po).max() <= 1e-8 * np.abs(po).max()


@pytest.mark.parametrize("dims,what", [((20, 10, 30), "plain"), ((6, 5, 7), "hetero"), ((70, 3, 9), "mixed_laws"), ((5, 4, 2), "two_levels"),
                                       ((9, 8, 6), "inactive_sources"), ((7, 6, 5), "lateral_first")])
def test_3d_mesh_of_complete_columns_is_assembled_by_the_column_march_plus_a_lateral_pass(dims, what):
    """Round 5: a 3-D mesh whose cells form complete soil columns no longer drops to the general auxvar pass + row
    gather because it has lateral connections -- the column march (k_assemble_cols) does the accumulation term and the chain links and
    writes the column tridiagonal at the CSR slots of the 7-point rows, a lateral pass (k_flux_rows_ell<.., LAT>) adds the lateral
    connections row by row.  Operators against the oracle (pattern equal, values / residual to rounding), against the general route
    of the same library, and two time steps; `lateral_first`: the connection list reordered so that the lateral connections precede
    the vertical ones (MeshCreate's x, y, z order) -- the per-row sums then run in another order than the reference's: rounding only"""
    from mpp_amd import mesh as M
    from test_gpu_parity import _operator_parity
    prob = P.vchannel(*dims)
    n = prob.mesh.ncells_local
    rng = np.random.default_rng(77)
    x = prob.press_ic - rng.uniform(0, 2.5e5, n)
    if what in ("hetero", "mixed_laws", "inactive_sources", "lateral_first"):
        prob.perm = prob.perm * np.exp(0.4 * rng.standard_normal(n))[:, None] * rng.uniform(0.5, 2.0, 3)[None, :]
        prob.por = prob.por * rng.uniform(0.8, 1.2, n)
        prob.alpha = prob.alpha * rng.uniform(0.8, 1.2, n)
        prob.sat_res = prob.sat_res * rng.uniform(0.8, 1.2, n)
    if what == "mixed_laws":
        satf = np.asarray([C.SAT_FUNC_VAN_GENUCHTEN, C.SAT_FUNC_BROOKS_COREY, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2,
                           C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3], dtype=np.int32)
        prob.satfunc_type = satf[rng.integers(0, 4, n)]
        prob.lam = np.where(prob.satfunc_type == C.SAT_FUNC_VAN_GENUCHTEN, rng.uniform(0.3, 0.7, n), rng.uniform(0.4, 1.6, n))
        prob.por_pressure_reference = np.full(n, 101325.0) + rng.uniform(-2e4, 2e4, n)
        prob.por_slope = rng.uniform(1e-8, 6e-8, n)
    if what == "inactive_sources":
        act = np.ones(n, dtype=np.int32)
        act[rng.choice(n, n // 30, replace=False)] = 0
        prob.mesh.is_active = act
        z = np.zeros(n)
        prob.sss = [(C.COND_MASS_RATE, M.ConnectionSet(np.full(n, -1), np.arange(1, n + 1), z, z, np.ones(n), np.zeros((n, 3))),
                     rng.uniform(-2e-4, 2e-4, n))]
    if what == "lateral_first":
        cs = prob.mesh.intrn_conn_sets[0]
        nv = dims[0] * dims[1] * (dims[2] - 1)
        order = np.concatenate([np.arange(nv, cs.num_connections), np.arange(nv)])
        prob.mesh.intrn_conn_sets = [M.ConnectionSet(cs.id_up[order], cs.id_dn[order], cs.dist_up[order], cs.dist_dn[order], cs.area[order],
                                                     cs.unit_vec[order], None)]
    o, g, (rp, cj, val), v = _operator_parity(prob, x, {})
    assert g.soe.assembly_info()["columns_plus_lateral_assembly"] == 1
    h = P.to_gpu(prob, {"cols3d": 0})
    assert h.soe.assembly_info()["columns_plus_lateral_assembly"] == 0
    h.soe.SetDtime(prob.dt); h.soe.PreSolve()
    f_g, f_h = g.soe.ComputeResidual(x), h.soe.ComputeResidual(x)
    (rp_g, cj_g, val_g), (rp_h, cj_h, val_h) = g.soe.ComputeJacobian(x), h.soe.ComputeJacobian(x)
    assert np.array_equal(rp_g, rp_h) and np.array_equal(cj_g, cj_h)
    assert np.abs(val_g - val_h).max() <= 1e-12 * np.abs(val_h).max()
    rowscale = np.zeros(n)
    np.maximum.at(rowscale, np.repeat(np.arange(n), np.diff(rp_h)), np.abs(val_h))
    assert (np.abs(f_g - f_h) <= 1e-11 * (np.abs(f_h) + rowscale * 250.0) + 1e-25).all()
    # (not to the bit even where the caller lists the vertical connections first: the march streams porosity x volume and
    # conductance x face area as one operand each, the general route multiplies the factors one by one)
    # residual + Jacobian in one pass (the Newton loop's route) against the two separate passes
    g2 = P.to_gpu(prob, {"fused_assembly": 0})
    g2.soe.SetDtime(prob.dt); g2.soe.PreSolve()
    res = [q.soe.StepDT(prob.dt, 1) for q in (g, h, g2)]
    assert res[0] == res[1] == res[2] and res[0][0]
    p = [q.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1) for q in (g, h, g2)]
    assert np.abs(p[0] - p[1]).max() <= 1e-10 * np.abs(p[1]).max() and np.abs(p[0] - p[2]).max() <= 1e-10 * np.abs(p[1]).max()
    cg, ch = g.soe.counters(), h.soe.counters()
    assert (cg["newton"], cg["time_cuts"]) == (ch["newton"], ch["time_cuts"]) and abs(cg["linear"] - ch["linear"]) <= 2 * ch["newton"]


@pytest.mark.parametrize("nranks", [2, 3])
def test_column_block_jacobi_is_the_same_operator_on_a_split_3d_mesh(nranks):
    """BASELINE's VSFM solver, BiCGStab + per-column block-Jacobi, on the 7-point v-channel split over ranks that own whole soil columns:
    a rank's cells keep the caller's order (ghost cells last), so the column blocks are found as linked chains of owned cells
    (ColChainArgs, k_chain_tri_factor / k_chain_tri_solve) -- the SAME preconditioner as on one handle, whatever the rank count:
    M^-1 r of the owned cells equals the single handle's to rounding, and two time steps take the single handle's Newton iterations and
    its linear iterations (+- 1 per solve: the all-reduced dot products are summed in another order)"""
    from mpp_amd import vsfm as V
    prob = P.vchannel(6, 6, 8)
    n = prob.mesh.ncells_local
    rng = np.random.default_rng(17)
    prob.perm = prob.perm * np.exp(0.3 * rng.standard_normal(n))[:, None]
    opts = {"ksp_type": "bcgs", "pc_type": "bjacobi", "ksp_rtol": 1e-9}
    ref = P.to_gpu(prob, opts)
    x = prob.press_ic - rng.uniform(0, 2.0e5, n)
    r = rng.standard_normal(n)
    ref.soe.SetDtime(prob.dt); ref.soe.PreSolve()
    ref.soe.ComputeJacobian(x)
    z_ref = ref.soe.PCApply(r)
    parts = P.split_problem(prob, nranks)
    gs = [P.to_gpu(lp, opts) for _, lp in parts]
    V.comm_init_local([g.soe for g in gs])

    def apply(part, g):
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        g.soe.ComputeJacobian(x[part.l2g])
        return g.soe.PCApply(r[part.l2g])

    zs = _run_ranks([lambda part=part, g=g: apply(part, g) for (part, _), g in zip(parts, gs)])
    for (part, lp), z in zip(parts, zs):
        own = part.l2g[:part.nowned]
        assert np.abs(z[:part.nowned] - z_ref[own]).max() <= 1e-12 * np.abs(z_ref).max()
    res_ref = [ref.soe.StepDT(prob.dt, s + 1) for s in range(2)]
    res = _run_ranks([lambda g=g: [g.soe.StepDT(prob.dt, s + 1) for s in range(2)] for g in gs])
    assert all(rr == res_ref for rr in res) and all(c for c, _ in res_ref)
    c_ref = ref.soe.counters()
    assert c_ref["linear"] > 3 * c_ref["newton"]            # (the lateral couplings make BiCGStab iterate: not a one-iteration solve)
    for g in gs:
        c = g.soe.counters()
        assert c["newton"] == c_ref["newton"] and abs(c["linear"] - c_ref["linear"]) <= c_ref["newton"], (c, c_ref)
    p_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    for (part, lp), g in zip(parts, gs):
        p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
        assert np.abs(p[:part.nowned] - p_ref[part.l2g[:part.nowned]]).max() <= 1e-7 * np.abs(p_ref).max()
