"""GPU tests of the round-2 kernels and failure paths, through the C ABI:
  * the column solve that parks L^{-1} r in LDS (k_tri_solve_lds) against dense algebra, the oracle and the through-memory
    solve, on ragged tiles and column heights from 2 to 120 levels;
  * a chained ILU(0) application whose dependency waits time out is redone with per-level launches (same answer);
  * a failure on one rank of an in-process group ends every rank with an error instead of blocking the others;
  * the benchmark's own workload (heterogeneous celia-extruded columns) against per-column oracle solves.
"""
import threading

import numpy as np
import pytest

import problems as P
from mpp_amd import constants as C
from test_gpu_parity import _dense, _hetero, _operator_parity
from test_oracle_operators import _mixed_soil_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ncols,nz", [(1, 2), (63, 5), (64, 23), (70, 24), (130, 100), (65, 120), (200, 47)])
@pytest.mark.parametrize("group", [8, 16, 24])
def test_lds_column_solve_against_dense_algebra(ncols, nz, group):
    """M^{-1} through k_tri_solve_lds inside BiCGStab / GMRES: the block-Jacobi solve of a vertical-only grid is exact, so one
    Krylov iteration must return the dense solution; ring sizes 2x4 .. 6x4 levels against every column height"""
    prob, x = _hetero(ncols=ncols, nz=nz, seed=5)
    opts = {"assembly": "columns", "pc_type": "bjacobi", "ksp_type": "bcgs", "tri_lds": 1, "tri_group": group}
    o, g, (rp, cj, val), v = _operator_parity(prob, x, opts)
    exact = np.linalg.solve(_dense(rp, cj, val, v.size), v) if v.size <= 4000 else None
    for ksp in ("bcgs", "gmres"):
        g.soe.SetOption("ksp_type", ksp)
        xs_lds, its, reason = g.soe.KSPSolve(v)
        assert reason > 0 and its <= 2
        g.soe.SetOption("tri_lds", 0)
        xs_mem, its2, reason2 = g.soe.KSPSolve(v)
        g.soe.SetOption("tri_lds", 1)
        assert (its, reason) == (its2, reason2)
        # same arithmetic per column; only the order in which the dot products are summed differs
        assert np.abs(xs_lds - xs_mem).max() <= 1e-12 * np.abs(xs_mem).max()
        if exact is not None:
            assert np.abs(xs_lds - exact).max() <= 1e-8 * np.abs(exact).max()


def test_lds_column_solve_steps_agree_with_the_oracle():
    prob = P.celia1990(ncols=70, nz=40, perm_mult=np.exp(0.4 * np.random.default_rng(5).standard_normal(70)))
    base = {"assembly": "columns", "nseg": 1, "ksp_type": "bcgs", "pc_type": "bjacobi", "spmv_variant": 20}
    on, off = P.to_gpu(prob, dict(base, tri_lds=1)), P.to_gpu(prob, dict(base, tri_lds=0))
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_BCGS, C.PC_BJACOBI)
    for s in range(3):
        r_on, r_off, r_o = on.soe.StepDT(prob.dt, s + 1), off.soe.StepDT(prob.dt, s + 1), o.step_dt(prob.dt, s + 1)
        assert r_on == r_off == tuple(r_o)
    assert on.soe.counters() == off.soe.counters()
    assert on.soe.counters()["newton"] == o.counters()["newton"]
    p_on = on.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    p_off = off.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    assert np.abs(p_on - p_off).max() <= 1e-10 * np.abs(p_off).max()
    assert np.abs(p_on - o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)).max() <= 1e-8 * np.abs(p_off).max()


def test_chained_ilu_timeout_is_redone_with_level_launches():
    """spin budget 0: every dependency wait of the chained triangular solves gives up at once, the application is redone
    per level on the spot and the step's answer is the one of the per-level solve; the fallback is counted"""
    tp = P.thermal_mms(3, 12, 10, 9, replicate_reference_soil_indexing=False)
    ref = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_chain": 0, "ilu_box": 0})
    bad = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_chain": 2, "ilu_chain_spins": 0, "ilu_box": 0})
    ok = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_chain": 2, "ilu_box": 0})
    for g in (ref, bad, ok):
        conv, _ = g.soe.StepDT(tp.dt, 1)
        assert conv
    assert bad.soe.comm_stats()["chain_fallbacks"] == 1        # the first application; the chain is off afterwards
    assert ok.soe.comm_stats()["chain_fallbacks"] == 0 and ref.soe.comm_stats()["chain_fallbacks"] == 0
    assert bad.soe.last_linear_iterations == ref.soe.last_linear_iterations == ok.soe.last_linear_iterations
    assert np.array_equal(bad.soe.GetSoln(), ref.soe.GetSoln())
    assert np.array_equal(ok.soe.GetSoln(), ref.soe.GetSoln())


def test_failure_on_one_rank_ends_every_rank_of_an_in_process_group():
    """rank 1 fails at its third scalar read-back (injected); ranks 0 and 2 are, or soon will be, waiting for it in a
    reduction: all three calls must return with an error, none may block"""
    from mpp_amd import parallel as PAR
    from mpp_amd import vsfm as V
    ncols, nz, nranks = 150, 20, 3
    mult = np.exp(0.5 * np.random.default_rng(11).standard_normal(ncols))
    shards = []
    for r in range(nranks):
        c0, nloc = PAR.shard_columns(ncols, nranks, r)
        prob = P.celia1990(ncols=nloc, nz=nz, perm_mult=mult[c0:c0 + nloc])
        shards.append((prob, P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})))
    V.comm_init_local([g.soe for _p, g in shards])
    shards[1][1].soe.SetOption("debug_fail_at_fetch", 3)
    errors = [None] * nranks

    def run(r):
        prob, g = shards[r]
        try:
            g.soe.StepDT(prob.dt, 1)
        except V.MPPError as e:
            errors[r] = str(e)

    th = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in th), "a rank is still blocked"
    assert "injected failure" in errors[1]
    assert errors[0] and "another rank failed" in errors[0]
    assert errors[2] and "another rank failed" in errors[2]


def test_gmres_restart_is_validated():
    from mpp_amd import vsfm as V
    prob = P.celia1990(ncols=3, nz=10)
    g = P.to_gpu(prob, {"ksp_type": "gmres"})
    for bad in (0, -3, 5000):
        with pytest.raises(V.MPPError, match="ksp_gmres_restart"):
            g.soe.SetOption("ksp_gmres_restart", bad)
    for ok in (32, 200, 5):   # (lengths above 31 are taken since round 4: tests/test_gpu_more.py)
        g.soe.SetOption("ksp_gmres_restart", ok)
    assert g.soe.StepDT(prob.dt, 1)[0]


@pytest.mark.parametrize("nx,ny,nz,steps", [(64, 48, 100, 3), (1024, 1024, 100, 2)], ids=["3072_columns", "bench_workload"])
def test_bench_workload_answer_against_per_column_oracle_solves(nx, ny, nz, steps):
    """bench.py's workload -- heterogeneous celia-extruded columns, BiCGStab + block-Jacobi, default tolerances -- with
    sampled columns compared with the CPU oracle solving each of them alone at tight tolerances over the (sub-)steps the GPU
    run accepted.  What separates the two is the Newton truncation of the global solve (rtol 1e-8 on the global norm):
    stated tolerance 1e-6 relative on pressure."""
    import bench
    ncols = nx * ny
    prob = bench.build_problem(ncols, nz, 0, ncols)
    g = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    history = []
    for s in range(steps):
        conv, reason = g.soe.StepDT(prob.dt, s + 1)
        assert conv and reason > 0
        history.append(g.soe.step_history())
        assert abs(sum(history[-1]) - prob.dt) <= 1e-9 * prob.dt
    # (at a million heterogeneous columns the first step is cut to 4 x 900 s: some column fails Newton at 3600 s, and a global
    # solve applies the cut to every column -- the oracle columns replay exactly those sub-steps)
    p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    chk = bench.verify_answer(p, ncols, nz, 0, ncols, history, True, nsample=64)
    assert chk["columns_checked"] == 64
    assert chk["max_rel_err"] <= 1e-6, chk


@pytest.mark.parametrize("linear_porosity,mixed", [(True, False), (False, True), (True, True)], ids=["linear_por", "mixed_sat", "both"])
@pytest.mark.parametrize("route", [{"assembly": "general"}, {"assembly": "columns", "nseg": 1}, {"assembly": "columns", "nseg": 3},
                                   {"assembly": "columns", "nseg": 1, "fused_rows": 0}],
                         ids=["general", "march_complete_rows", "march_segments", "march_plain"])
def test_linear_porosity_and_mixed_saturation_functions(route, linear_porosity, mixed):
    """a5 / a3': PorosityFunctionMod.F90:144-163 in the accumulation term and its derivative (GE:1603-1693), and cells of
    one mesh with different saturation functions (SaturationFunction.F90:564-650), on every assembly route: operators
    against the oracle, then two time steps with the oracle's Newton counts"""
    prob, x = _mixed_soil_problem(ncols=70, nz=19, seed=9, linear_porosity=linear_porosity, mixed=mixed)
    opts = dict({"ksp_type": "bcgs", "pc_type": "bjacobi"}, **route)
    o, g, _csr, _v = _operator_parity(prob, x, opts)
    o.set_solver(C.KSP_BCGS, C.PC_BJACOBI)
    for s in range(2):
        assert g.soe.StepDT(prob.dt, s + 1) == tuple(o.step_dt(prob.dt, s + 1))
    assert g.soe.counters()["newton"] == o.counters()["newton"]
    # VAR_MASS = por(P) den sat vol at the solution (the porosity model enters here too)
    mass_o, mass_g = o.get(C.AUXVAR_INTERNAL, C.VAR_MASS), g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_MASS, -1)
    assert np.abs(mass_g - mass_o).max() <= 1e-8 * np.abs(mass_o).max()
    np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                               o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=1e-8)


def test_linear_porosity_is_refused_where_it_is_not_built():
    """the model belongs to the flow equation (VSFM, and both equations of the coupled TH system: tests/test_gpu_th.py); the soil
    heat-diffusion path has no pressure to evaluate it at"""
    from mpp_amd import vsfm as V
    tp = P.thermal_mms(3, 3, 3, 3)
    g = P.to_thermal_gpu(tp)
    n = tp.mesh.ncells_local
    with pytest.raises(V.MPPError, match="linear porosity"):
        V.VSFMMPPSetSoilPorosityLinearModel(g, 1, np.full(n, 0.4), np.full(n, 101325.0), np.full(n, 1e-7))


@pytest.mark.parametrize("ls", ["basic", "l2", "cp", "bt"])
@pytest.mark.parametrize("route", [{"assembly": "columns", "nseg": 1, "spmv_variant": 20}, {"assembly": "general"}],
                         ids=["march", "general"])
def test_other_line_searches_follow_the_oracle(ls, route):
    """a13: -snes_linesearch_type basic | l2 | cp (PETSc v3.16.2, restated) on the GPU driver: same accept / reject decisions,
    time-step cuts and Newton counts as the oracle over three celia steps, pressures equal to Newton tolerance"""
    prob = P.celia1990(ncols=70, nz=40, perm_mult=np.exp(0.4 * np.random.default_rng(5).standard_normal(70)))
    g = P.to_gpu(prob, dict({"ksp_type": "bcgs", "pc_type": "bjacobi", "snes_linesearch_type": ls}, **route))
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_BCGS, C.PC_BJACOBI)
    o.set_linesearch(ls)
    for s in range(3):
        assert g.soe.StepDT(prob.dt, s + 1) == tuple(o.step_dt(prob.dt, s + 1))
    cg, co = g.soe.counters(), o.counters()
    assert (cg["newton"], cg["time_cuts"], cg["func_evals"]) == (co["newton"], co["time_cuts"], co["func_evals"])
    np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                               o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=1e-8)


@pytest.mark.parametrize("ls", ["bt", "basic", "cp", "l2"])
def test_dynamic_linesearch_sequence_before_a_time_step_cut(ls):
    """solver%use_dynamic_linesearch (SystemOfEquationsBaseType.F90:432-506): a day-long step with six Newton iterations
    allowed fails repeatedly; the alternatives (bt <-> l2; basic / cp -> l2 -> bt) are tried at the same dt before it is
    halved, the alternative stays selected until every one has failed, and the default is restored at the end"""
    prob = P.celia1990(ncols=66, nz=40)
    opts = {"assembly": "columns", "nseg": 1, "ksp_type": "preonly", "pc_type": "bjacobi", "snes_max_it": 6,
            "snes_linesearch_type": ls, "use_dynamic_linesearch": 1}
    g = P.to_gpu(prob, opts)
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_PREONLY, C.PC_BJACOBI, snes_max_it=6)
    o.set_linesearch(ls, True)
    assert g.soe.StepDT(86400.0, 1) == tuple(o.step_dt(86400.0, 1))
    cg, co = g.soe.counters(), o.counters()
    assert cg["time_cuts"] == co["time_cuts"] > 0 and cg["newton"] == co["newton"]
    assert o.linesearch_switches() > 0
    np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                               o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=1e-7)
    with pytest.raises(Exception, match="linesearch"):
        g.soe.SetOption("snes_linesearch_type", "secant")


@pytest.mark.parametrize("name,cfg,loosen", [
    ("vsfm_celia1990", "vsfm.cfg", {"pressure": "1.0e-9 relative", "general": "1.0e-9 absolute"}),
    ("vsfm_vchannel", "vsfm.cfg", {"pressure": "1.0e-9 relative", "general": "1.0e-9 absolute"}),
    ("thermal_mms", "thermal.cfg", {"general": "1.0e-9 relative"}),
    ("thermal_mms_2d", "thermal.cfg", {"general": "1.0e-9 relative"}),
    ("thermal_mms_3d", "thermal.cfg", {"general": "1.0e-9 relative"}),
    ("mass_and_heat", "th.cfg", {"pressure": "1.0e-9 relative"}),
    ("th_mms", "th.cfg", {"pressure": "1.0e-9 relative"}),   # manufactured sources built by the product (mpp_amd.workloads.th_mms)
])
def test_standalone_driver_from_namelist_to_regression_file(golden_dir, tmp_path, name, cfg, loosen):
    """f-3 end to end on the GPU: the reference's own .namelist -> mpp_amd.standalone (problem on the GPU, PETSc's default
    GMRES(30) + ILU(0)) -> RegressionWriter -> compare() against the reference's .regression.baseline with the reference's
    .cfg tolerances, loosened only where DESIGN.md section 5 says so: the cfg's 1e-10 / 1e-12 / 1e-16 absolute tolerances are
    same-binary reproducibility bounds of a PETSc run; this is a different implementation of the same arithmetic, held to
    1e-9 relative (pressure, temperature of the KSP problems) -- the TH temperature keeps the cfg's own 1e-8 K."""
    import os
    import shutil
    from mpp_amd import regression as R
    from mpp_amd import standalone as S
    nl_path = tmp_path / (name + ".namelist")
    shutil.copy(os.path.join(golden_dir, "namelists", name + ".namelist"), nl_path)
    S.main(["-namelist", str(nl_path)])
    out = (tmp_path / (name + ".regression")).read_text()
    gold = open(os.path.join(golden_dir, name + ".regression.baseline")).read()
    tol = R.Tolerances.from_cfg(open(os.path.join(golden_dir, cfg)).read(), name)
    for cat, spec in loosen.items():
        tol.update_from_name(cat, spec)
    status, msgs = R.compare(out, gold, tol)
    assert status == "pass", msgs
    # sections, keys and their order are the reference file's
    assert [l.split("=")[0] for l in out.splitlines()] == [l.split("=")[0] for l in gold.splitlines()]


@pytest.mark.parametrize("sf,lam", [(C.SAT_FUNC_VAN_GENUCHTEN, 0.5), (C.SAT_FUNC_VAN_GENUCHTEN, None), (C.SAT_FUNC_BROOKS_COREY, None),
                                    (C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2, None), (C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3, None)],
                         ids=["vg_half", "vg", "bc", "sbc_bz2", "sbc_bz3"])
@pytest.mark.parametrize("nseg", [1, 3])
def test_soil_classes_and_level_geometry_change_no_bit(sf, lam, nseg):
    """the byte savers of the column march -- one class byte per cell instead of the soil-parameter streams, per-level
    upwind weight / gravity distance instead of the per-connection streams -- feed the same numbers into the same
    arithmetic: residual and Jacobian are bit-identical with them on and off, and equal to the oracle's"""
    rng = np.random.default_rng(21)
    ncols, nz = 130, 23
    prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=np.exp(0.5 * rng.standard_normal(ncols)))
    n = prob.mesh.ncells_local
    classes = [(prob.alpha[0] * a, (lam if lam else l), sr) for a, l, sr in ((1.0, 0.45, 0.2772), (1.15, 0.6, 0.21), (0.9, 0.8, 0.3))]
    pick = rng.integers(0, 3, n)
    prob.satfunc_type = np.full(n, sf, dtype=np.int32)
    prob.alpha = np.array([classes[q][0] for q in pick]); prob.lam = np.array([classes[q][1] for q in pick])
    prob.sat_res = np.array([classes[q][2] for q in pick])
    x = prob.press_ic + rng.uniform(0, 1.2e5, n)
    base = {"assembly": "columns", "nseg": nseg, "ksp_type": "bcgs", "pc_type": "bjacobi"}
    _operator_parity(prob, x, base)                                   # against the oracle, byte savers on (the default)
    out = {}
    for cls in (0, 1):
        for geo in (0, 1):
            g = P.to_gpu(prob, dict(base, soil_classes=cls, level_geometry=geo))
            g.soe.SetDtime(prob.dt); g.soe.PreSolve()
            out[cls, geo] = (g.soe.ComputeResidual(x), g.soe.ComputeJacobian(x)[2])
    for key in ((0, 1), (1, 0), (1, 1)):
        assert np.array_equal(out[key][0], out[0, 0][0]) and np.array_equal(out[key][1], out[0, 0][1]), key


@pytest.mark.parametrize("sf", [C.SAT_FUNC_VAN_GENUCHTEN, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2],
                         ids=["vg", "sbc_bz3", "sbc_bz2"])
@pytest.mark.parametrize("nclasses", [700, 5000])
def test_more_than_256_soil_tuples_stream_the_per_cell_arrays(sf, nclasses):
    """soil maps with more than 256 distinct parameter tuples get no class table: the per-cell parameter streams are used (for
    smoothed Brooks-Corey they carry p_u and the one non-zero smoothing coefficient, and p_s = -0.9 / alpha is formed on the
    device) -- the oracle's residual and Jacobian, and the same bits whether classes are asked for or not"""
    rng = np.random.default_rng(33)
    ncols, nz = 200, 31
    prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=np.exp(0.5 * rng.standard_normal(ncols)))
    n = prob.mesh.ncells_local
    ta, tl, ts = 1.0 + 0.2 * rng.random(nclasses), 0.4 + 0.3 * rng.random(nclasses), 0.2 + 0.1 * rng.random(nclasses)
    pick = rng.integers(0, nclasses, n)
    prob.satfunc_type = np.full(n, sf, dtype=np.int32)
    prob.alpha = prob.alpha * ta[pick]; prob.lam = tl[pick]; prob.sat_res = ts[pick]
    x = prob.press_ic + rng.uniform(0, 1.2e5, n)
    base = {"assembly": "columns", "nseg": 1, "ksp_type": "bcgs", "pc_type": "bjacobi"}
    _operator_parity(prob, x, base)
    out = {}
    for cls in (0, 1):
        g = P.to_gpu(prob, dict(base, soil_classes=cls))
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        out[cls] = (g.soe.ComputeResidual(x), g.soe.ComputeJacobian(x)[2])
        assert g.soe.assembly_info()["soil_classes"] == 0
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])


def test_level_geometry_falls_back_when_columns_differ():
    """one column with a different layer spacing: the per-connection streams are used (and the answer is the oracle's)"""
    prob, x = _hetero(ncols=70, nz=19, seed=12)
    cs = prob.mesh.internal_connections()
    cs.dist_up = np.asarray(cs.dist_up, dtype=np.float64).copy(); cs.dist_dn = np.asarray(cs.dist_dn, dtype=np.float64).copy()
    cs.dist_up[5] *= 1.3; cs.dist_dn[40] *= 0.7
    _operator_parity(prob, x, {"assembly": "columns", "nseg": 1})
    _operator_parity(prob, x, {"assembly": "columns", "nseg": 3})


@pytest.mark.parametrize("name,ptype,dims,ncell", [("thermal_mms_3d", 3, (3, 3, 3), 3), ("thermal_mms_2d", 2, (3, 3, 1), 2)])
def test_fortran_hosted_thermal_mms(golden_dir, name, ptype, dims, ncell):
    """fortran/thermal_mms_gpu.F90: the thermal KSP problem driven from Fortran through sysofeqns_thermal_gpu_type (the type-bound
    Create / SetSoils / SetSolnPrevCLM / PreStepDT / StepDTKSP / GetSoln of fortran/mppgpu_mod.F90); mesh, soil and condition
    arrays are handed over on standard input; the printed regression block is compared with the reference's baseline"""
    import os
    import subprocess
    from mpp_amd import regression as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fortran", "_build", "thermal_mms_gpu")
    if not os.path.exists(exe):
        pytest.skip("no Fortran compiler was available at build time")
    tp = P.thermal_mms(ptype, *dims)
    m, cs, bc = tp.mesh, tp.mesh.internal_connections(), tp.bc
    lines = ["%d %d %d %d %d" % (m.ncells_local, m.nlev, cs.num_connections, bc.num_connections, ncell),
             "%r %r %r" % (float(tp.t_init), float(tp.cnfac), float(tp.dt))]
    for arr in (m.vol, m.dz, cs.id_up, cs.id_dn, cs.dist_up, cs.dist_dn, cs.area, np.asarray(cs.unit_vec).ravel(),
                bc.id_dn, bc.dist_up, bc.dist_dn, bc.area, np.asarray(bc.unit_vec).ravel(), tp.bc_temperature, tp.tkdry, tp.heat_source):
        a = np.asarray(arr)
        lines.append(" ".join(("%d" % v) if a.dtype.kind in "iu" else repr(float(v)) for v in a.ravel()))
    run = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    if run.returncode != 0 and "error while loading shared libraries" in run.stderr:
        pytest.skip("Fortran runtime libraries are not installed on this box: " + run.stderr.strip())
    assert run.returncode == 0, run.stdout + run.stderr
    gold = open(os.path.join(golden_dir, name + ".regression.baseline")).read()
    tol = R.Tolerances()
    tol.update_from_name("general", "1.0e-9 relative")
    status, msgs = R.compare(run.stdout, gold, tol)
    assert status == "pass", msgs


@pytest.mark.parametrize("dims,box_len", [((37, 19, 11), 0), ((37, 19, 11), 5), ((70, 9, 3), 16), ((23, 17, 1), 0), ((20, 1, 1), 0),
                                          ((64, 16, 8), 64), ((33, 40, 21), 32)]
                         # odd and user-chosen box lengths, the tail where the row pairs do not divide the steps, boxes longer than
                         # 64 (faces beyond one LDS batch, the gather through the slot map instead of the LDS image), 2-D grids
                         + [(d, bl) for d in ((5, 3, 1), (33, 7, 9), (130, 9, 17)) for bl in (5, 32, 128)])
def test_box_ilu_solves_reproduce_the_level_scheduled_ones_bit_for_bit(dims, box_len):
    """ILU(0) triangular solves of a structured grid box by box (one wave per box, dependencies through registers and cross-lane
    moves) against one launch per dependency level: per row the entries are eliminated in the same order, so the Krylov
    iterates, the iteration count and the answer must be the same bits; ragged extents, 1-D / 2-D / 3-D, several box lengths"""
    ptype = 3 if dims[2] > 1 else 2 if dims[1] > 1 else 1
    tp = P.thermal_mms(ptype, *dims, replicate_reference_soil_indexing=False)
    box = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_len": box_len})
    prg = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_len": box_len, "ilu_box_factor": 0})
    lev = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 0, "ilu_chain": 0})
    for g in (box, prg, lev):   # factorised and solved by boxes; factorised by the elimination program, solved by boxes; neither
        conv, _ = g.soe.StepDT(tp.dt, 1)
        assert conv
    assert box.soe.last_linear_iterations == lev.soe.last_linear_iterations == prg.soe.last_linear_iterations
    assert np.array_equal(box.soe.GetSoln(), lev.soe.GetSoln())
    assert np.array_equal(prg.soe.GetSoln(), lev.soe.GetSoln())
    bi, li = box.soe.pc_info(), lev.soe.pc_info()
    assert bi["path"] == "boxes" and li["path"] == "levels" and bi["launches"] < li["launches"] or min(dims) == 1
    assert bi["box_extents"][1] * bi["box_extents"][2] == 64
    # ... and with ONE launch per triangle (boxes by ticket in level order, completion flags; round 4): the same bits again, on many
    # resident waves and on three (every box then waits for boxes the same wave solved a moment ago, or that two others are solving)
    for mode in ("ilu_box_queue", "ilu_box_stream"):   # (streams: pencils of boxes with faces handed over 8 steps at a time; 8 x 8 cross-sections only)
        for workers in (1024, 3):
            que = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_len": box_len, mode: 1,
                                        "ilu_box_workers": workers})
            assert que.soe.StepDT(tp.dt, 1)[0]
            assert que.soe.last_linear_iterations == lev.soe.last_linear_iterations
            assert np.array_equal(que.soe.GetSoln(), lev.soe.GetSoln())
            assert que.soe.comm_stats()["chain_fallbacks"] == 0
            if mode == "ilu_box_queue":
                assert que.soe.pc_info()["launches"] == 4 or min(dims) == 1
            elif dims[2] >= 8:
                assert que.soe.pc_info()["launches"] == 4      # the streamed route was really taken
    # ... and with the solution leaving the box layout through the LDS image of a box instead of the per-row slot map
    img = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_len": box_len, "ilu_box_img_out": 1})
    assert img.soe.StepDT(tp.dt, 1)[0]
    assert img.soe.last_linear_iterations == lev.soe.last_linear_iterations
    assert np.array_equal(img.soe.GetSoln(), lev.soe.GetSoln())


@pytest.mark.parametrize("dims", [(2, 9, 5), (9, 2, 5), (11, 7, 2), (13, 2, 1), (3, 3, 3)])
def test_box_ilu_on_grids_with_an_extent_of_two(dims):
    """with an extent of 2 an upper entry of a pivot row can meet an off-diagonal entry of the row being eliminated, so the
    factorisation by boxes (diagonal updates only) does not apply: those grids are factorised by the general elimination program
    and only solved box by box -- same bits either way; 3 x 3 x 3 is the smallest grid factorised by boxes"""
    ptype = 3 if dims[2] > 1 else 2
    tp = P.thermal_mms(ptype, *dims, replicate_reference_soil_indexing=False)
    box = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2})
    prg = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_factor": 0})
    lev = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 0, "ilu_chain": 0})
    for g in (box, prg, lev):
        conv, _ = g.soe.StepDT(tp.dt, 1)
        assert conv
    assert box.soe.last_linear_iterations == lev.soe.last_linear_iterations == prg.soe.last_linear_iterations
    assert np.array_equal(box.soe.GetSoln(), lev.soe.GetSoln())
    assert np.array_equal(prg.soe.GetSoln(), lev.soe.GetSoln())


def test_box_ilu_applications_on_the_3d_vsfm_jacobian():
    """the same on the 7-point Jacobian of the v-channel problem (VSFM, z-fastest caller numbering, 63 K rows): forty
    applications in a row (a stale operand would show up now and then), then a whole time step"""
    prob = P.vchannel(nx=40, ny=36, nz=44)
    n = prob.mesh.ncells_local
    x = prob.press_ic - np.random.default_rng(2).uniform(0, 2.5e5, n)
    box = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2})
    que = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_queue": 1})   # one launch per triangle
    stm = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_stream": 1})  # ... as streams of pencils
    stm0 = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_stream": 1, "ilu_box_fused_io": 0})   # (gather passes kept)
    img = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2, "ilu_box_img_out": 1})   # the way back through an LDS image
    lev = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 0, "ilu_chain": 0})
    for g in (box, que, stm, stm0, img, lev):
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        g.soe.ComputeJacobian(x)
    rng = np.random.default_rng(17)
    for rep in range(40):
        v = rng.standard_normal(n)
        z = lev.soe.PCApply(v)
        assert np.array_equal(box.soe.PCApply(v), z), rep
        assert np.array_equal(que.soe.PCApply(v), z), rep
        assert np.array_equal(stm.soe.PCApply(v), z), rep
        assert np.array_equal(stm0.soe.PCApply(v), z), rep
        assert np.array_equal(img.soe.PCApply(v), z), rep
    assert que.soe.comm_stats()["chain_fallbacks"] == 0 and stm.soe.comm_stats()["chain_fallbacks"] == 0
    assert stm.soe.pc_info()["launches"] == 4
    assert box.soe.StepDT(prob.dt, 1) == lev.soe.StepDT(prob.dt, 1)
    assert box.soe.counters() == lev.soe.counters()
    assert np.array_equal(box.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                          lev.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1))


def test_ilu_box_option_can_be_switched_on_after_a_first_setup_without_it():
    """a handle whose first preconditioner set-up ran with ilu_box = 0 has not been tested for a structured grid yet: switching the
    option on later must run the test then (1: boxes are used; 2: no 'not a structured grid' error for a grid that is one)"""
    tp = P.thermal_mms(3, 13, 9, 7, replicate_reference_soil_indexing=False)
    g = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 0})
    ref = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2})
    assert g.soe.StepDT(tp.dt, 1)[0] and ref.soe.StepDT(tp.dt, 1)[0]
    assert g.soe.pc_info()["path"] == "levels" and ref.soe.pc_info()["path"] == "boxes"
    t1 = g.soe.GetSoln()
    assert np.array_equal(t1, ref.soe.GetSoln())
    for mode in (1, 2):
        g.soe.SetOption("ilu_box", mode)
        assert g.soe.StepDT(tp.dt, 2)[0]
        assert g.soe.pc_info()["path"] == "boxes"
    ref.soe.StepDT(tp.dt, 2); ref.soe.StepDT(tp.dt, 3)
    assert np.array_equal(g.soe.GetSoln(), ref.soe.GetSoln())


def test_box_ilu_is_refused_by_name_on_a_pattern_that_is_not_a_grid():
    """the v-channel grid with one lateral connection missing is not a grid any more: ilu_box = 2 must say so, the default
    falls back to the level-scheduled solves"""
    from mpp_amd import mesh as M
    from mpp_amd import vsfm as V

    def holed():
        prob = P.vchannel(nx=6, ny=5, nz=7)
        cs = prob.mesh.intrn_conn_sets[0]
        prob.mesh.intrn_conn_sets = [M.ConnectionSet(cs.id_up[:-1], cs.id_dn[:-1], cs.dist_up[:-1], cs.dist_dn[:-1], cs.area[:-1],
                                                     cs.unit_vec[:-1], None if cs.itype is None else cs.itype[:-1])]
        return prob
    prob = holed()
    x = prob.press_ic - np.random.default_rng(3).uniform(0, 2e4, prob.press_ic.size)
    strict = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 2})
    strict.soe.SetDtime(prob.dt); strict.soe.PreSolve(); strict.soe.ComputeJacobian(x)
    with pytest.raises(V.MPPError, match="not a structured grid"):
        strict.soe.PCApply(np.ones(x.size))
    dflt = P.to_gpu(holed(), {"ksp_type": "gmres", "pc_type": "ilu"})
    lev = P.to_gpu(holed(), {"ksp_type": "gmres", "pc_type": "ilu", "ilu_box": 0})
    for g in (dflt, lev):
        g.soe.SetDtime(prob.dt); g.soe.PreSolve(); g.soe.ComputeJacobian(x)
    v = np.random.default_rng(5).standard_normal(x.size)
    assert np.array_equal(dflt.soe.PCApply(v), lev.soe.PCApply(v))


@pytest.mark.parametrize("nranks,nz,boxed", [(2, 12, True), (3, 12, True), (3, 14, False)])
def test_box_ilu_on_ranks_with_ghost_planes(nranks, nz, boxed):
    """a 3-D grid split into contiguous id ranges with ghost cells: ghost rows are identity rows with a zero right-hand side in
    the preconditioner, so a rank's ILU(0) is that of its block of owned cells.  Whole z-slabs (12 planes over 2 or 3 ranks) are
    structured grids again and are factorised and solved by boxes; 14 planes over 3 ranks cut through planes and every rank falls
    back to the general route by itself.  Same bits as the general route on the same partition either way, and the
    single-handle answer to the linear tolerance"""
    from mpp_amd import vsfm as V
    from test_gpu_more import _run_ranks
    tp = P.thermal_mms(3, 11, 9, nz, replicate_reference_soil_indexing=False)
    ref = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-12})
    ref.soe.PreStepDT()
    assert ref.soe.StepDT(tp.dt)[0]
    t_ref = ref.soe.GetSoln()
    answers = {}
    for box in (2 if boxed else 1, 0):
        opts = {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-12, "ilu_box": box, "ilu_chain": 0}
        parts = P.split_thermal_problem(tp, nranks)
        gs = [P.to_thermal_gpu(lp, opts) for _, lp in parts]
        V.comm_init_local([g.soe for g in gs])

        def step(g):
            g.soe.PreStepDT()
            return g.soe.StepDT(tp.dt)

        res = _run_ranks([lambda g=g: step(g) for g in gs])
        assert all(r[0] for r in res)
        assert all(g.soe.pc_info()["path"] == ("boxes" if box and boxed else "levels") for g in gs)
        answers[box] = [(part, g.soe.GetSoln(), g.soe.last_linear_iterations) for (part, _), g in zip(parts, gs)]
    for (part, tb, ib), (_, tl, il) in zip(answers[2 if boxed else 1], answers[0]):
        assert ib == il
        assert np.array_equal(tb[:part.nowned], tl[:part.nowned])
        assert np.abs(tb[:part.nowned] - t_ref[part.l2g[:part.nowned]]).max() <= 1e-8
