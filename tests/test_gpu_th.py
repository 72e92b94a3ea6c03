"""GPU parity tests of the coupled thermal-hydrology system: every call goes through the C
ABI (libmppgpu.so) and is compared with the CPU oracle (oracle/th_oracle.c, pinned on the reference's th baselines)
on the same inputs, and with the reference's golden baselines.

Tolerances (fp64): the HIP path evaluates the same expressions as the oracle with the device pow/sqrt and FMA
contraction: residual rows <= 1e-11 of the size of their cancelling terms, Jacobian entries <= 1e-10 of the row
scale of their block; end-to-end <= 1e-9 relative on pressure, <= 1e-8 K on temperature (the reference's tolerance).
"""
import os

import numpy as np
import pytest

import problems as P
from conftest import parse_regression_file
from mpp_amd import constants as C

pytestmark = pytest.mark.gpu


def _dense(rp, cj, val, n):
    A = np.zeros((n, n))
    for i in range(n):
        A[i, cj[rp[i]:rp[i + 1]]] = val[rp[i]:rp[i + 1]]
    return A


def _state(tp, seed=0, dp=(-2e4, 3e4), dt=(0.0, 20.0)):
    n = tp.mesh.ncells_local
    rng = np.random.default_rng(seed)
    return np.concatenate([tp.press_ic + rng.uniform(*dp, n), tp.temp_ic + rng.uniform(*dt, n)])


@pytest.mark.parametrize("case", ["mass_and_heat", "th_mms"])
def test_th_operators(case):
    """residual of both equations and all four Jacobian blocks against the oracle at a random state"""
    tp = P.mass_and_heat(nx=37) if case == "mass_and_heat" else P.th_mms(nx=23)
    n = tp.mesh.ncells_local
    o = P.to_th_oracle(tp)
    g = P.to_th_gpu(tp)
    x = _state(tp, seed=4)
    o.set_dtime(tp.dt); o.pre_solve()
    g.soe.SetDtime(tp.dt); g.soe.PreSolve()
    f_o, f_g = o.residual(x), g.soe.ComputeResidual(x)
    rp, cj, val = o.jacobian(x)
    rp_g, cj_g, val_g = g.soe.ComputeJacobian(x)
    assert np.array_equal(rp, rp_g) and np.array_equal(cj, cj_g)
    J, Jg = _dense(rp, cj, val, 2 * n), _dense(rp_g, cj_g, val_g, 2 * n)
    for (r0, c0) in ((0, 0), (0, n), (n, 0), (n, n)):
        B, Bg = J[r0:r0 + n, c0:c0 + n], Jg[r0:r0 + n, c0:c0 + n]
        scale = np.abs(B).max(axis=1, keepdims=True) + 1e-300
        assert (np.abs(Bg - B) <= 1e-10 * scale).all(), (r0, c0, np.abs((Bg - B) / scale).max())
    # residual: against the size of the cancelling terms ~ |J| |dx| + |F|
    dx = np.concatenate([np.full(n, 1.0), np.full(n, 1e-3)])
    fscale = np.abs(f_o) + np.abs(J) @ dx
    fscale[n:] += np.abs(np.diag(J)[n:]) * 300.0   # heat rows: differences of accumulation terms ~ dE/dT * T
    assert (np.abs(f_g - f_o) <= 1e-11 * fscale + 1e-25).all(), np.abs((f_g - f_o) / fscale).max()
    # SpMV and ILU(0) of the 2N x 2N system in the caller's [P; T] ordering
    v = np.random.default_rng(1).standard_normal(2 * n)
    y_g = g.soe.MatMult(v)
    assert np.abs(y_g - Jg @ v).max() <= 1e-12 * np.abs(Jg @ v).max()
    g.soe.SetOption("pc_type", "ilu")
    z_g = g.soe.PCApply(v)
    z_o = o.glob.pc_apply(C.PC_ILU0, rp, cj, val, v)
    assert np.abs(z_g - z_o).max() <= 1e-9 * np.abs(z_o).max()


def test_th_mass_and_heat_baseline(golden_dir):
    """regression_tests/th/mass_and_heat end to end on the GPU: Newton + GMRES(30) + ILU(0), same iteration counts
    as the oracle, every value of the reference's baseline to 1e-9 (pressure, relative) / 1e-8 K"""
    gold = parse_regression_file(os.path.join(golden_dir, "mass_and_heat.regression.baseline"))
    tp = P.mass_and_heat()
    g = P.to_th_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu"})
    conv, reason = g.soe.StepDT(tp.dt, 1)
    assert conv and reason > 0
    p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    t = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605)
    sp, stt = P.regression_stats(p, 5), P.regression_stats(t, 5)
    for key in ("min", "max", "mean"):
        assert sp[key] == pytest.approx(gold["liquid_pressure"][key], rel=1e-9), key
        assert stt[key] == pytest.approx(gold["temperature"][key], abs=1e-8), key
    for cell, v in sp["cells"].items():
        assert v == pytest.approx(gold["liquid_pressure"]["cell %d" % cell], rel=1e-9), cell
    for cell, v in stt["cells"].items():
        assert v == pytest.approx(gold["temperature"]["cell %d" % cell], abs=1e-8), cell
    o = P.to_th_oracle(tp); o.step_dt(tp.dt)
    co, cg = o.counters(), g.soe.counters()
    assert cg["newton"] == co["newton"] and cg["linear"] == co["linear"]


def test_th_mms_baseline(golden_dir):
    gold = parse_regression_file(os.path.join(golden_dir, "th_mms.regression.baseline"))
    tp = P.th_mms()
    g = P.to_th_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu"})
    conv, reason = g.soe.StepDT(tp.dt, 1)
    assert conv and reason > 0
    p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    t = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605)
    sp, stt = P.regression_stats(p, 5), P.regression_stats(t, 5)
    for cell, v in sp["cells"].items():
        assert v == pytest.approx(gold["liquid_pressure"]["cell %d" % cell], rel=1e-9), cell
    for cell, v in stt["cells"].items():
        assert v == pytest.approx(gold["temperature"]["cell %d" % cell], abs=1e-8), cell
    assert g.soe.counters()["newton"] == 10


def test_th_error_paths():
    from mpp_amd import vsfm as V
    tp = P.mass_and_heat(nx=5)
    g = P.to_th_gpu(tp)
    with pytest.raises(V.MPPError):
        g.soe.SetOption("pc_type", "bjacobi"); g.soe.SetDtime(1.0); g.soe.PreSolve(); g.soe.PCApply(np.zeros(10))
    with pytest.raises(V.MPPError):
        g.soe.SetBoundaryAuxVarTemperature(1, np.zeros(1))       # condition 1 belongs to the heat equation
    with pytest.raises(V.MPPError):
        g.Restart(np.zeros(5), np.zeros(4))


@pytest.mark.parametrize("assembly", ["general", "columns"])
@pytest.mark.parametrize("ncols,nz", [(3, 20), (70, 9)])
def test_th_columns_vs_oracle(ncols, nz, assembly):
    """soil columns with gravity, Dirichlet boundaries on both equations, permuted internal layout (column tiles):
    operators and two coupled time steps against the oracle"""
    _th_columns_vs_oracle(ncols, nz, assembly, mixed=False)


@pytest.mark.parametrize("assembly", ["general", "columns"])
def test_th_columns_with_mixed_saturation_functions(assembly):
    """satParams belong to the auxvar (ThermalEnthalpySoilAuxType.F90:156-157, 240-249), so the cells of one TH mesh may use
    different retention curves: van Genuchten, Brooks-Corey and both smoothed Brooks-Corey kinds side by side, every cell its own
    parameters -- operators and two coupled steps against the oracle (parity unpinned by the reference: no fixture mixes them)"""
    _th_columns_vs_oracle(70, 9, assembly, mixed=True)


@pytest.mark.parametrize("assembly", ["general", "columns"])
def test_th_columns_with_the_linear_porosity_model(assembly):
    """POROSITY_LINEAR in both equations (PorosityFunctionMod.F90:144-163 at the cell pressure, ThermalEnthalpySoilAuxType.F90:246-249):
    porosity(P) in both accumulation terms, dpor_dP in the mass equation's diagonal (GE:1675-1677) and in d(energy)/dP
    (GoveqnThermalEnthalpySoilType.F90:2178-2182); together with mixed saturation functions; against the oracle, whose
    derivative is checked by finite differences (tests/test_oracle_operators.py).  Parity unpinned by the reference."""
    _th_columns_vs_oracle(70, 9, assembly, mixed=True, linear_porosity=True)


def _th_columns_vs_oracle(ncols, nz, assembly, mixed, linear_porosity=False, seed=5, density_type=None, ie_type=2,
                          solver=("gmres", "ilu"), extra=None, jtol=1e-10, tp=None):
    rng = np.random.default_rng(seed)
    if tp is None:
        tp = P.th_columns(ncols=ncols, nz=nz, perm_mult=np.exp(0.3 * rng.standard_normal(ncols)), density_type=density_type, ie_type=ie_type)
    n = tp.mesh.ncells_local
    if linear_porosity:
        tp.por_pressure_reference = np.full(n, 101325.0) + rng.uniform(-2e4, 2e4, n)
        tp.por_slope = rng.uniform(1e-7, 6e-7, n)
    if mixed:
        kinds = np.array([C.SAT_FUNC_VAN_GENUCHTEN, C.SAT_FUNC_BROOKS_COREY, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2,
                          C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3], dtype=np.int32)
        tp.satfunc_type = kinds[rng.integers(0, 4, n)]
        tp.alpha = tp.alpha * (1.0 + 0.2 * rng.random(n))
        tp.lam = 0.4 + 0.3 * rng.random(n)
        tp.sat_res = 0.2 + 0.1 * rng.random(n)
    o = P.to_th_oracle(tp)
    g = P.to_th_gpu(tp, dict({"ksp_type": solver[0], "pc_type": solver[1], "assembly": assembly}, **(extra or {})))
    if solver != ("gmres", "ilu"):
        o.glob.set_solver({"gmres": C.KSP_GMRES, "bcgs": C.KSP_BCGS}[solver[0]], {"ilu": C.PC_ILU0, "bjacobi": C.PC_BJACOBI}[solver[1]])
    x = _state(tp, seed=seed + 2 if seed != 5 else 2, dp=(0.0, 9e4), dt=(0.0, 15.0))
    o.set_dtime(tp.dt); o.pre_solve()
    g.soe.SetDtime(tp.dt); g.soe.PreSolve()
    rp, cj, val = o.jacobian(x)
    rp_g, cj_g, val_g = g.soe.ComputeJacobian(x)
    assert np.array_equal(rp, rp_g) and np.array_equal(cj, cj_g)
    J = _dense(rp, cj, val, 2 * n)
    scale = np.zeros(val.size)
    for i in range(2 * n):
        for (a, b) in ((0, n), (n, 2 * n)):
            sel = np.arange(rp[i], rp[i + 1])[(cj[rp[i]:rp[i + 1]] >= a) & (cj[rp[i]:rp[i + 1]] < b)]
            scale[sel] = np.abs(val[sel]).max()
    assert (np.abs(val_g - val) <= jtol * scale + 1e-300).all()
    f_o, f_g = o.residual(x), g.soe.ComputeResidual(x)
    # the heat rows are differences of two accumulation terms of size ~ dE/dT * T
    fscale = np.abs(f_o) + np.abs(J) @ np.concatenate([np.full(n, 1.0), np.full(n, 1e-3)])
    fscale[n:] += np.abs(np.diag(J)[n:]) * 300.0
    assert (np.abs(f_g - f_o) <= 1e-11 * fscale + 1e-25).all(), np.abs((f_g - f_o) / fscale).max()
    v = np.random.default_rng(1).standard_normal(2 * n)
    if solver[1] == "ilu":
        z_g, z_o = g.soe.PCApply(v), o.glob.pc_apply(C.PC_ILU0, rp, cj, val, v)
        assert np.abs(z_g - z_o).max() <= 1e-9 * np.abs(z_o).max()
    for s in range(2):
        P.th_set_conditions(tp, o); P.th_gpu_set_conditions(tp, g)
        co, ro = o.step_dt(tp.dt)
        cg, rg = g.soe.StepDT(tp.dt, s + 1)
        assert co and cg and ro == rg
    p_o, t_o = o.solution()
    p_g = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    t_g = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605)
    assert np.abs(p_g - p_o).max() <= 1e-8 * np.abs(p_o).max()
    assert np.abs(t_g - t_o).max() <= 1e-8
    assert g.soe.counters()["newton"] == o.counters()["newton"]


@pytest.mark.parametrize("ncols,nz", [(3, 20), (130, 7)])
def test_th_block_jacobi_is_the_exact_column_solve(ncols, nz):
    """pc_type bjacobi of the coupled system on a vertical-only grid = block Thomas with 2 x 2 {P,T} blocks per soil
    column: M^-1 is the exact inverse of the Jacobian, the Krylov solver converges at once, the time step agrees with
    the GMRES + ILU(0) solution"""
    rng = np.random.default_rng(8)
    tp = P.th_columns(ncols=ncols, nz=nz, perm_mult=np.exp(0.3 * rng.standard_normal(ncols)))
    n = tp.mesh.ncells_local
    g = P.to_th_gpu(tp, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    x = _state(tp, seed=2, dp=(0.0, 9e4), dt=(0.0, 15.0))
    g.soe.SetDtime(tp.dt); g.soe.PreSolve()
    rp, cj, val = g.soe.ComputeJacobian(x)
    J = _dense(rp, cj, val, 2 * n)
    v = np.random.default_rng(1).standard_normal(2 * n)
    z = g.soe.PCApply(v)
    exact = np.linalg.solve(J, v)
    assert np.abs(z - exact).max() <= 1e-9 * np.abs(exact).max()
    xs, its, reason = g.soe.KSPSolve(v)
    assert reason > 0 and its <= 2
    ref = P.to_th_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-10})
    for s in range(2):
        P.th_gpu_set_conditions(tp, g); P.th_gpu_set_conditions(tp, ref)
        c1, _ = g.soe.StepDT(tp.dt, s + 1)
        c2, _ = ref.soe.StepDT(tp.dt, s + 1)
        assert c1 and c2
    for var, tol in ((C.VAR_PRESSURE, 1e-6), (605, 1e-7)):
        a = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var); b = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var)
        assert np.abs(a - b).max() <= tol * np.abs(b).max()


@pytest.mark.parametrize("ksp,pc", [("bcgs", "bjacobi"), ("gmres", "ilu")])
def test_th_columns_sharded_over_ranks_equals_the_single_handle_solve(ksp, pc):
    """BASELINE config 5 is a multi-GPU run: the soil columns of the coupled problem split over 3 handles (ranks) of this
    process, one host thread each, joined through the in-process communicator (the call sites of the RCCL all-reduce) --
    every rank takes the same Newton / Krylov decisions and the pieces equal the single-handle solution"""
    import threading
    from mpp_amd import parallel as PAR
    from mpp_amd import vsfm as V
    ncols, nz, nranks = 150, 12, 3
    mult = np.exp(0.4 * np.random.default_rng(8).standard_normal(ncols))
    opts = {"ksp_type": ksp, "pc_type": pc}
    ref_tp = P.th_columns(ncols=ncols, nz=nz, perm_mult=mult)
    ref = P.to_th_gpu(ref_tp, opts)
    shards = []
    for r in range(nranks):
        c0, nloc = PAR.shard_columns(ncols, nranks, r)
        tp = P.th_columns(ncols=nloc, nz=nz, perm_mult=mult[c0:c0 + nloc])
        shards.append((c0, nloc, tp, P.to_th_gpu(tp, opts)))
    V.comm_init_local([s[3].soe for s in shards])
    nsteps = 2
    results, errors = [None] * nranks, []

    def run(r):
        try:
            _c0, _nloc, tp, g = shards[r]
            out = []
            for s in range(nsteps):
                P.th_gpu_set_conditions(tp, g)
                out.append(g.soe.StepDT(tp.dt, s + 1))
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
    ref_out = []
    for s in range(nsteps):
        P.th_gpu_set_conditions(ref_tp, ref)
        ref_out.append(ref.soe.StepDT(ref_tp.dt, s + 1))
    p_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE).reshape(ncols, nz)
    t_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605).reshape(ncols, nz)
    for r in range(nranks):
        c0, nloc, _tp, g = shards[r]
        assert results[r] == ref_out
        assert g.soe.counters()["newton"] == ref.soe.counters()["newton"]
        assert g.soe.counters()["linear"] == ref.soe.counters()["linear"]
        p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE).reshape(nloc, nz)
        t = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605).reshape(nloc, nz)
        assert np.abs(p - p_ref[c0:c0 + nloc]).max() <= 1e-9 * np.abs(p_ref).max()
        assert np.abs(t - t_ref[c0:c0 + nloc]).max() <= 1e-9 * np.abs(t_ref).max()


def test_th_compact_spmv_index_of_the_field_blocked_system():
    """the 16-bit SpMV index of the [P; T] system (entries carry their block): MatMult equals the default kernel's and
    the oracle's, and time steps taken with it (line-search reductions inside the SpMV) equal the default path's"""
    ncols, nz = 128, 9                                  # N a multiple of 64: no tile across the block boundary
    tp = P.th_columns(ncols=ncols, nz=nz, perm_mult=np.exp(0.3 * np.random.default_rng(3).standard_normal(ncols)))
    n = tp.mesh.ncells_local
    base = {"ksp_type": "bcgs", "pc_type": "bjacobi"}
    a = P.to_th_gpu(tp, dict(base, spmv_variant=20))
    b = P.to_th_gpu(tp, dict(base, spmv_variant=1, fused_ls_dots=0))
    c = P.to_th_gpu(tp, base)            # default: the index-free product of the column-structured coupled system (k_th_spmv_cols, round 4)
    o = P.to_th_oracle(tp)
    x = _state(tp, seed=4, dp=(0.0, 9e4), dt=(0.0, 15.0))
    for h in (a, b, c):
        h.soe.SetDtime(tp.dt); h.soe.PreSolve()
    o.set_dtime(tp.dt); o.pre_solve()
    rp, cj, val = o.jacobian(x)
    a.soe.ComputeJacobian(x); b.soe.ComputeJacobian(x); c.soe.ComputeJacobian(x)
    v = np.random.default_rng(1).standard_normal(2 * n)
    from oracle import vsfm_oracle as VO
    y_a, y_b, y_o = a.soe.MatMult(v), b.soe.MatMult(v), VO.spmv(rp, cj, val, v)
    assert np.array_equal(y_a, y_b)
    assert np.array_equal(c.soe.MatMult(v), y_b)       # same entry order, products rounded one by one: the same bits
    assert np.abs(y_a - y_o).max() <= 1e-12 * np.abs(y_o).max()
    for s in range(2):
        P.th_gpu_set_conditions(tp, a); P.th_gpu_set_conditions(tp, b)
        assert a.soe.StepDT(tp.dt, s + 1) == b.soe.StepDT(tp.dt, s + 1)
    assert a.soe.counters()["newton"] == b.soe.counters()["newton"]
    for var in (C.VAR_PRESSURE, 605):
        va, vb = a.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var), b.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var)
        assert np.abs(va - vb).max() <= 1e-10 * np.abs(vb).max()


def test_th_at_config_5_size_every_column_gives_the_single_column_answer():
    """BASELINE config 5's size on one GPU (512 x 512 columns x 80 layers, 21 M cells, 42 M unknowns, BiCGStab + per-column
    block-Jacobi): identical columns are independent copies of one problem, so after two coupled time steps every column must
    hold the answer of the CPU oracle solving ONE such column (the global Newton norm sees 262 144 copies, so the stopping
    point can differ by an iteration: compared at the nonlinear tolerance), and all columns must agree with each other to
    rounding -- a size-independent check of the layout, the fused column march and the 2 x 2 block column solve at full size"""
    ncols, nz = 512 * 512, 80
    tp = P.th_columns(ncols=ncols, nz=nz)
    g = P.to_th_gpu(tp, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    one = P.th_columns(ncols=1, nz=nz)
    o = P.to_th_oracle(one)
    for s in range(2):
        P.th_gpu_set_conditions(tp, g); P.th_set_conditions(one, o)
        conv, reason = g.soe.StepDT(tp.dt, s + 1)
        assert conv and reason > 0
        co, ro = o.step_dt(one.dt)
        assert co
    p_o, t_o = o.solution()
    p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE).reshape(ncols, nz)
    t = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605).reshape(ncols, nz)
    assert np.abs(p - p[0]).max() <= 1e-9 * np.abs(p[0]).max() and np.abs(t - t[0]).max() <= 1e-9
    assert np.abs(p[0] - p_o).max() <= 1e-6 * np.abs(p_o).max()
    assert np.abs(t[0] - t_o).max() <= 1e-6
    assert g.soe.counters()["time_cuts"] == 0


@pytest.mark.parametrize("case,nranks", [("vchannel", 3), ("vchannel", 2), ("th_mms", 2)])
def test_th_mesh_split_across_ranks_with_ghost_cells(case, nranks):
    """SURVEY 8(e) for the coupled system: a TH mesh split by WHOLE SOIL COLUMNS (x-y blocks of the column set; th_mms has no
    vertical connection: id ranges) over `nranks` handles of this process -- owned cells + ghost cells, both fields of the ghosts
    refreshed from their owners before every assembly pass and SpMV, ghost rows identity rows.  The unknowns are field-blocked
    [P; T], so the ghost rows are two ranges of every vector (kept zero outside the kernels that read them).  The reference's
    default solver, GMRES(30) + (rank-local) ILU(0): residual rows of the owned cells equal the single-handle rows at a random
    state; the time steps reach the single-handle solution to the linear tolerance, with the single handle's Newton count and its
    linear iterations +- 2 per solve (dense algebra on the oracle's Jacobian: tests/test_oracle_th_split_krylov.py, 29 / 29 / 29)."""
    from mpp_amd import vsfm as V
    from test_gpu_more import _run_ranks
    tp = P.th_vchannel(6, 5, 7) if case == "vchannel" else P.th_mms(24)
    n = tp.mesh.ncells_local
    opts = {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-11}
    ref = P.to_th_gpu(tp, opts)
    x = _state(tp, seed=4, dp=(0.0, 3e4), dt=(0.0, 8.0))
    ref.soe.SetDtime(tp.dt); ref.soe.PreSolve()
    f_ref = ref.soe.ComputeResidual(x)
    parts = P.split_th_problem(tp, nranks)
    gs = [P.to_th_gpu(lp, opts) for _, lp in parts]
    V.comm_init_local([g.soe for g in gs])

    def local(part, v):      # [P; T] of the rank's cells (owned, then ghosts)
        return np.concatenate([v[:n][part.l2g], v[n:][part.l2g]])

    def resid(part, g):
        g.soe.SetDtime(tp.dt); g.soe.PreSolve()
        return g.soe.ComputeResidual(local(part, x))

    fs = _run_ranks([lambda part=part, g=g: resid(part, g) for (part, _), g in zip(parts, gs)])
    for (part, lp), f in zip(parts, fs):
        nl, own = lp.mesh.ncells_local, part.l2g[:part.nowned]
        for fld in (0, 1):
            a, b = f[fld * nl:fld * nl + part.nowned], f_ref[fld * n:(fld + 1) * n][own]
            assert np.abs(a - b).max() <= 1e-11 * np.abs(f_ref[fld * n:(fld + 1) * n]).max()
            assert (f[fld * nl + part.nowned:(fld + 1) * nl] == 0.0).all()       # ghost rows: zero right-hand side
    for s in range(tp.nstep):
        P.th_gpu_set_conditions(tp, ref)
        assert ref.soe.StepDT(tp.dt, s + 1)[0]
    p_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    t_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605)

    def steps(lp, g):
        out = []
        for s in range(tp.nstep):
            P.th_gpu_set_conditions(lp, g)
            out.append(g.soe.StepDT(tp.dt, s + 1))
        return out

    res = _run_ranks([lambda lp=lp, g=g: steps(lp, g) for (_, lp), g in zip(parts, gs)])
    assert all(c for r in res for (c, _) in r)
    assert len({tuple(r) for r in map(tuple, res)}) == 1          # every rank took the same decisions
    c_ref = ref.soe.counters()
    for g in gs:
        c = g.soe.counters()
        assert c["newton"] == c_ref["newton"], (c, c_ref)
        if case == "vchannel":    # whole columns per rank: the split ILU(0) is as good as the global one
            assert abs(c["linear"] - c_ref["linear"]) <= 2 * c_ref["newton"], (c, c_ref)
        # (th_mms is a chain of cells along x with no column to keep whole: cut in the middle it takes 6 iterations per solve against 2)
    if case == "vchannel":
        assert all(part.mode == "columns" for part, _ in parts)
        # halo bytes per SpMV and neighbour = nz x (cells of the edge) x 8 B per field
        for (part, _), g in zip(parts, gs):
            assert all(int(cnt) % 7 == 0 for cnt in part.recv_count)
            hs = g.soe.halo_stats()     # both fields of a ghost cell travel: 16 bytes per ghost cell and exchange
            assert hs["cells_received"] == part.nghost and hs["bytes_received_per_exchange"] == 16 * part.nghost
            assert hs["neighbours"] == len(part.peer) and hs["max_cells_from_one_neighbour"] == int(max(part.recv_count))
    for (part, lp), g in zip(parts, gs):
        own = part.l2g[:part.nowned]
        p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
        t = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605)
        assert np.abs(p[:part.nowned] - p_ref[own]).max() <= 1e-9 * np.abs(p_ref).max()     # measured: 3e-7 Pa of 3e5
        assert np.abs(t[:part.nowned] - t_ref[own]).max() <= 1e-8                          # measured: 6e-11 K


@pytest.mark.parametrize("ncols,nz", [(70, 9), (128, 5), (5, 2)])
def test_th_fused_spmv_and_block_solve_reproduces_the_two_kernels_bit_for_bit(ncols, nz):
    """k_th_spmv_tri (round 4): t = M^-1 A v with its dot products in one kernel -- the forward sweep forms A v from the matrix rows of
    its level instead of reading it -- against the index-free SpMV followed by the block-Thomas solve with fused dots: the same
    entry order and the same elimination order, so two coupled time steps must give the same bits, Newton and linear counts
    (full column tiles take the staged route, the ragged tile of 70 = 64 + 6 columns and the 5-column mesh the direct one)"""
    tp = P.th_columns(ncols=ncols, nz=nz, perm_mult=np.exp(0.3 * np.random.default_rng(5).standard_normal(ncols)))
    base = {"ksp_type": "bcgs", "pc_type": "bjacobi", "profile": 1}   # (profile: the library counts the launches of a kernel class)
    f = P.to_th_gpu(tp, dict(base, th_spmv_tri=1))
    u = P.to_th_gpu(tp, dict(base, th_spmv_tri=0))
    s0 = P.to_th_gpu(tp, dict(base, th_spmv_tri=1, th_stage=0))     # (rows read lane by lane)
    for s in range(2):
        for g in (f, u, s0):
            P.th_gpu_set_conditions(tp, g)
        ru = u.soe.StepDT(tp.dt, s + 1)
        assert f.soe.StepDT(tp.dt, s + 1) == ru and s0.soe.StepDT(tp.dt, s + 1) == ru and ru[0]
    assert f.soe.counters() == u.soe.counters() == s0.soe.counters()
    for var in (C.VAR_PRESSURE, 605):
        vu = u.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var)
        assert np.array_equal(f.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var), vu)
        assert np.array_equal(s0.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var), vu)
    assert f.soe.kernel_stats("spmv_pc")["calls"] > 0 and u.soe.kernel_stats("spmv_pc")["calls"] == 0   # the fused kernel really ran


@pytest.mark.parametrize("ncols,nz", [(70, 9), (128, 5), (5, 2)])
def test_th_block_factorisation_inside_the_first_solve_is_the_factorisation(ncols, nz):
    """k_th_tri_factor<true, .> (round 5, option th_lazy_factor): the block-Thomas factorisation rides on the sweep of BiCGStab's first
    solve z = M^-1 b -- no factorisation pass of its own in a Newton iteration -- against the factorisation kernel followed by the
    stand-alone solve: one th_factor_level / th_forward_level for both, so the same bits, Newton and linear counts; the lazy run
    launches no kernel of the pc_setup class at all.  Then the consumers that need the stand-alone factors (PCApply through the ABI
    reads plane L, which the combined kernel does not write) get them: the same M^-1 r from both handles"""
    tp = P.th_columns(ncols=ncols, nz=nz, perm_mult=np.exp(0.3 * np.random.default_rng(5).standard_normal(ncols)))
    base = {"ksp_type": "bcgs", "pc_type": "bjacobi", "profile": 1}
    lazy = P.to_th_gpu(tp, dict(base, th_lazy_factor=1, th_inv_fly=1))    # (the default: pivots redone by the fused products as well)
    lazy_inv = P.to_th_gpu(tp, dict(base, th_lazy_factor=1, th_inv_fly=0))
    eager = P.to_th_gpu(tp, dict(base, th_lazy_factor=0, th_inv_fly=0))
    for s in range(2):
        for g in (lazy, lazy_inv, eager):
            P.th_gpu_set_conditions(tp, g)
        re = eager.soe.StepDT(tp.dt, s + 1)
        assert lazy.soe.StepDT(tp.dt, s + 1) == re and lazy_inv.soe.StepDT(tp.dt, s + 1) == re and re[0]
    assert lazy.soe.counters() == eager.soe.counters() == lazy_inv.soe.counters()
    for var in (C.VAR_PRESSURE, 605):
        ve = eager.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var)
        assert np.array_equal(lazy.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var), ve)
        assert np.array_equal(lazy_inv.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var), ve)
    assert lazy.soe.kernel_stats("pc_setup")["calls"] == 0 and lazy_inv.soe.kernel_stats("pc_setup")["calls"] == 0
    assert eager.soe.kernel_stats("pc_setup")["calls"] > 0
    r = np.random.default_rng(11).standard_normal(2 * tp.mesh.ncells_local)
    ze = eager.soe.PCApply(r)
    assert np.array_equal(lazy.soe.PCApply(r), ze) and np.array_equal(lazy_inv.soe.PCApply(r), ze)
