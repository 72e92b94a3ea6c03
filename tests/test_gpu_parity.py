"""GPU parity tests proper: every call goes through the C ABI (libmppgpu.so) and is compared with the CPU
oracle on the same seeded inputs, and with the reference's golden baselines.

Tolerances (fp64): the HIP path evaluates the same expressions as the oracle, with the device `pow` (<= a few
ulp from libm) and FMA contraction, so
  * auxvar-level quantities, Jacobian entries, SpMV, preconditioner: <= 1e-12 relative to the row/vector scale;
  * residual: <= 1e-11 of the size of the terms that cancel in it (accumulation + fluxes);
  * end-to-end (24 Newton-solved steps): <= 1e-9 relative on pressure against the reference baseline.
"""
import os

import numpy as np
import pytest

import problems as P
from conftest import parse_regression_file
from mpp_amd import constants as C

pytestmark = pytest.mark.gpu

RTOL_OP = 1e-12
RTOL_E2E = 1e-9


def _hetero(ncols=37, nz=23, seed=0, **kw):
    rng = np.random.default_rng(seed)
    prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=np.exp(0.5 * rng.standard_normal(ncols)))
    n = prob.mesh.ncells_local
    prob.por = prob.por * rng.uniform(0.8, 1.2, n)
    prob.alpha = prob.alpha * rng.uniform(0.8, 1.2, n)
    prob.sat_res = prob.sat_res * rng.uniform(0.8, 1.2, n)
    prob.lam = np.clip(prob.lam * rng.uniform(0.8, 1.2, n), 0.05, 0.95)
    for k, v in kw.items():
        setattr(prob, k, v)
    x = prob.press_ic + rng.uniform(0, 1.2e5, n)   # both sides of the saturation switch and of p0
    return prob, x


def _dense(rp, cj, val, n):
    A = np.zeros((n, n))
    for i in range(n):
        A[i, cj[rp[i]:rp[i + 1]]] = val[rp[i]:rp[i + 1]]
    return A


def _operator_parity(prob, x, options):
    o = P.to_oracle(prob)
    g = P.to_gpu(prob, options)
    o.set_dtime(prob.dt); o.pre_solve()
    g.soe.SetDtime(prob.dt); g.soe.PreSolve()
    n = x.size
    f_o, f_g = o.residual(x), g.soe.ComputeResidual(x)
    rp, cj, val = o.jacobian(x)
    rp_g, cj_g, val_g = g.soe.ComputeJacobian(x)
    assert np.array_equal(rp, rp_g) and np.array_equal(cj, cj_g)
    rowscale = np.zeros(n)
    np.maximum.at(rowscale, np.repeat(np.arange(n), np.diff(rp)), np.abs(val))
    assert np.abs(val_g - val).max() <= RTOL_OP * np.abs(val).max()
    assert (np.abs(val_g - val) <= 1e-10 * np.repeat(rowscale, np.diff(rp))).all()
    # residual: compare against the size of the cancelling terms ~ |J| |x - x0| + |F| (the fluxes of a row are each of the
    # order |J| |dP| and cancel to |F|; x0 = the state the accumulation term was taken at)
    fscale = np.abs(f_o) + rowscale * max(1.0, float(np.abs(x - prob.press_ic).max()) * 1e-3)
    fscale = fscale + np.asarray(prob.por) * 55.5 * np.asarray(prob.mesh.vol) / prob.dt    # ... and the two accumulation terms
    assert (np.abs(f_g - f_o) <= 1e-11 * fscale + 1e-25).all(), np.abs(f_g - f_o).max()
    v = np.random.default_rng(7).standard_normal(n)
    from oracle import vsfm_oracle as VO
    y_o, y_g = VO.spmv(rp, cj, val, v), g.soe.MatMult(v)
    assert np.abs(y_g - y_o).max() <= RTOL_OP * np.abs(y_o).max()
    return o, g, (rp, cj, val), v


@pytest.mark.parametrize("assembly", ["columns", "general"])
def test_operators_celia_extruded(assembly):
    prob, x = _hetero()
    o, g, (rp, cj, val), v = _operator_parity(prob, x, {"assembly": assembly, "pc_type": "bjacobi", "ksp_type": "bcgs"})
    z_o = o.pc_apply(C.PC_BJACOBI, rp, cj, val, v)
    z_g = g.soe.PCApply(v)
    assert np.abs(z_g - z_o).max() <= 1e-10 * np.abs(z_o).max()
    xs, its, reason = g.soe.KSPSolve(v)
    exact = np.linalg.solve(_dense(rp, cj, val, v.size), v)
    assert reason > 0 and its <= 2
    assert np.abs(xs - exact).max() <= 1e-8 * np.abs(exact).max()
    g.soe.SetOption("ksp_type", "gmres")
    xs, its, reason = g.soe.KSPSolve(v)
    assert reason > 0 and its <= 2
    assert np.abs(xs - exact).max() <= 1e-8 * np.abs(exact).max()


@pytest.mark.parametrize("nseg", [1, 3, 23])
def test_column_segments_agree(nseg):
    prob, x = _hetero(ncols=70, nz=23, seed=3)
    _operator_parity(prob, x, {"assembly": "columns", "nseg": nseg})


@pytest.mark.parametrize("sf,den", [(C.SAT_FUNC_BROOKS_COREY, C.DENSITY_CONSTANT),
                                    (C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2, C.DENSITY_IFC67),
                                    (C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3, C.DENSITY_TGDPB01)])
def test_operators_other_constitutive_laws(sf, den):
    prob, x = _hetero(ncols=9, nz=15, seed=5)
    prob.satfunc_type = np.full(x.size, sf, dtype=np.int32)
    prob.density_type = den
    _operator_parity(prob, x, {})


def test_operators_inactive_cells_seepage_and_sources():
    prob, x = _hetero(ncols=6, nz=12, seed=11)
    n = x.size
    act = np.ones(n, dtype=np.int32)
    act[[5, 17, 40]] = 0
    prob.mesh.is_active = act
    # top boundary as a seepage face, plus a mass-rate sink on every cell (ELM-style SOIL_CELLS source)
    t, cs, vals = prob.bcs[0]
    prob.bcs[0] = (C.COND_SEEPAGE_BC, cs, np.full(vals.size, 101325.0))
    from mpp_amd import mesh as M
    ids = np.arange(1, n + 1)
    z = np.zeros(n)
    ss = M.ConnectionSet(np.full(n, -1), ids, z, z, np.ones(n), np.zeros((n, 3)))
    prob.sss = [(C.COND_MASS_RATE, ss, np.random.default_rng(1).uniform(-1e-6, 1e-6, n))]
    _operator_parity(prob, x, {})
    _operator_parity(prob, x, {"assembly": "general"})


def test_operators_vchannel_7pt():
    prob = P.vchannel(nx=6, ny=5, nz=7)
    rng = np.random.default_rng(2)
    x = prob.press_ic - rng.uniform(0, 2.5e5, prob.press_ic.size)
    o, g, (rp, cj, val), v = _operator_parity(prob, x, {"ksp_type": "bcgs", "pc_type": "bjacobi", "ksp_rtol": 1e-12})
    xs, its, reason = g.soe.KSPSolve(v)
    exact = np.linalg.solve(_dense(rp, cj, val, v.size), v)
    assert reason > 0
    assert np.abs(xs - exact).max() <= 1e-7 * np.abs(exact).max()
    g.soe.SetOption("ksp_type", "gmres")
    xs, its, reason = g.soe.KSPSolve(v)
    assert reason > 0
    assert np.abs(xs - exact).max() <= 1e-7 * np.abs(exact).max()


@pytest.mark.parametrize("variant", [0, 1, 17, 20, 21, 22, 23])
def test_spmv_variants(variant):
    """every SpMV kernel variant sums each row in entry order (only FMA contraction differs between them):
    3-point and 7-point patterns, ragged sizes (rows not a multiple of 64, last column tile narrower than 64)"""
    from oracle import vsfm_oracle as VO
    for prob, x in ((_hetero(ncols=150, nz=17, seed=4)), (P.vchannel(nx=9, ny=7, nz=11), None)):
        if x is None:
            x = prob.press_ic - np.random.default_rng(3).uniform(0, 2e5, prob.press_ic.size)
        g0 = P.to_gpu(prob, {"spmv_variant": 0})
        g = P.to_gpu(prob, {"spmv_variant": variant})
        for h in (g0, g):
            h.soe.SetDtime(prob.dt); h.soe.PreSolve()
        rp, cj, val = g0.soe.ComputeJacobian(x)
        g.soe.ComputeJacobian(x)
        v = np.random.default_rng(8).standard_normal(x.size)
        y0, y = g0.soe.MatMult(v), g.soe.MatMult(v)
        y_o = VO.spmv(rp, cj, val, v)
        rowscale = VO.spmv(rp, cj, np.abs(val), np.abs(v))
        assert (np.abs(y - y_o) <= 4e-16 * rowscale).all()
        assert (np.abs(y - y0) <= 4e-16 * rowscale).all()


def _stats_close(stats, gold, rtol):
    for key in ("min", "max", "mean"):
        assert stats[key] == pytest.approx(gold[key], rel=rtol), key
    for cell, v in stats["cells"].items():
        assert v == pytest.approx(gold["cell %d" % cell], rel=rtol), cell


@pytest.mark.parametrize("opts", [{"ksp_type": "bcgs", "pc_type": "bjacobi"}, {"ksp_type": "gmres", "pc_type": "ilu"},
                                  {"ksp_type": "preonly", "pc_type": "bjacobi", "assembly": "general"}])
def test_celia1990_baseline_end_to_end(golden_dir, opts):
    """regression_tests/vsfm/vsfm_celia1990: 24 x 3600 s, compared with the reference's own golden output."""
    gold = parse_regression_file(os.path.join(golden_dir, "vsfm_celia1990.regression.baseline"))
    prob = P.celia1990()
    g = P.to_gpu(prob, opts)
    for s in range(prob.nstep):
        g.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, 1, prob.bcs[0][2])
        g.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, 2, prob.bcs[1][2])
        conv, reason = g.soe.StepDT(prob.dt, s + 1)
        assert conv and reason > 0
    pr = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    sat = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_LIQ_SAT, -1)
    _stats_close(P.regression_stats(pr), gold["liquid_pressure"], RTOL_E2E)
    _stats_close(P.regression_stats(sat), gold["liquid_saturation"], RTOL_E2E)
    # same Newton trajectory as the oracle (which reproduces the reference run digit for digit)
    o = P.to_oracle(prob)
    for s in range(prob.nstep):
        o.step_dt(prob.dt, s + 1)
    assert g.soe.counters()["newton"] == o.counters()["newton"]
    np.testing.assert_allclose(pr, o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=RTOL_E2E)
    for var in (C.VAR_MASS, C.VAR_SOIL_MATRIX_POT):
        np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, var, -1), o.get(C.AUXVAR_INTERNAL, var),
                                   rtol=1e-8)
    for bc in (1, 2):
        np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_BC, C.VAR_MASS_FLUX, bc),
                                   o.get(C.AUXVAR_BC, C.VAR_MASS_FLUX, bc, 1), rtol=1e-6, atol=1e-18)
    np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_CONN_INTERNAL, C.VAR_MASS_FLUX, -1),
                               o.get(C.AUXVAR_CONN_INTERNAL, C.VAR_MASS_FLUX, -1, 99), rtol=1e-6, atol=1e-15)


def test_vchannel_baseline_end_to_end(golden_dir):
    """regression_tests/vsfm/vsfm_vchannel (7-pt, 3 x 8640 s).  The reference solved it with GMRES(30)+ILU(0);
    here BiCGStab + per-column block-Jacobi to a tight linear tolerance, so agreement is to the nonlinear
    tolerance only: 1e-6 relative on pressure (0.5 Pa), 1e-6 absolute on saturation."""
    gold = parse_regression_file(os.path.join(golden_dir, "vsfm_vchannel.regression.baseline"))
    prob = P.vchannel()
    g = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi", "ksp_rtol": 1e-9})
    for s in range(prob.nstep):
        conv, reason = g.soe.StepDT(prob.dt, s + 1)
        assert conv and reason > 0
    pr = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    sat = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_LIQ_SAT, -1)
    _stats_close(P.regression_stats(pr), gold["liquid_pressure"], 1e-6)
    st = P.regression_stats(sat)
    for key in ("min", "max", "mean"):
        assert abs(st[key] - gold["liquid_saturation"][key]) < 1e-6


def test_celia_extruded_every_column_reproduces_the_baseline(golden_dir):
    """on the celia-extruded grid every column must give the single-column golden answer
    (ragged column count: not a multiple of the wave or block size)."""
    gold = parse_regression_file(os.path.join(golden_dir, "vsfm_celia1990.regression.baseline"))
    ncols = 333
    prob = P.celia1990(ncols=ncols)
    g = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    for s in range(prob.nstep):
        conv, reason = g.soe.StepDT(prob.dt, s + 1)
        assert conv and reason > 0
    pr = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1).reshape(ncols, -1)
    for c in (0, 1, 63, 64, 255, 256, ncols - 1):
        _stats_close(P.regression_stats(pr[c]), gold["liquid_pressure"], RTOL_E2E)
    assert np.abs(pr - pr[0]).max() <= 1e-9 * np.abs(pr[0]).max()


def test_time_step_cut_path():
    """dt far too large for Newton: the driver must halve dt (SystemOfEquationsBaseType.F90:500-505) exactly as the oracle."""
    prob = P.celia1990(ncols=2, nz=40)
    opts = {"ksp_type": "preonly", "pc_type": "bjacobi", "snes_max_it": 6}
    g = P.to_gpu(prob, opts)
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_PREONLY, C.PC_BJACOBI, snes_max_it=6)
    conv_g, reason_g = g.soe.StepDT(86400.0, 1)
    conv_o, reason_o = o.step_dt(86400.0, 1)
    assert (conv_g, reason_g) == (conv_o, reason_o)
    cg, co = g.soe.counters(), o.counters()
    assert cg["time_cuts"] == co["time_cuts"] > 0 and cg["newton"] == co["newton"]
    np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                               o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=1e-8)


def test_error_paths():
    from mpp_amd import vsfm as V
    prob = P.celia1990(ncols=1, nz=5)
    g = P.to_gpu(prob)
    with pytest.raises(V.MPPError):
        g.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, 3, np.zeros(1))   # no such condition
    with pytest.raises(V.MPPError):
        g.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, 1, np.zeros(2))   # size(data_1d) > nauxvar
    with pytest.raises(V.MPPError):
        g.Restart(np.zeros(4))
    with pytest.raises(V.MPPError):
        g.soe.SetOption("pc_type", "sor")
    g.soe.SetOption("pc_type", "ilu")          # vertical-only grid: ILU(0) == exact column LU
    P.to_gpu(P.vchannel(nx=3, ny=3, nz=4), {"pc_type": "ilu"})       # lateral connections: level-scheduled ILU(0)
    with pytest.raises(V.MPPError):
        g.soe.SetOption("fused_trials", "many")


@pytest.mark.parametrize("cond_type", [512, 513])
@pytest.mark.parametrize("assembly", ["columns", "general"])
def test_downregulated_sinks(cond_type, assembly):
    """down-regulated potential sinks (Campbell / FETCH2): residual, Jacobian, exported actual rate and two time steps
    against the oracle"""
    from test_oracle_operators import _sink_problem
    prob, x = _sink_problem(cond_type, ncols=70, nz=11)
    o, g, _, _ = _operator_parity(prob, x, {"assembly": assembly})
    o.residual(x); g.soe.ComputeResidual(x)
    for s in range(2):
        co, ro = o.step_dt(prob.dt, s + 1)
        cg, rg = g.soe.StepDT(prob.dt, s + 1)
        assert co and cg and ro == rg
    p_o = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    p_g = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.abs(p_g - p_o).max() <= 1e-8 * np.abs(p_o).max()
    r_o = o.get(C.AUXVAR_SS, C.VAR_MASS_FLUX, 1)
    r_g = g.soe.GetDataForCLM(C.AUXVAR_SS, C.VAR_MASS_FLUX, 1)
    assert np.abs(r_g - r_o).max() <= 1e-9 * np.abs(r_o).max()


def test_lateral_flux_operator_split():
    """soe%ComputeLateralFlux on the GPU: the 'Lateral_flux' source equals the oracle's, the split residual equals the
    coupled 7-point residual at the same state, and a split time step matches the oracle's"""
    split = P.vchannel_operator_split(6, 5, 7)
    n = split.mesh.ncells_local
    x = split.press_ic - np.random.default_rng(9).uniform(0, 2.5e5, n)
    o = P.to_oracle(split, with_blocks=False)
    g = P.to_gpu(split, {"ksp_type": "bcgs", "pc_type": "bjacobi"})
    o.restart(x); g.Restart(x)
    o.compute_lateral_flux(split.dt, 1)
    g.soe.ComputeLateralFlux(split.dt, 1)
    lat_o = o.get_condition_values(C.COND_SS, 1, n)
    lat_g = g.soe.GetConditionValues(C.COND_SS, 1)
    assert np.abs(lat_g - lat_o).max() <= 1e-12 * np.abs(lat_o).max()
    f_o, f_g = o.residual(x), g.soe.ComputeResidual(x)
    full = P.to_gpu(P.vchannel(6, 5, 7))
    full.Restart(x); full.soe.SetDtime(split.dt); full.soe.PreSolve()
    f_full = full.soe.ComputeResidual(x)
    scale = np.abs(lat_o) / 18.01534 + np.abs(f_o)
    assert (np.abs(f_g - f_o) <= 1e-11 * scale + 1e-25).all()
    assert (np.abs(f_g - f_full) <= 1e-11 * scale + 1e-25).all()
    # one operator-split step from an unsaturated, mildly perturbed state (the explicit lateral source is only
    # conditionally stable): lateral fluxes at the old state, then the implicit vertical solve
    x1 = split.press_ic - 3e5 - np.random.default_rng(3).uniform(0, 2e3, n)   # unsaturated everywhere
    o.restart(x1); g.Restart(x1)
    o.compute_lateral_flux(600.0, 1); g.soe.ComputeLateralFlux(600.0, 1)
    co, ro = o.step_dt(600.0, 1)
    cg, rg = g.soe.StepDT(600.0, 1)
    assert co and cg and ro == rg
    p_o = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    p_g = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.abs(p_g - p_o).max() <= 1e-8 * np.abs(p_o).max()


def test_elm_surface_frac_liq_sat_and_exchanged_mass(golden_dir):
    """ELM coupling variables: per-cell VAR_FRAC_LIQ_SAT (Brooks-Corey relative permeability is scaled by it), the
    PreStepDT / PostStepDT state hand-over, and the exchanged-mass accumulators VAR_BC_MASS_EXCHANGED /
    VAR_LATERAL_MASS_EXCHANGED -- against the oracle"""
    # (1) frac_liq_sat on a Brooks-Corey soil: operators
    prob, x = _hetero(ncols=9, nz=15, seed=5)
    n = x.size
    prob.satfunc_type = np.full(n, C.SAT_FUNC_BROOKS_COREY, dtype=np.int32)
    frac = np.random.default_rng(2).uniform(0.4, 1.0, n)
    for assembly in ("columns", "general"):
        o = P.to_oracle(prob)
        g = P.to_gpu(prob, {"assembly": assembly})
        o.set_frac_liq_sat(frac)
        g.soe.SetDataFromCLM(C.AUXVAR_INTERNAL, C.VAR_FRAC_LIQ_SAT, -1, frac)
        o.set_dtime(prob.dt); o.pre_solve(); g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        f_o, f_g = o.residual(x), g.soe.ComputeResidual(x)
        rp, cj, val = o.jacobian(x)
        _, _, val_g = g.soe.ComputeJacobian(x)
        assert np.abs(val_g - val).max() <= RTOL_OP * np.abs(val).max()
        assert np.abs(f_g - f_o).max() <= 1e-11 * (np.abs(f_o).max() + np.abs(val).max())
        assert np.array_equal(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_FRAC_LIQ_SAT), frac)
        o2 = P.to_oracle(prob); o2.set_dtime(prob.dt); o2.pre_solve()
        assert np.abs(o2.residual(x) - f_o).max() > 1e-6 * np.abs(f_o).max()     # the scaling is not a no-op
    # (2) boundary mass exchanged on the celia column over 3 steps with PreStepDT / PostStepDT around every step
    prob = P.celia1990(ncols=5, nz=40, perm_mult=[1.0, 0.8, 1.3, 0.6, 1.1])
    o = P.to_oracle(prob)
    g = P.to_gpu(prob, {"ksp_type": "bcgs", "pc_type": "bjacobi", "track_mass_exchange": 1})
    for s in range(3):
        o.pre_step_dt(); g.soe.PreStepDT()
        assert o.step_dt(prob.dt, s + 1)[0] and g.soe.StepDT(prob.dt, s + 1)[0]
        o.post_step_dt(); g.soe.PostStepDT()
        for b in (1, 2):
            m_o = o.get(C.AUXVAR_BC, C.VAR_BC_MASS_EXCHANGED, b, 5)
            m_g = g.soe.GetDataForCLM(C.AUXVAR_BC, C.VAR_BC_MASS_EXCHANGED, b)
            assert np.abs(m_g - m_o).max() <= 1e-8 * np.abs(m_o).max() + 1e-12
            # one sub-step per step here: the accumulator holds flux * dt of this step only (reset by PreStepDT)
            assert np.abs(m_g - g.soe.GetDataForCLM(C.AUXVAR_BC, C.VAR_MASS_FLUX, b) * prob.dt).max() <= 1e-10 * np.abs(m_g).max() + 1e-12
    # (3) lateral mass exchanged on the 7-point grid: conservative, equal to the oracle's
    prob = P.vchannel(6, 5, 7)
    o = P.to_oracle(prob, with_blocks=False)
    g = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "track_mass_exchange": 1})
    assert o.step_dt(prob.dt, 1)[0] and g.soe.StepDT(prob.dt, 1)[0]
    l_o = o.get(C.AUXVAR_INTERNAL, C.VAR_LATERAL_MASS_EXCHANGED)
    l_g = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_LATERAL_MASS_EXCHANGED)
    assert np.abs(l_g).max() > 0 and abs(l_g.sum()) <= 1e-10 * np.abs(l_g).sum()
    assert np.abs(l_g - l_o).max() <= 1e-7 * np.abs(l_o).max()


@pytest.mark.parametrize("sat", [False, True])
def test_vsfm_mms_analytic_check_through_the_abi(sat):
    """vsfm_mms (src/driver/standalone/vsfm/vsfm_mms_problem.F90:97-102: PreStepDT, sources, boundary values, one StepDT(1.0)) through
    the per-cell-array setters (:405-418): the library against the oracle at nx = 20 (operators, Newton count, pressures), then the
    discretisation error against P(x) = 15000 sin(pi x / 10) -+ 20000 + 101325 over nx = 20, 40, 80, 160 -- second order on the
    -fully_saturated branch, first order with the upwinded relative permeability of the default branch (RichardsMod.F90:289-293;
    tests/test_oracle_vsfm_mms.py has the oracle's own numbers)"""
    from oracle import adapters as A
    prob = P.vsfm_mms(20, sat)
    x = prob.exact + np.random.default_rng(3).uniform(-3e3, 3e3, prob.exact.size)
    _operator_parity(prob, x, {})
    o = A.to_oracle(A.vsfm_mms(20, sat), with_blocks=False)
    g = P.to_gpu(prob)
    o.pre_step_dt(); g.soe.PreStepDT()
    assert g.soe.StepDT(prob.dt, 1) == tuple(o.step_dt(prob.dt, 1))
    assert g.soe.counters()["newton"] == o.counters()["newton"]
    pg, po = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1), o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.abs(pg - po).max() <= RTOL_E2E * np.abs(po).max()
    sg, so = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_LIQ_SAT, -1), o.get(C.AUXVAR_INTERNAL, C.VAR_LIQ_SAT)
    assert np.abs(sg - so).max() <= 1e-9
    errs = []
    for nx in (20, 40, 80, 160):
        pr = P.vsfm_mms(nx, sat)
        h = P.to_gpu(pr, {"snes_rtol": 1e-10, "snes_stol": 1e-12, "ksp_rtol": 1e-10})
        h.soe.PreStepDT()
        conv, reason = h.soe.StepDT(pr.dt, 1)
        assert conv and reason > 0
        errs.append(float(np.abs(h.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1) - pr.exact).max()))
    lo, hi = (3.9, 4.1) if sat else (1.9, 2.1)
    assert all(lo < a / b < hi for a, b in zip(errs, errs[1:])), errs
    assert errs[0] < (40.0 if sat else 1000.0), errs


def test_vsfm_mms_many_columns_side_by_side():
    """the manufactured column replicated as rows of a 2-D mesh (nx x ny, connections along x only inside a row plus y connections
    between rows that carry no flux: every row holds the same manufactured state) -- the general-assembly route at a size where
    kernels run more than one block; every row reproduces the single column's answer"""
    nx, ny = 40, 70
    one = P.vsfm_mms(nx)
    g1 = P.to_gpu(one, {"snes_rtol": 1e-10, "snes_stol": 1e-12, "ksp_rtol": 1e-10})
    g1.soe.PreStepDT()
    assert g1.soe.StepDT(one.dt, 1)[0]
    p1 = g1.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    from mpp_amd import mesh as M
    m = M.mesh_create("Soil mesh", 10.0, float(ny), 1.0, nx, ny, 1, C.CONN_IN_XYZ_DIR)
    n = m.ncells_local
    tile = lambda a: np.tile(np.asarray(a), ny)
    cells_lo, cells_hi = np.arange(ny) * nx + 1, np.arange(ny) * nx + nx
    dxh = 10.0 / nx / 2.0
    uv = np.zeros((2 * ny, 3)); uv[:ny, 0] = 1.0; uv[ny:, 0] = -1.0
    ends = M.ConnectionSet(np.zeros(2 * ny, dtype=np.int64), np.concatenate([cells_lo, cells_hi]), np.zeros(2 * ny), np.full(2 * ny, dxh),
                           np.ones(2 * ny), uv)
    _t, _cs, bcv = one.bcs[0]
    z = np.zeros(n)
    allcells = M.ConnectionSet(np.full(n, -1), np.arange(1, n + 1), z, z, np.ones(n), np.zeros((n, 3)))
    prob = P.Problem(name="vsfm_mms_rows", mesh=m, perm=np.tile(one.perm, (ny, 1)), por=np.zeros(n), satfunc_type=tile(one.satfunc_type),
                     alpha=tile(one.alpha), lam=tile(one.lam), sat_res=tile(one.sat_res), density_type=one.density_type,
                     press_ic=tile(one.press_ic), bcs=[(C.COND_DIRICHLET, ends, np.concatenate([np.full(ny, bcv[0]), np.full(ny, bcv[1])]))],
                     sss=[(C.COND_MASS_RATE, allcells, tile(one.sss[0][2]))], dt=1.0, nstep=1)
    g = P.to_gpu(prob, {"snes_rtol": 1e-10, "snes_stol": 1e-12, "ksp_rtol": 1e-10})
    g.soe.PreStepDT()
    assert g.soe.StepDT(prob.dt, 1)[0]
    p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1).reshape(ny, nx)
    assert np.abs(p - p1[None, :]).max() <= 1e-8 * np.abs(p1).max()
