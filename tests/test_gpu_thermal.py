"""GPU parity of the thermal KSP path and of the level-scheduled ILU(0), through the C ABI."""
import os

import numpy as np
import pytest

import problems as P
from conftest import parse_regression_file
from mpp_amd import constants as C

pytestmark = pytest.mark.gpu
CASES = [(1, (20, 1, 1), 5, "thermal_mms"), (2, (3, 3, 1), 2, "thermal_mms_2d"), (3, (3, 3, 3), 3, "thermal_mms_3d")]


@pytest.mark.parametrize("ptype,dims,ncell,name", CASES)
def test_thermal_mms_baselines_with_petsc_defaults(golden_dir, ptype, dims, ncell, name):
    """regression_tests/thermal: GMRES(30) + ILU(0), rtol 1e-5 -- the reference's default solver, on the GPU"""
    gold = parse_regression_file(os.path.join(golden_dir, name + ".regression.baseline"))["temperature"]
    tp = P.thermal_mms(ptype, *dims)
    g = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu"})
    conv, _ = g.soe.StepDT(tp.dt, 1)
    assert conv
    st = P.regression_stats(g.soe.GetSoln(), ncell)
    for key in ("min", "max", "mean"):
        assert st[key] == pytest.approx(gold[key], rel=1e-9)
    for cell, v in st["cells"].items():
        assert v == pytest.approx(gold["cell %d" % cell], rel=1e-9)
    o = P.to_thermal_oracle(tp)
    o.step_dt(tp.dt)
    assert g.soe.last_linear_iterations == o.linear_iterations()


@pytest.mark.parametrize("dims", [(7, 5, 6), (13, 1, 9), (24, 20, 16)])
def test_thermal_operator_and_rhs_match_the_oracle(dims):
    tp = P.thermal_mms(3, *dims)
    tp.cnfac = 0.5                                   # exercise the Crank-Nicolson terms too
    g = P.to_thermal_gpu(tp)
    o = P.to_thermal_oracle(tp)
    b_o = o.compute_rhs(1800.0)
    rp, cj, val = o.compute_operators()
    b_g = g.soe.ComputeRHS(1800.0)
    rp_g, cj_g, val_g = g.soe.ComputeOperators(1800.0)
    assert np.array_equal(rp, rp_g) and np.array_equal(cj, cj_g)
    assert np.abs(val_g - val).max() <= 1e-13 * np.abs(val).max()
    assert np.abs(b_g - b_o).max() <= 1e-12 * np.abs(b_o).max()


@pytest.mark.parametrize("ksp,pc", [("gmres", "ilu"), ("bcgs", "ilu"), ("bcgs", "bjacobi"), ("gmres", "jacobi")])
def test_thermal_solve_matches_tight_oracle_solution(ksp, pc):
    tp = P.thermal_mms(3, 12, 10, 9, replicate_reference_soil_indexing=False)   # consistent lambda(x,y,z) per cell
    g = P.to_thermal_gpu(tp, {"ksp_type": ksp, "pc_type": pc, "ksp_rtol": 1e-12})
    conv, _ = g.soe.StepDT(tp.dt, 1)
    assert conv
    o = P.to_thermal_oracle(tp, ksp_rtol=1e-13)
    o.step_dt(tp.dt)
    assert np.abs(g.soe.GetSoln() - o.get_soln()).max() <= 1e-8
    # second-order accurate against the manufactured solution as well
    assert np.abs(g.soe.GetSoln() - tp.exact).max() < 1.0   # 0.75 K on this grid, 0.19 K at twice the resolution


def test_thermal_mms_second_order_at_128_and_256_cubed_with_petsc_defaults():
    """an eighth of BASELINE config 4 (512^3) on one GPU with the reference's solver choice, GMRES(30) + ILU(0), solved box by
    box: the size-independent property of the manufactured-solution problem is second-order convergence, so at a tolerance that
    resolves the discretisation error (PETSc's default 1e-5 on the preconditioned residual does not) the error
    against the manufactured solution must drop fourfold from 128^3 to 256^3 (16.7 M cells, 766 dependency levels)"""
    err = {}
    for n in (128, 256):
        tp = P.thermal_mms(3, n, n, n, replicate_reference_soil_indexing=False)
        g = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-9, "ksp_max_it": 5000})
        conv, _ = g.soe.StepDT(tp.dt, 1)
        assert conv
        assert g.soe.pc_info()["path"] == "boxes"
        err[n] = np.abs(g.soe.GetSoln() - tp.exact).max()
        del g, tp
    assert err[256] < 2e-3 and 3.6 <= err[128] / err[256] <= 4.4, err


def test_config_4_at_full_size_512_cubed_with_petsc_defaults():
    """BASELINE config 4 at its full size on one GPU -- thermal_mms_3d on 512^3 = 134 M cells, GMRES(30) + ILU(0) by boxes -- solved to
    ksp_rtol 1e-8: size-independent properties of the problem and of the kernels instead of an oracle run (the oracle would take hours):
    the manufactured solution is met to the discretisation + linear-solve error measured for this size (0.0052 K; 5.3 K at PETSc's
    default 1e-5); the operator's interior rows sum to zero (pure diffusion, no heat capacity in this problem) and its boundary rows do
    not; the ILU(0) application is linear to rounding; the true residual of the answer is small against the right-hand side"""
    n = 512
    tp = P.thermal_mms(3, n, n, n, replicate_reference_soil_indexing=False)
    g = P.to_thermal_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-8, "ksp_max_it": 6000})
    conv, _ = g.soe.StepDT(tp.dt, 1)
    assert conv and g.soe.pc_info()["path"] == "boxes"
    t = g.soe.GetSoln()
    assert np.isfinite(t).all()
    assert np.abs(t - tp.exact).max() < 8e-3
    b = g.soe.ComputeRHS(tp.dt)                       # (assembles with the same soln_prev; the operator values stay those of the step)
    r = b - g.soe.MatMult(t)
    assert np.linalg.norm(r) <= 1e-5 * np.linalg.norm(b)
    del r
    ones = np.ones(n ** 3)
    y = g.soe.MatMult(ones).reshape(n, n, n)
    inner = np.abs(y[1:-1, 1:-1, 1:-1]).max()
    assert inner <= 1e-9 * np.abs(y).max() and y[0].min() > 0.0 and y[-1].min() > 0.0
    del y, ones
    rng = np.random.default_rng(3)
    v1, v2 = rng.standard_normal(n ** 3), rng.standard_normal(n ** 3)
    z1, z2 = g.soe.PCApply(v1), g.soe.PCApply(v2)
    z3 = g.soe.PCApply(2.0 * v1 - 3.0 * v2)
    assert np.abs(z3 - (2.0 * z1 - 3.0 * z2)).max() <= 1e-9 * max(np.abs(z1).max(), np.abs(z2).max())
    assert g.soe.comm_stats()["chain_fallbacks"] == 0


def test_ilu0_apply_matches_the_oracle_on_the_vchannel_jacobian():
    """level-scheduled ILU(0) in the caller's natural ordering == sequential ILU(0) of the oracle / PETSc"""
    prob = P.vchannel(nx=9, ny=7, nz=8)
    g = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu"})
    o = P.to_oracle(prob)
    for h in (o,):
        h.set_dtime(prob.dt); h.pre_solve()
    g.soe.SetDtime(prob.dt); g.soe.PreSolve()
    x = prob.press_ic - np.random.default_rng(4).uniform(0, 2e5, prob.press_ic.size)
    rp, cj, val = o.jacobian(x)
    g.soe.ComputeJacobian(x)
    r = np.random.default_rng(6).standard_normal(x.size)
    z_o = o.pc_apply(C.PC_ILU0, rp, cj, val, r)
    z_g = g.soe.PCApply(r)
    assert np.abs(z_g - z_o).max() <= 1e-11 * np.abs(z_o).max()


def test_vchannel_baseline_with_petsc_default_solver(golden_dir):
    """regression_tests/vsfm/vsfm_vchannel with the reference's own solver choice (GMRES(30) + ILU(0)) on the GPU:
    now the Krylov iterates follow PETSc's, so the baseline is met to 1e-9 instead of the nonlinear tolerance."""
    gold = parse_regression_file(os.path.join(golden_dir, "vsfm_vchannel.regression.baseline"))
    prob = P.vchannel()
    g = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu"})
    o = P.to_oracle(prob)
    for s in range(prob.nstep):
        conv, reason = g.soe.StepDT(prob.dt, s + 1)
        assert conv and reason > 0
        o.step_dt(prob.dt, s + 1)
    pr = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    st = P.regression_stats(pr)
    for key in ("min", "max", "mean"):
        assert st[key] == pytest.approx(gold["liquid_pressure"][key], rel=1e-9)
    assert g.soe.counters()["newton"] == o.counters()["newton"]


def test_chained_and_per_level_ilu_solves_agree_bit_for_bit():
    """ILU(0) triangular solves as one launch per triangle (tiles of a level chained by completion flags) against one
    launch per level: the arithmetic per row is the same, so every application must give the same bits -- applied many
    times in a row, because a synchronisation slip would show up as an occasional stale operand"""
    prob = P.vchannel(nx=40, ny=36, nz=44)                  # 63 K rows, 118 dependency levels, ragged tiles
    n = prob.mesh.ncells_local
    x = prob.press_ic - np.random.default_rng(2).uniform(0, 2.5e5, n)
    chain = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_chain": 2, "ilu_box": 0})
    level = P.to_gpu(prob, {"ksp_type": "gmres", "pc_type": "ilu", "ilu_chain": 0, "ilu_graph": 0, "ilu_box": 0})
    for g in (chain, level):
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        g.soe.ComputeJacobian(x)
    rng = np.random.default_rng(17)
    for rep in range(40):
        v = rng.standard_normal(n)
        assert np.array_equal(chain.soe.PCApply(v), level.soe.PCApply(v)), rep
    assert chain.soe.StepDT(prob.dt, 1) == level.soe.StepDT(prob.dt, 1)
    assert chain.soe.counters() == level.soe.counters()
    assert np.array_equal(chain.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                          level.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1))


def test_multibase_spmv_index_on_a_grid_with_far_neighbours():
    """a 3-D grid whose y-neighbours lie further than 2^15 rows from their tile (the single-base 16-bit index gives up):
    the windowed index is picked automatically from 2 M rows; here it is forced on a smaller grid with the same property
    and must reproduce the 32-bit kernels"""
    prob = P.vchannel(nx=640, ny=3, nz=100)            # y-neighbour 10 column tiles = 64 000 rows away
    n = prob.mesh.ncells_local
    x = prob.press_ic - np.random.default_rng(2).uniform(0, 2.5e5, n)
    ref = P.to_gpu(prob, {"spmv_variant": 0})
    mb = P.to_gpu(prob, {"spmv_variant": 22})
    mb8 = P.to_gpu(prob, {"spmv_variant": 23})
    for g in (ref, mb, mb8):
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        g.soe.ComputeJacobian(x)
    v = np.random.default_rng(5).standard_normal(n)
    y0, y1, y2 = ref.soe.MatMult(v), mb.soe.MatMult(v), mb8.soe.MatMult(v)
    assert np.abs(y1 - y0).max() <= 4e-16 * np.abs(y0).max() + 1e-300
    assert np.array_equal(y1, y2)
    bad = P.to_gpu(prob, {"spmv_variant": 20})
    bad.soe.SetDtime(prob.dt); bad.soe.PreSolve(); bad.soe.ComputeJacobian(x)
    with pytest.raises(Exception):
        bad.soe.MatMult(v)                               # no single-base index for this pattern


@pytest.mark.parametrize("flux_first", [False, True])
@pytest.mark.parametrize("defects", [False, True])
def test_heat_flux_boundary_and_reference_defects_switch_match_the_oracle(flux_first, defects):
    """COND_HEAT_FLUX with frac and dhsdT (GE :339-354, 909-924, 1199-1217) next to Dirichlet faces; with
    `thermal_reference_defects` the stale-area right-hand side (:906-908) and `dhsdT**area` (:1215) as the reference has them"""
    case = P.thermal_flux_case()
    g = P.to_thermal_flux_gpu(case, flux_first, {"thermal_reference_defects": int(defects)})
    o = P.to_thermal_flux_oracle(case, flux_first, defects=defects)
    b_o = o.compute_rhs(1800.0)
    rp, cj, val = o.compute_operators()
    b_g = g.soe.ComputeRHS(1800.0)
    rp_g, cj_g, val_g = g.soe.ComputeOperators(1800.0)
    assert np.array_equal(rp, rp_g) and np.array_equal(cj, cj_g)
    assert np.abs(val_g - val).max() <= 1e-13 * np.abs(val).max()
    assert np.abs(b_g - b_o).max() <= 1e-12 * np.abs(b_o).max()
    # the switch changes both on this mesh (so the comparison above is not vacuous)
    o2 = P.to_thermal_flux_oracle(case, flux_first, defects=not defects)
    assert np.abs(o2.compute_rhs(1800.0) - b_o).max() > 1e-3 * np.abs(b_o).max()
    # and a step: GMRES(30) + ILU(0) to a tight tolerance on both sides
    g2 = P.to_thermal_flux_gpu(case, flux_first, {"thermal_reference_defects": int(defects), "ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-12})
    conv, _ = g2.soe.StepDT(1800.0, 1)
    assert conv
    o3 = P.to_thermal_flux_oracle(case, flux_first, defects=defects, ksp_rtol=1e-13)
    assert o3.step_dt(1800.0)
    assert np.abs(g2.soe.GetSoln() - o3.get_soln()).max() <= 1e-9 * np.abs(o3.get_soln()).max()   # (ill-conditioned and far from physical with the defects on)


@pytest.mark.parametrize("seed", [11, 12])
def test_land_units_water_ice_snow_and_bedrock_match_the_oracle(seed):
    """every branch of ThermKSPTempSoilAuxVarCompute (auxvar :66-168) -- soil / wetland / land ice, shallow and bedrock layers,
    dry to saturated, frozen and unfrozen, snow water with and without snow layers, tuning factors, inactive cells -- in cells and in the
    boundary auxvars (dist_up > 0, so their conductivity matters), heat-flux and Dirichlet faces"""
    case = P.thermal_flux_case(9, 6, 7, bc_dist_up=0.4)
    tp = case[0]
    n = tp.mesh.ncells_local
    land = P.thermal_land_units(n, seed)
    rng = np.random.default_rng(seed)
    tp.mesh.is_active = (rng.random(n) > 0.08).astype(np.int32)     # filter_thermal = 0 columns' cells: identity rows
    t_prev = 273.15 + 6.0 * (rng.random(n) - 0.5)
    case = (tp, case[1], 273.15 + 8.0 * (rng.random(case[1].num_connections) - 0.5)) + case[3:]
    g = P.to_thermal_flux_gpu(case, False, land=land, t_prev=t_prev)
    o = P.to_thermal_flux_oracle(case, False, land=land, t_prev=t_prev)
    b_o = o.compute_rhs(1800.0)
    rp, cj, val = o.compute_operators()
    b_g = g.soe.ComputeRHS(1800.0)
    rp_g, cj_g, val_g = g.soe.ComputeOperators(1800.0)
    assert np.array_equal(rp, rp_g) and np.array_equal(cj, cj_g)
    assert np.abs(val_g - val).max() <= 1e-13 * np.abs(val).max()
    assert np.abs(b_g - b_o).max() <= 1e-12 * np.abs(b_o).max()
    g2 = P.to_thermal_flux_gpu(case, False, {"ksp_type": "gmres", "pc_type": "ilu", "ksp_rtol": 1e-12}, land=land, t_prev=t_prev)
    conv, _ = g2.soe.StepDT(1800.0, 1)
    assert conv
    o2 = P.to_thermal_flux_oracle(case, False, ksp_rtol=1e-13, land=land, t_prev=t_prev)
    assert o2.step_dt(1800.0)
    assert np.abs(g2.soe.GetSoln() - o2.get_soln()).max() <= 1e-10 * np.abs(o2.get_soln()).max()
