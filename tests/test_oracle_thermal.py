"""Pins the thermal CPU oracle on the reference's regression baselines
regression_tests/thermal/thermal_mms{,_2d,_3d}.regression.baseline (13 printed digits)."""
import os

import numpy as np
import pytest

import problems as P
from conftest import parse_regression_file

CASES = [(1, (20, 1, 1), 5, "thermal_mms"), (2, (3, 3, 1), 2, "thermal_mms_2d"), (3, (3, 3, 3), 3, "thermal_mms_3d")]


@pytest.mark.parametrize("ptype,dims,ncell,name", CASES)
def test_thermal_mms_baselines(golden_dir, ptype, dims, ncell, name):
    gold = parse_regression_file(os.path.join(golden_dir, name + ".regression.baseline"))["temperature"]
    tp = P.thermal_mms(ptype, *dims)
    o = P.to_thermal_oracle(tp)
    assert o.step_dt(tp.dt)
    st = P.regression_stats(o.get_soln(), ncell)
    for key in ("min", "max", "mean"):
        assert st[key] == pytest.approx(gold[key], rel=1e-12)
    for cell, v in st["cells"].items():
        assert v == pytest.approx(gold["cell %d" % cell], rel=1e-12)


def test_thermal_mms_3d_soil_indexing_quirk_is_invisible_on_the_baseline():
    """MPPThermalSetSoils reads the soil arrays z-fastest while the mesh is numbered x-fastest; on the 3x3x3 cube
    with lambda = exp(x+y+z-1) (symmetric in x,y,z) both associations give the same numbers, so the baseline
    cannot distinguish them -- but a non-cubic grid can."""
    a = P.to_thermal_oracle(P.thermal_mms(3, 3, 3, 3, replicate_reference_soil_indexing=True))
    b = P.to_thermal_oracle(P.thermal_mms(3, 3, 3, 3, replicate_reference_soil_indexing=False))
    a.step_dt(1.0); b.step_dt(1.0)
    np.testing.assert_allclose(a.get_soln(), b.get_soln(), rtol=1e-12)
    c = P.to_thermal_oracle(P.thermal_mms(3, 4, 3, 2, replicate_reference_soil_indexing=True))
    d = P.to_thermal_oracle(P.thermal_mms(3, 4, 3, 2, replicate_reference_soil_indexing=False))
    c.step_dt(1.0); d.step_dt(1.0)
    assert np.abs(c.get_soln() - d.get_soln()).max() > 1e-3


def test_thermal_mms_second_order_convergence():
    """discretisation error of the restated operator against the manufactured solution halves twice per refinement"""
    errs = []
    for nx in (10, 20, 40):
        tp = P.thermal_mms(1, nx, 1, 1)
        o = P.to_thermal_oracle(tp, ksp_rtol=1e-12)
        o.step_dt(tp.dt)
        errs.append(np.abs(o.get_soln() - tp.exact).max())
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5


def _flux_rhs_and_operator(case, flux_first, defects, H=None, dhsdT=None, tdir=None):
    tp, diri, tdir0, flux, H0, frac, dhsdT0 = case
    c = (tp, diri, tdir0 if tdir is None else tdir, flux, H0 if H is None else H, frac, dhsdT0 if dhsdT is None else dhsdT)
    o = P.to_thermal_flux_oracle(c, flux_first, defects=defects)
    b = o.compute_rhs(1800.0)
    rp, cj, val = o.compute_operators()
    diag = np.array([val[rp[i]:rp[i + 1]][cj[rp[i]:rp[i + 1]] == i][0] for i in range(len(b))])
    return b, diag


def test_heat_flux_boundary_is_the_linearised_surface_flux():
    """COND_HEAT_FLUX: the right-hand side takes (H - dH/dT T_old) frac area (GetFromSOEAuxVarsBC :339-354, Divergence :909-924),
    the operator's diagonal -frac dH/dT area (ComputeOperatorsDiag :1199-1217 as intended) -- CLM's H + dH/dT (T_new - T_old)."""
    case = P.thermal_flux_case()
    tp, diri, tdir, flux, H, frac, dhsdT = case
    cells = np.asarray(flux.id_dn) - 1
    b, d = _flux_rhs_and_operator(case, False, False)
    b0, d0 = _flux_rhs_and_operator(case, False, False, H=0 * H, dhsdT=0 * dhsdT)
    want_b = np.zeros_like(b); want_d = np.zeros_like(b)
    np.add.at(want_b, cells, (H - dhsdT * tp.exact[cells]) * frac * flux.area)
    np.add.at(want_d, cells, -frac * dhsdT * flux.area)
    assert np.abs((b - b0) - want_b).max() <= 1e-12 * np.abs(b).max()
    assert np.abs((d - d0) - want_d).max() <= 1e-12 * np.abs(d).max()
    assert np.abs(want_b).max() > 0 and np.abs(want_d).max() > 0


@pytest.mark.parametrize("flux_first", [False, True])
def test_reference_defects_switch_of_the_thermal_operator(flux_first):
    """With the switch the COND_DIRICHLET right-hand side takes the routine's stale `area` (:906-908) -- the last internal
    connection's, or the last COND_HEAT_FLUX connection's when that condition comes first -- and the heat-flux operator term is
    dhsdT**area (:1215).  Off (the default) neither happens."""
    case = P.thermal_flux_case()
    tp, diri, tdir, flux, H, frac, dhsdT = case
    ic = tp.mesh.internal_connections()
    stale = flux.area[-1] if flux_first else ic.area[-1]
    assert not np.isclose(flux.area[-1], ic.area[-1])
    b, d = _flux_rhs_and_operator(case, flux_first, False)
    bd, dd = _flux_rhs_and_operator(case, flux_first, True)
    # Dirichlet part of the right-hand side: linear in the boundary temperature (dry soil: no T in the conductivity)
    D = b - _flux_rhs_and_operator(case, flux_first, False, tdir=0 * tdir)[0]
    Dd = bd - _flux_rhs_and_operator(case, flux_first, True, tdir=0 * tdir)[0]
    cells = np.asarray(diri.id_dn) - 1
    once = np.bincount(cells, minlength=len(b)) == 1           # cells with exactly one Dirichlet face
    area_of = np.zeros(len(b)); area_of[cells] = diri.area
    sel = once & (D != 0)
    assert sel.sum() > 50
    np.testing.assert_allclose(Dd[sel] / D[sel], stale / area_of[sel], rtol=1e-11)
    assert np.abs(Dd[sel] / D[sel] - 1.0).max() > 0.1                # ... and it matters on this mesh
    # operator: -frac dhsdT**area instead of -frac dhsdT area on the diagonal of the heat-flux cells
    fc = np.asarray(flux.id_dn) - 1
    want = np.zeros_like(d)
    np.add.at(want, fc, -frac * dhsdT ** flux.area + frac * dhsdT * flux.area)
    assert np.abs((dd - d) - want).max() <= 1e-12 * np.abs(d).max()
    assert np.abs(want).max() > 1e-3


def test_boundary_auxvars_are_never_shallow_soil():
    """MPPThermalSetSoils copies itype, por and the soil constants to the boundary auxvars but not is_soil_shallow
    (MultiPhysicsProbThermal.F90:186-204), which ThermKSPTempSoilAuxVarInit left FALSE: a soil cell's Dirichlet auxvar conducts
    like bedrock (3 W/m/K).  With dist_up = 0 (the MMS baselines) the harmonic mean does not see it; with dist_up > 0 it does."""
    case = P.thermal_flux_case(bc_dist_up=0.4)
    tp, diri, tdir, flux, H, frac, dhsdT = case
    b = _flux_rhs_and_operator(case, False, False)[0]
    b0 = _flux_rhs_and_operator(case, False, False, tdir=0 * tdir)[0]
    cells = np.asarray(diri.id_dn) - 1
    kc = tp.tkdry[cells]                                      # dry shallow soil: the cell's conductivity
    du, dd = np.asarray(diri.dist_up), np.asarray(diri.dist_dn)
    kav = 3.0 * kc * (du + dd) / (3.0 * dd + kc * du)
    want = np.zeros_like(b)
    np.add.at(want, cells, kav / (du + dd) * tdir * diri.area)
    assert np.abs((b - b0) - want).max() <= 1e-12 * np.abs(want).max()


def test_boundary_auxvars_start_inactive_with_zero_fraction():
    """ThermKSPTempBaseAuxVarInit (ThermalKSPTemperatureBaseAuxType.F90:56-60): is_active FALSE, frac 0 -- a Dirichlet face does
    nothing until the caller activates it (the MMS driver does: thermal_mms_problem.F90:664-668), a heat-flux face enters with
    its fraction"""
    from oracle import thermal_oracle as TO
    tp = P.thermal_mms(3, 4, 3, 5)
    n = tp.mesh.ncells_local

    def build(activate):
        o = TO.ThermalOracle(n)
        o.set_mesh(tp.mesh.vol, tp.mesh.dz)
        o.set_connections(tp.mesh.internal_connections())
        o.add_condition(TO.COND_BC, TO.COND_DIRICHLET, tp.bc)
        o.set_soils(TO.SOIL, 1, 0.1, 0.0, tp.tkdry, 0.0)
        o.set_options(cnfac=0.0)
        o.set_soln_prev_clm(np.full(n, 290.0))
        o.set_cell_data(TO.VAR_TUNING_FACTOR, 1.0)
        o.set_condition_values(TO.COND_BC, 1, tp.bc_temperature)
        if activate:
            o.set_condition_aux(1, "active", np.ones(tp.bc.num_connections))
            o.set_condition_aux(1, "frac", np.ones(tp.bc.num_connections))
        o.pre_step_dt()
        b = o.compute_rhs(1.0)
        rp, cj, val = o.compute_operators()
        return b, val

    b0, v0 = build(False)
    b1, v1 = build(True)
    cells = np.unique(np.asarray(tp.bc.id_dn) - 1)
    # without activation the right-hand side is the accumulation term alone, the operator has no boundary conductance
    o = P.to_thermal_oracle(tp)
    assert np.abs(b1 - b0)[cells].min() > 0 and np.abs(v1 - v0).max() > 0
    others = np.setdiff1d(np.arange(n), cells)
    assert np.array_equal(b0[others], b1[others])
