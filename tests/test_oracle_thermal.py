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
