"""The constitutive laws the library evaluates on the host for drivers (mppgpu_host_density / _internal_energy_enthalpy /
_sat_relperm, mpp_amd.laws) against the CPU oracle, and the product's th_mms workload (manufactured sources formed with
them, th_mms_problem.F90:901-934, 1262-1452) against the oracle-side builder of the same problem.  No GPU."""
import os

import numpy as np
import pytest

from mpp_amd import constants as C
from mpp_amd import laws as L
from mpp_amd import workloads as W
from oracle import adapters as A
from oracle import th_oracle as TO
from oracle import vsfm_oracle as VO


def test_density_known_answers_of_the_reference():
    """src/tests/test_eos_*_density.F90:19-25 (p = 120000 Pa, T = 300 K), the tests' own absolute tolerances"""
    for dtype, want in ((C.DENSITY_CONSTANT, (55.508250191225926, 0.0, 0.0)),
                        (C.DENSITY_TGDPB01, (55.317560635066179, 2.4884914247886521e-8, -1.5203176216371761e-2)),
                        (C.DENSITY_IFC67, (55.323696656461536, 2.4854904480147891e-8, -1.5298638598102345e-2))):
        den, ddp, ddt = L.density([120000.0], [300.0], dtype)
        assert abs(den[0] - want[0]) <= 1e-11 and abs(ddp[0] - want[1]) <= 1e-16 and abs(ddt[0] - want[2]) <= 1e-15


@pytest.mark.parametrize("dtype", [C.DENSITY_CONSTANT, C.DENSITY_TGDPB01, C.DENSITY_IFC67])
def test_density_and_enthalpy_follow_the_oracle(dtype):
    rng = np.random.default_rng(4)
    p = rng.uniform(2e4, 4e5, 64)
    t = rng.uniform(275.0, 320.0, 64)
    den, ddp, ddt = L.density(p, t, dtype)
    for i in range(p.size):
        o = VO.density(p[i], t[i], dtype)
        assert den[i] == pytest.approx(o[0], rel=1e-14) and ddp[i] == pytest.approx(o[1], rel=1e-12, abs=1e-30)
        assert ddt[i] == pytest.approx(o[2], rel=1e-12, abs=1e-30)
    for ie in (1, 2):
        out = L.internal_energy_enthalpy(p, t, ie, den * C.FMWH2O, ddt * C.FMWH2O, ddp * C.FMWH2O)
        for i in range(p.size):
            o = TO.internal_energy_enthalpy(p[i], t[i], ie, den[i] * C.FMWH2O, ddt[i] * C.FMWH2O, ddp[i] * C.FMWH2O)
            for a, b in zip(out, o):
                assert a[i] == pytest.approx(b, rel=1e-12, abs=1e-12)


@pytest.mark.parametrize("sf", [C.SAT_FUNC_VAN_GENUCHTEN, C.SAT_FUNC_BROOKS_COREY, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2,
                                C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3])
@pytest.mark.parametrize("lam", [0.5, 0.37])
def test_saturation_functions_follow_the_oracle(sf, lam):
    """the library evaluates some of the reference's real powers as quotients of values at hand (physics.hpp: Brooks-Corey kr,
    the van Genuchten derivative terms): within a few ulp of the oracle's literal restatement"""
    p = np.concatenate([np.linspace(2e3, 1.3e5, 97), [101325.0, 101324.9, 97725.0]])
    sat, dsat, kr, dkr = L.sat_relperm(p, sf, 0.21, 1.0 / 4000.0, lam)
    for i in range(p.size):
        o = VO.auxvar_compute(sf, 0.21, 1.0 / 4000.0, lam, 0.3, C.DENSITY_CONSTANT, p[i])
        assert sat[i] == pytest.approx(o["sat"], rel=1e-13) and dsat[i] == pytest.approx(o["dsat_dP"], rel=1e-12, abs=1e-300)
        assert kr[i] == pytest.approx(o["kr"], rel=1e-13) and dkr[i] == pytest.approx(o["dkr_dP"], rel=1e-12, abs=1e-300)


def test_unknown_ids_are_errors():
    with pytest.raises(L.LawError, match="density type"):
        L.density([1e5], [300.0], 9)
    with pytest.raises(L.LawError, match="saturation function"):
        L.sat_relperm([1e5], 77, 0.2, 1e-4, 0.5)


def test_th_mms_workload_equals_the_oracle_side_builder():
    """mass sources to rounding; heat sources to 1e-11: the driver's 1e-6 central difference of the IFC-67 enthalpy amplifies
    last-bit differences a million times, which is why the host entry points evaluate the formulation's real powers through the
    math library, as the reference's compiled code does (th_physics.hpp rpow<true>; with the kernels' multiplication chains the
    two builders differ by 8e-10 and the th_mms temperature leaves the reference's 1e-8 K tolerance)"""
    a, b = W.th_mms(), A.th_mms()
    assert np.array_equal(a.press_ic, b.press_ic) and np.array_equal(a.temp_ic, b.temp_ic) and np.array_equal(a.perm, b.perm)
    for (ca, sa, va, oa), (cb, sb, vb, ob) in zip(a.mass_bcs + a.energy_bcs, b.mass_bcs + b.energy_bcs):
        assert ca == cb and np.array_equal(va, vb) and np.array_equal(oa, ob) and np.array_equal(sa.id_dn, sb.id_dn)
    ma, mb = a.mass_sss[0][2], b.mass_sss[0][2]
    ha, hb = a.energy_sss[0][2], b.energy_sss[0][2]
    assert np.abs(ma - mb).max() <= 1e-14 * np.abs(mb).max()
    assert np.abs(ha - hb).max() <= 1e-11 * np.abs(hb).max()


def test_shared_ifc67_evaluation_of_the_coupled_kernels_matches_the_term_by_term_functions():
    """th_physics.hpp's ifc67_temperature_terms / ifc67_at_pressure (one temperature, two pressures, reciprocals formed once: what
    k_th_cols evaluates per cell) against density_ifc67_pt<true> / enthalpy_ifc67<true> of the same header (the reference's formulas
    with libm's pow, what the oracle restates), 0.05-100 degC x 100 Pa-20 MPa, on the host (tests/th_eos_check.cpp).  Differences
    are rounding-level against the size of the terms; the enthalpy and dH/dT pass through cancellations near 0 degC and 4 degC"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = os.path.join(root, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe, src = os.path.join(out, "th_eos_check"), os.path.join(root, "tests", "th_eos_check.cpp")
    hdrs = [os.path.join(root, "mpp_amd", "csrc", h) for h in ("th_physics.hpp", "physics.hpp", "fastpow.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(map(os.path.getmtime, [src] + hdrs)):
        subprocess.check_call([hipcc, "-O2", "-o", exe, src], stderr=subprocess.DEVNULL)
    vals = subprocess.check_output([exe], text=True).split()
    n, (d, dp, dt, h, hp, ht) = int(vals[0]), map(float, vals[1:])
    assert n > 100_000
    assert d < 1e-14 and dp < 1e-13 and dt < 1e-12 and hp < 1e-13      # measured 6e-16, 1.5e-15, 3.5e-14, 1.2e-15
    assert h < 1e-8 and ht < 1e-10                                     # measured 1.3e-10 (of |H| + 1e4 J/kmol, H -> 0 at 0 degC), 2.3e-12


def test_host_side_helpers_of_the_setup_passes():
    """parallel running sums against a serial loop (sizes around the chunking), the overflow error, the task pool and the
    no-initialisation vector type of mpp_amd/csrc/runtime.{hpp,cpp} -- host code, no GPU needed (tests/host_setup_check.cpp, linked
    with the library's runtime.o)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    import __graft_entry__ as G
    G.build()                                       # (runtime.o)
    out = os.path.join(root, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe, src = os.path.join(out, "host_setup_check"), os.path.join(root, "tests", "host_setup_check.cpp")
    obj = os.path.join(root, "mpp_amd", "_build", "runtime.o")
    deps = [src, obj, os.path.join(root, "mpp_amd", "csrc", "runtime.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(map(os.path.getmtime, deps)):
        subprocess.check_call([hipcc, "-O2", "-std=c++17", "-c", src, "-o", exe + ".o"], stderr=subprocess.DEVNULL)
        subprocess.check_call([hipcc, "-o", exe, exe + ".o", obj, "-ldl", "-lpthread"], stderr=subprocess.DEVNULL)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.startswith("ok "), res.stdout + res.stderr
