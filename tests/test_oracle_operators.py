"""Self-consistency of the oracle's operator-level pieces (no reference fixture exists for these:
'parity unpinned by the reference') -- checked against dense algebra and finite differences."""
import numpy as np
import pytest

import problems as P
from mpp_amd import constants as C
from oracle import vsfm_oracle as VO


def _dense(rp, cj, val, n):
    A = np.zeros((n, n))
    for i in range(n):
        for k in range(rp[i], rp[i + 1]):
            A[i, cj[k]] += val[k]
    return A


def _setup(ncols=2, nz=12, seed=0):
    rng = np.random.default_rng(seed)
    prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=np.exp(0.5 * rng.standard_normal(ncols)))
    o = P.to_oracle(prob)
    o.set_dtime(prob.dt)
    o.pre_solve()
    x = prob.press_ic + rng.uniform(0, 9.0e4, prob.mesh.ncells_local)
    return prob, o, x


def test_jacobian_matches_finite_differences():
    prob, o, x = _setup()
    n = x.size
    rp, cj, val = o.jacobian(x)
    J = _dense(rp, cj, val, n)
    f0 = o.residual(x)
    for j in range(n):
        h = 1e-3
        xp = x.copy(); xp[j] += h
        xm = x.copy(); xm[j] -= h
        col = (o.residual(xp) - o.residual(xm)) / (2 * h)
        scale = np.abs(J[:, j]).max() + 1e-30
        assert np.abs(col - J[:, j]).max() / scale < 2e-5, j
    assert np.isfinite(f0).all()


def test_spmv_and_pcs_against_dense():
    prob, o, x = _setup(ncols=3, nz=10, seed=1)
    n = x.size
    rp, cj, val = o.jacobian(x)
    A = _dense(rp, cj, val, n)
    v = np.random.default_rng(2).standard_normal(n)
    np.testing.assert_allclose(VO.spmv(rp, cj, val, v), A @ v, rtol=1e-13, atol=1e-25)
    exact = np.linalg.solve(A, v)
    scale = np.abs(exact).max()
    # vertical-only connectivity: both ILU(0) and column block-Jacobi are exact factorizations
    for pc in (C.PC_ILU0, C.PC_BJACOBI):
        z = o.pc_apply(pc, rp, cj, val, v)
        assert np.abs(z - exact).max() / scale < 1e-9
    for ksp in (C.KSP_GMRES, C.KSP_BCGS):
        xs, its, reason = o.ksp_solve(ksp, C.PC_BJACOBI, rp, cj, val, v)
        assert reason > 0 and its <= 2
        assert np.abs(xs - exact).max() / scale < 1e-8


def test_krylov_on_7pt_pattern():
    prob = P.vchannel(nx=5, ny=4, nz=6)
    o = P.to_oracle(prob)
    o.set_dtime(prob.dt)
    o.pre_solve()
    x = prob.press_ic.copy()
    n = x.size
    rp, cj, val = o.jacobian(x)
    A = _dense(rp, cj, val, n)
    exact = np.random.default_rng(5).standard_normal(n)
    b = A @ exact
    for ksp, pc in ((C.KSP_GMRES, C.PC_ILU0), (C.KSP_BCGS, C.PC_BJACOBI), (C.KSP_BCGS, C.PC_ILU0),
                    (C.KSP_GMRES, C.PC_BJACOBI)):
        xs, its, reason = o.ksp_solve(ksp, pc, rp, cj, val, b, rtol=1e-10)
        assert reason > 0, (ksp, pc, reason)
        assert np.abs(xs - exact).max() / np.abs(exact).max() < 1e-6, (ksp, pc)


def test_seepage_and_swap_order_flux_branches():
    up = [90000.0, 0.3, 1e-5, 55.3, 2.5e-8]
    dn = [120000.0, 1.0, 0.0, 55.32, 2.5e-8]
    perm = [1e-12, 1e-12, 1e-12]
    uv = [0.0, 0.0, -1.0]
    f_dir, a, b = VO.richards_flux(up, perm, dn, perm, 1.0, 0.0, 0.25, uv, False, False, C.COND_DIRICHLET)
    f_swap, a2, b2 = VO.richards_flux(dn, perm, up, perm, 1.0, 0.0, 0.25, uv, False, True, C.COND_DIRICHLET)
    assert f_swap == -f_dir and a2 == -b and b2 == -a
    # seepage: outflow potential (dphi>0) with P_up <= Pref switches the face off
    up_s = [101000.0, 0.9, 1e-5, 55.3, 2.5e-8]
    dn_s = [90000.0, 0.5, 1e-5, 55.3, 2.5e-8]
    f_seep, _, bseep = VO.richards_flux(up_s, perm, dn_s, perm, 1.0, 0.0, 0.25, uv, False, False, C.COND_SEEPAGE_BC)
    assert f_seep == 0.0
    f_mf, a3, b3 = VO.richards_flux(up, perm, dn, perm, 1.0, 0.0, 0.25, uv, False, False, C.COND_MASS_FLUX)
    assert (f_mf, a3, b3) == (0.0, 0.0, 0.0)


@pytest.mark.parametrize("sf", [C.SAT_FUNC_VAN_GENUCHTEN, C.SAT_FUNC_BROOKS_COREY,
                                C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3])
def test_saturation_derivatives(sf):
    alpha, lam, sr = 3.4e-4, 0.5, 0.2
    for p in (3.5e3, 5.0e4, 9.0e4, 9.9e4, 1.0e5, 1.2e5):
        a = VO.auxvar_compute(sf, sr, alpha, lam, 0.4, C.DENSITY_TGDPB01, p)
        h = 0.5
        ap = VO.auxvar_compute(sf, sr, alpha, lam, 0.4, C.DENSITY_TGDPB01, p + h)
        am = VO.auxvar_compute(sf, sr, alpha, lam, 0.4, C.DENSITY_TGDPB01, p - h)
        for q, dq in (("sat", "dsat_dP"), ("kr", "dkr_dP"), ("den", "dden_dP")):
            fd = (ap[q] - am[q]) / (2 * h)
            assert abs(fd - a[dq]) <= 1e-4 * max(abs(a[dq]), 1e-12) + 1e-14, (sf, p, q)
        assert 0.0 < a["sat"] <= 1.0 and 0.0 < a["kr"] <= 1.0
