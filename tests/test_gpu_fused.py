"""GPU parity of the fused passes of the VSFM path: the complete-row column march (boundary
fluxes folded in, ||F|| / ||X|| and the column LU produced by the assembly pass) and the line-search reductions inside the
SpMV.  Every case is checked against the oracle and against the unfused kernels of the same library."""
import numpy as np
import pytest

from mpp_amd import constants as C

import problems as P
from test_gpu_parity import _hetero, _operator_parity

pytestmark = pytest.mark.gpu

ROWS = {"assembly": "columns", "nseg": 1, "pc_type": "bjacobi", "ksp_type": "bcgs"}   # complete rows need nseg == 1


def _two_bcs_on_top(prob):
    """a second condition on the top cells (seepage next to the Dirichlet one): two entries in one side-array slot"""
    t, cs, vals = prob.bcs[0]
    prob.bcs.append((C.COND_SEEPAGE_BC, cs, np.full(vals.size, 101325.0)))
    return prob


@pytest.mark.parametrize("variant", ["plain", "two_bcs", "inactive_sources", "ragged_tile"])
def test_complete_rows_operators(variant):
    prob, x = _hetero(ncols=130 if variant == "ragged_tile" else 70, nz=19, seed=21)
    n = x.size
    if variant == "two_bcs":
        _two_bcs_on_top(prob)
    if variant == "inactive_sources":
        act = np.ones(n, dtype=np.int32)
        act[[0, 18, 19, 40, 700]] = 0          # column ends included: a boundary cell that is inactive
        prob.mesh.is_active = act
        from mpp_amd import mesh as M
        ids = np.arange(1, n + 1)
        z = np.zeros(n)
        ss = M.ConnectionSet(np.full(n, -1), ids, z, z, np.ones(n), np.zeros((n, 3)))
        prob.sss = [(C.COND_MASS_RATE, ss, np.random.default_rng(1).uniform(-1e-6, 1e-6, n))]
    o, g, (rp, cj, val), v = _operator_parity(prob, x, ROWS)
    # the column LU left behind by the Jacobian pass
    z_o = o.pc_apply(C.PC_BJACOBI, rp, cj, val, v)
    z_g = g.soe.PCApply(v)
    assert np.abs(z_g - z_o).max() <= 1e-10 * np.abs(z_o).max()
    # and the unfused kernels of the same library, bit for bit where only one boundary entry sits on a cell
    h = P.to_gpu(prob, dict(ROWS, fused_rows=0))
    h.soe.SetDtime(prob.dt); h.soe.PreSolve()
    f_g, f_h = g.soe.ComputeResidual(x), h.soe.ComputeResidual(x)
    val_g, val_h = g.soe.ComputeJacobian(x)[2], h.soe.ComputeJacobian(x)[2]
    if variant == "two_bcs":
        assert np.abs(f_g - f_h).max() <= 1e-14 * np.abs(f_h).max()
        assert np.abs(val_g - val_h).max() <= 1e-14 * np.abs(val_h).max()
    else:
        assert np.array_equal(f_g, f_h) and np.array_equal(val_g, val_h)
        assert np.array_equal(g.soe.PCApply(v), h.soe.PCApply(v))


def test_boundary_on_a_middle_cell_falls_back():
    """a condition on a cell that is not a column end: the march cannot finish that row, the separate kernels run"""
    prob, x = _hetero(ncols=70, nz=19, seed=4)
    from mpp_amd import mesh as M
    t, cs, vals = prob.bcs[0]
    ids = np.asarray(cs.id_dn).copy()
    ids[3] -= 5                                            # move one top-cell condition five levels down its column
    prob.bcs[0] = (t, M.ConnectionSet(cs.id_up, ids, cs.dist_up, cs.dist_dn, cs.area, cs.unit_vec), vals)
    _operator_parity(prob, x, ROWS)


@pytest.mark.parametrize("ksp,pc", [("bcgs", "bjacobi"), ("gmres", "ilu")])
def test_fused_and_unfused_steps_agree(ksp, pc):
    """three celia steps with every fused pass on and off: same Newton / Krylov counts, solutions equal to rounding,
    and both equal to the oracle"""
    prob = P.celia1990(ncols=70, nz=40, perm_mult=np.exp(0.4 * np.random.default_rng(5).standard_normal(70)))
    base = {"assembly": "columns", "nseg": 1, "ksp_type": ksp, "pc_type": pc, "spmv_variant": 20}
    on = P.to_gpu(prob, base)
    off = P.to_gpu(prob, dict(base, fused_rows=0, fused_ls_dots=0))
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_BCGS if ksp == "bcgs" else C.KSP_GMRES, C.PC_BJACOBI if pc == "bjacobi" else C.PC_ILU0)
    for s in range(3):
        r_on, r_off, r_o = on.soe.StepDT(prob.dt, s + 1), off.soe.StepDT(prob.dt, s + 1), o.step_dt(prob.dt, s + 1)
        assert r_on == r_off == tuple(r_o)
    c_on, c_off, c_o = on.soe.counters(), off.soe.counters(), o.counters()
    assert c_on["newton"] == c_off["newton"] == c_o["newton"]
    assert c_on["linear"] == c_off["linear"]
    p_on = on.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    p_off = off.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    p_o = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.abs(p_on - p_off).max() <= 1e-10 * np.abs(p_off).max()
    assert np.abs(p_on - p_o).max() <= 1e-8 * np.abs(p_o).max()


def test_complete_rows_time_step_cut():
    """dt halving with the fused passes: identical branch sequence to the oracle"""
    prob = P.celia1990(ncols=66, nz=40)
    opts = dict(ROWS, ksp_type="preonly", snes_max_it=6)
    g = P.to_gpu(prob, opts)
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_PREONLY, C.PC_BJACOBI, snes_max_it=6)
    assert g.soe.StepDT(86400.0, 1) == tuple(o.step_dt(86400.0, 1))
    cg, co = g.soe.counters(), o.counters()
    assert cg["time_cuts"] == co["time_cuts"] > 0 and cg["newton"] == co["newton"]
    np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                               o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=1e-8)


def test_more_column_blocks_than_one_reduction_stage_holds():
    """1,048,832 columns = 4097 blocks of 256: the per-block partial sums of the column solves and of the complete-row march
    go through the two-stage finalisation; fused and unfused solves must agree"""
    ncols, nz = 4097 * 256, 4
    prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=np.exp(0.3 * np.random.default_rng(2).standard_normal(ncols)))
    base = {"ksp_type": "bcgs", "pc_type": "bjacobi", "spmv_variant": 20}
    on = P.to_gpu(prob, base)
    off = P.to_gpu(prob, dict(base, fused_rows=0, fused_ls_dots=0))
    for s in range(2):
        assert on.soe.StepDT(prob.dt, s + 1) == off.soe.StepDT(prob.dt, s + 1)
    c_on, c_off = on.soe.counters(), off.soe.counters()
    assert c_on["newton"] == c_off["newton"] and c_on["linear"] == c_off["linear"]
    p_on = on.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    p_off = off.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    assert np.isfinite(p_on).all()
    assert np.abs(p_on - p_off).max() <= 1e-9 * np.abs(p_off).max()


@pytest.mark.parametrize("ncols,nz", [(1, 2), (65, 2), (3, 3), (64, 1)])
def test_complete_rows_smallest_columns(ncols, nz):
    """columns of one to three cells: every row is a boundary row (nz == 1 is left to the general kernels)"""
    prob, x = _hetero(ncols=ncols, nz=nz, seed=31)
    o, g, (rp, cj, val), v = _operator_parity(prob, x, ROWS)
    z_o = o.pc_apply(C.PC_BJACOBI, rp, cj, val, v)
    z_g = g.soe.PCApply(v)
    assert np.abs(z_g - z_o).max() <= 1e-10 * np.abs(z_o).max()
    assert g.soe.StepDT(prob.dt, 1) == tuple(o.step_dt(prob.dt, 1))
    np.testing.assert_allclose(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                               o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=1e-8)


@pytest.mark.parametrize("ncols,nz", [(70, 19), (130, 2), (64, 3), (3, 100), (200, 50)])
@pytest.mark.parametrize("ksp", ["bcgs", "gmres"])
def test_spmv_fused_with_the_column_solve_changes_no_bit(ncols, nz, ksp):
    """M^{-1} A v as ONE kernel on a vertical-only grid (the CSR SpMV fused with the LDS column solve: A v is never written) against
    the CSR SpMV kernel followed by the column solve: the fused kernel sums a row as the SpMV kernels do (products rounded one by
    one, added in entry order), so Krylov iterates, iteration counts and answers must be the same bits -- ragged last tile, two- and
    three-level columns (end rows only / one middle row), tall columns, inactive cells; BiCGStab and GMRES; and the linear solve
    alone on a random right-hand side"""
    prob, x = _hetero(ncols=ncols, nz=nz, seed=31)
    n = x.size
    if nz > 3:
        act = np.ones(n, dtype=np.int32)
        act[[1, nz, 2 * nz + 1]] = 0
        prob.mesh.is_active = act
    base = dict(ROWS, ksp_type=ksp, tri_lds=1, ls_fold=0)   # (ls_fold recombines the search's F . (J Y): rounding, its own test below)
    # all three fused (the default: the solve's first kernel also does the back substitution of the forward-eliminated residual),
    # the two products fused only, nothing fused
    # (also: s = r - alpha v formed inside the kernel of the second product, fused_waxpy)
    # (and: the reductions of those kernels finished by the kernels themselves, fold_finalize -- the last block sums in k_finalize's order)
    gs = [P.to_gpu(prob, dict(base, fused_spmv_pc=f, fused_first=ff, fused_waxpy=fw, fold_finalize=fo))
          for f, ff, fw, fo in ((1, 1, 1, 0), (1, 0, 0, 0), (0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 1), (1, 0, 0, 1))]
    for g in gs:
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        g.soe.ComputeJacobian(x)
    b = np.random.default_rng(4).standard_normal(n)
    sols = [g.soe.KSPSolve(b) for g in gs]
    assert sols[0][1:] == sols[2][1:] and sols[0][1] > 0           # iterations, reason
    assert all(np.array_equal(q[0], sols[2][0]) and q[1:] == sols[2][1:] for q in sols)
    for g in gs:
        for s in range(2):
            conv, reason = g.soe.StepDT(prob.dt, s + 1)
            assert conv
    for g in (gs[0], gs[1], gs[3], gs[4], gs[5]):
        assert g.soe.counters() == gs[2].soe.counters()
        assert np.array_equal(g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                              gs[2].soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1))


def test_a_krylov_solve_that_goes_on_discards_the_speculative_work():
    """behind the first BiCGStab iteration the solver queues what follows a one-iteration solve: the line search's reductions, its
    first trial point (which overwrites the Jacobian), and it skips writing r.  When the solve does go on -- forced here by a
    ksp_rtol the first iteration cannot reach -- all of that is void: the matrix and factors are assembled again, r is formed, the
    function evaluation is not counted.  Same decisions, counters and pressures as the plain sequence of operations"""
    prob, x = _hetero(ncols=70, nz=19, seed=41)
    big = dict(ROWS, spmv_variant=20, tri_lds=1, ls_fold=0)   # the kernels a large grid selects by itself (the look-ahead needs the compact-index SpMV)
    gs = [P.to_gpu(prob, dict(big, **o)) for o in ({}, {"ls_spec": 0, "ls_ahead": 0, "lazy_r": 0, "fused_first": 0, "fused_spmv_pc": 0})]
    out = []
    for g in gs:
        res = [g.soe.StepDT(prob.dt, 1)]           # a normal step: every solve takes one iteration, speculation gets armed
        g.soe.SetOption("ksp_rtol", 1e-300)        # unreachable (the exact column solve leaves ~1e-17 after one iteration)
        g.soe.SetOption("ksp_max_it", 3)           # ... so every linear solve of this step fails and the step is cut until it gives up
        res.append(g.soe.StepDT(prob.dt, 2))
        g.soe.SetOption("ksp_rtol", 1e-5)
        res.append(g.soe.StepDT(prob.dt, 3))
        out.append((res, g.soe.counters(), g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)))
    assert out[0][0] == out[1][0]
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2])
    assert gs[0].soe.assembly_info()["speculative_trials_discarded"] > 0
    assert gs[1].soe.assembly_info()["speculative_trials_discarded"] == 0


@pytest.mark.parametrize("ncols,nz", [(70, 19), (130, 2), (200, 50)])
def test_elimination_factors_rebuilt_on_demand_are_the_ones_the_march_would_have_stored(ncols, nz):
    """Round 4: the fused SpMV + column solve kernels redo l_k = lower_k / pivot_{k-1} from the matrix rows they read, so the march does
    not store l where they will run (lazy_tri_l) and a plain column solve asked for afterwards rebuilds it (k_tri_l).  Against a
    handle whose march stores l: the same bits from a plain PCApply (which triggers the rebuild), from a Krylov solve and after two
    time steps; and again with the rebuild in between two fused solves of the same factorisation"""
    prob, x = _hetero(ncols=ncols, nz=nz, seed=17)
    n = x.size
    base = dict(ROWS, tri_lds=1)
    lazy, stored = P.to_gpu(prob, dict(base, lazy_tri_l=1)), P.to_gpu(prob, dict(base, lazy_tri_l=0))
    for g in (lazy, stored):
        g.soe.SetDtime(prob.dt); g.soe.PreSolve()
        g.soe.ComputeResidual(x)
        g.soe.ComputeJacobian(x)
    rng = np.random.default_rng(8)
    b = rng.standard_normal(n)
    s1 = [g.soe.KSPSolve(b) for g in (lazy, stored)]                 # fused kernels (no l read)
    assert np.array_equal(s1[0][0], s1[1][0]) and s1[0][1:] == s1[1][1:] and s1[0][1] > 0
    v = rng.standard_normal(n)
    assert np.array_equal(lazy.soe.PCApply(v), stored.soe.PCApply(v))   # plain solve: l rebuilt from the rows and the pivots
    s2 = [g.soe.KSPSolve(v) for g in (lazy, stored)]
    assert np.array_equal(s2[0][0], s2[1][0]) and s2[0][1:] == s2[1][1:]
    for g in (lazy, stored):
        for s in range(2):
            assert g.soe.StepDT(prob.dt, s + 1)[0]
    assert lazy.soe.counters() == stored.soe.counters()
    assert np.array_equal(lazy.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1),
                          stored.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1))


@pytest.mark.parametrize("ncols,nz,spec", [(70, 19, 1), (200, 50, 1), (130, 7, 0), (64, 3, 1)])
def test_line_search_sums_out_of_the_krylov_kernels(ncols, nz, spec):
    """Round 5 (option ls_fold): when a BiCGStab solve ends with its first iteration, Y = alpha p + omega s, and the two fused product
    kernels of that iteration held the rows of J p and J s in registers -- they reduce F . (J p) and F . (J s) on the way, the update
    kernel reduces Y . Y, and the line search takes F . (J Y) = alpha F.(J p) + omega F.(J s) from the host instead of launching a
    product pass of its own.  Equal to MatMult + VecDot up to the rounding of the recombination: same Newton / linear / function
    counts as the separate pass, pressures equal to 1e-12, and both as close to the oracle; a solve that goes on (second step:
    ksp_rtol out of reach) falls back to the product pass"""
    prob, x = _hetero(ncols=ncols, nz=nz, seed=53)
    base = dict(ROWS, tri_lds=1, ls_spec=spec)
    on, off = P.to_gpu(prob, dict(base, ls_fold=1)), P.to_gpu(prob, dict(base, ls_fold=0))
    o = P.to_oracle(prob)
    o.set_solver(C.KSP_BCGS, C.PC_BJACOBI)
    for s in range(3):
        r_on, r_off = on.soe.StepDT(prob.dt, s + 1), off.soe.StepDT(prob.dt, s + 1)
        assert r_on == r_off == tuple(o.step_dt(prob.dt, s + 1))
    assert on.soe.counters() == off.soe.counters()
    assert on.soe.counters()["newton"] == o.counters()["newton"]
    folds = on.soe.assembly_info()["line_search_sums_folded"]
    assert folds > 0 and off.soe.assembly_info()["line_search_sums_folded"] == 0
    # every Newton iteration but the first of a run (the look-ahead is armed by a one-iteration solve) took its sums from the solve
    assert folds >= on.soe.counters()["newton"] - 3
    p_on = on.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    p_off = off.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    p_o = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.abs(p_on - p_off).max() <= 1e-12 * np.abs(p_off).max()
    assert np.abs(p_on - p_o).max() <= 1e-9 * np.abs(p_o).max()
    # the product launches of the search are gone where the fold ran (profile: the library counts the launches of a kernel class)
    for g in (on, off):
        g.soe.SetOption("profile", 1); g.soe.reset_stats()
        assert g.soe.StepDT(prob.dt, 4)[0]
    assert on.soe.kernel_stats("spmv")["calls"] < off.soe.kernel_stats("spmv")["calls"]
    # a solve that needs a second iteration: the folded sums are void, the search launches its product
    for g in (on, off):
        g.soe.SetOption("ksp_rtol", 1e-17)
    r = [g.soe.StepDT(prob.dt, 5) for g in (on, off)]
    assert r[0] == r[1]
    assert on.soe.counters() == off.soe.counters()
    assert np.abs(on.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
                  - off.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)).max() <= 1e-12 * np.abs(p_off).max()
