"""Randomised differential test of the VSFM path: every seed draws a grid, soils, constitutive laws, boundary / source
conditions, inactive cells and a solver route, then compares the GPU library with the CPU oracle -- operators at a random
state first, then two time steps (convergence reason, Newton count, pressures).  The single-feature tests live in
test_gpu_parity.py / test_gpu_r02.py; this one is after the interactions between them (complete-row march with inline
boundary fluxes x seepage x sinks x soil classes x fused Krylov kernels ...)."""
import numpy as np
import pytest

import problems as P
from mpp_amd import constants as C
from mpp_amd import mesh as M
from test_gpu_parity import _operator_parity

pytestmark = pytest.mark.gpu

SATF = [C.SAT_FUNC_VAN_GENUCHTEN, C.SAT_FUNC_BROOKS_COREY, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2, C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3]
ROUTES = [{}, {"assembly": "general"}, {"assembly": "columns", "nseg": 1}, {"assembly": "columns", "nseg": 3},
          {"assembly": "columns", "nseg": 1, "fused_rows": 0}, {"assembly": "columns", "nseg": 1, "bc_inline": 0},
          {"assembly": "columns", "nseg": 1, "spmv_variant": 20, "tri_lds": 1},
          {"assembly": "columns", "nseg": 1, "spmv_variant": 20, "tri_lds": 1, "fused_first": 0, "fused_waxpy": 0},
          {"assembly": "columns", "nseg": 1, "soil_classes": 0, "level_geometry": 0}, {"ls_spec": 0, "host_poll": 0}]
SOLVERS = [("bcgs", "bjacobi", C.KSP_BCGS, C.PC_BJACOBI), ("gmres", "bjacobi", C.KSP_GMRES, C.PC_BJACOBI),
           ("preonly", "bjacobi", C.KSP_PREONLY, C.PC_BJACOBI), ("bcgs", "ilu", C.KSP_BCGS, C.PC_ILU0)]


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    ncols = int(rng.choice([1, 7, 70, 130, 257]))
    nz = int(rng.choice([2, 3, 11, 24, 40]))
    prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=np.exp(0.5 * rng.standard_normal(ncols)))
    n = prob.mesh.ncells_local
    what = {"ncols": ncols, "nz": nz}
    # soils: per-cell noise, or a handful of classes (the soil-class table of the march), or the benchmark's uniform soil
    soil = what["soil"] = str(rng.choice(["percell", "classes", "uniform"]))
    if soil == "percell":
        f = lambda lo, hi: rng.uniform(lo, hi, n)
    elif soil == "classes":
        k = rng.integers(0, 5, n)
        f = lambda lo, hi: rng.uniform(lo, hi, 5)[k]
    else:
        f = lambda lo, hi: np.ones(n)
    prob.por = prob.por * f(0.8, 1.2)
    prob.alpha = prob.alpha * f(0.8, 1.2)
    prob.sat_res = prob.sat_res * f(0.8, 1.2)
    law = what["law"] = str(rng.choice(["vg_half", "vg", "bc", "sbc2", "sbc3", "mixed"]))
    if law == "vg_half":
        prob.lam = np.full(n, 0.5)
    elif law == "vg":
        prob.lam = np.clip(prob.lam * f(0.7, 1.3), 0.05, 0.95)
    elif law == "mixed":
        prob.satfunc_type = np.asarray(SATF, dtype=np.int32)[rng.integers(0, 4, n)]
        prob.lam = np.where(prob.satfunc_type == C.SAT_FUNC_VAN_GENUCHTEN, rng.uniform(0.3, 0.7, n), rng.uniform(0.4, 1.6, n))
    else:
        prob.satfunc_type = np.full(n, {"bc": SATF[1], "sbc2": SATF[2], "sbc3": SATF[3]}[law], dtype=np.int32)
        prob.lam = 0.4 + 1.2 * (f(0.8, 1.2) - 0.8) / 0.4 if soil != "uniform" else np.full(n, 0.9)
    prob.density_type = what["density"] = int(rng.choice([C.DENSITY_CONSTANT, C.DENSITY_TGDPB01, C.DENSITY_IFC67]))
    if rng.random() < 0.3:
        what["linear_porosity"] = True
        prob.por_pressure_reference = np.full(n, 101325.0) + rng.uniform(-2e4, 2e4, n)
        prob.por_slope = rng.uniform(1e-7, 6e-7, n)
    # boundaries: celia1990 has Dirichlet top and bottom
    top = what["top"] = str(rng.choice(["dirichlet", "seepage", "none"]))
    (tt, tcs, tv), (bt_, bcs_, bv) = prob.bcs           # celia1990: [top, bottom]
    new = []
    if top == "seepage":
        new.append((C.COND_SEEPAGE_BC, tcs, np.full(tv.size, 101325.0)))
    elif top == "dirichlet":
        new.append((tt, tcs, tv * rng.uniform(0.9, 1.1, tv.size)))
    if rng.random() < 0.25:
        what["bottom"] = "none"
    else:
        new.append((bt_, bcs_, bv * rng.uniform(0.9, 1.1, bv.size)))
    prob.bcs = new
    if rng.random() < 0.35:
        what["sources"] = True
        ids = np.arange(1, n + 1); z = np.zeros(n)
        prob.sss = [(C.COND_MASS_RATE, M.ConnectionSet(np.full(n, -1), ids, z, z, np.ones(n), np.zeros((n, 3))), rng.uniform(-2e-7, 2e-7, n))]
    if rng.random() < 0.3 and n > 8:
        what["inactive"] = True
        act = np.ones(n, dtype=np.int32)
        act[rng.choice(n, max(1, n // 50), replace=False)] = 0
        prob.mesh.is_active = act
    route = dict(ROUTES[int(rng.integers(0, len(ROUTES)))])
    ksp, pc, oksp, opc = SOLVERS[int(rng.integers(0, len(SOLVERS)))]
    ls = str(rng.choice(["bt", "bt", "basic"]))
    opts = dict({"ksp_type": ksp, "pc_type": pc, "snes_linesearch_type": ls}, **route)
    what["options"] = opts
    x = prob.press_ic + rng.uniform(0, 1.2e5, n)
    prob.dt = float(rng.choice([600.0, 3600.0]))
    start = what["start"] = str(rng.choice(["ic", "rough", "wet"]))
    if start == "rough":     # a state far from equilibrium: more Newton iterations, backtracking, now and then a cut
        prob.press_ic = prob.press_ic + rng.uniform(0, 3.0e4, n)
    elif start == "wet":     # saturated patches: both branches of every law and of the compressibility switch
        prob.press_ic = np.where(rng.random(n) < 0.3, 101325.0 + rng.uniform(0, 2e4, n), prob.press_ic + rng.uniform(0, 6.0e4, n))
    return prob, x, opts, (oksp, opc, ls), what


@pytest.mark.parametrize("seed", range(96))
def test_random_vsfm_problem_matches_the_oracle(seed):
    prob, x, opts, (oksp, opc, ls), what = draw(seed)
    o, g, _csr, _v = _operator_parity(prob, x, opts)
    o.set_solver(oksp, opc)
    o.set_linesearch(ls)
    for s in range(2):
        assert g.soe.StepDT(prob.dt, s + 1) == tuple(o.step_dt(prob.dt, s + 1)), what
    cg, co = g.soe.counters(), o.counters()
    assert (cg["newton"], cg["time_cuts"]) == (co["newton"], co["time_cuts"]), what
    p_o, p_g = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)
    assert np.abs(p_g - p_o).max() <= 1e-8 * np.abs(p_o).max(), what
