"""Adapters from the workload definitions (mpp_amd.workloads) to the CPU oracle.  TEST INFRASTRUCTURE: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / answer-check legs -- never by the product package."""
from __future__ import annotations

import numpy as np

from mpp_amd import constants as C
from mpp_amd import mesh as M
from mpp_amd.workloads import Problem, THProblem, ThermalProblem


def to_oracle(prob: Problem, with_blocks=True):
    from oracle import vsfm_oracle as VO
    o = VO.OracleProblem(prob.mesh.ncells_local)
    o.set_mesh(prob.mesh.vol, prob.mesh.is_active)
    o.set_connections(prob.mesh.internal_connections())
    for cond_type, cs, _vals in prob.bcs:
        o.add_condition(C.COND_BC, cond_type, cs)
    for cond_type, cs, _vals in prob.sss:
        o.add_condition(C.COND_SS, cond_type, cs)
    o.set_soil_permeability(prob.perm[:, 0], prob.perm[:, 1], prob.perm[:, 2])
    if prob.por_slope is not None:
        o.set_soil_porosity_linear(prob.por, prob.por_pressure_reference, prob.por_slope)
    else:
        o.set_soil_porosity(prob.por)
    o.set_saturation_function(prob.satfunc_type, prob.alpha, prob.lam, prob.sat_res)
    o.set_density_type(prob.density_type)
    for i, (_t, _cs, vals) in enumerate(prob.bcs):
        o.set_condition_values(C.COND_BC, i + 1, vals)
    for i, (_t, _cs, vals) in enumerate(prob.sss):
        o.set_condition_values(C.COND_SS, i + 1, vals)
    if prob.lateral is not None:
        o.set_lateral_connections(prob.lateral)
    if prob.sink_pressure is not None:
        o.set_source_sink_auxvar(C.VAR_POT_MASS_SINK_PRESSURE, prob.sink_pressure)
        o.set_source_sink_auxvar(C.VAR_POT_MASS_SINK_EXPONENT, prob.sink_exponent)
    o.restart(prob.press_ic)
    if with_blocks:
        ptr, cells = M.column_blocks(prob.mesh)
        o.set_blocks(ptr, cells)
    return o


def to_thermal_oracle(tp: ThermalProblem, ksp_rtol=0.0):
    from oracle import thermal_oracle as TO
    n = tp.mesh.ncells_local
    o = TO.ThermalOracle(n)
    o.set_mesh(tp.mesh.vol, tp.mesh.dz)
    o.set_connections(tp.mesh.internal_connections())
    o.add_condition(TO.COND_BC, TO.COND_DIRICHLET, tp.bc)
    ids = np.arange(1, n + 1)
    z = np.zeros(n)
    o.add_condition(TO.COND_SS, TO.COND_HEAT_RATE, M.ConnectionSet(np.zeros(n, dtype=np.int64), ids, z, z, z, np.zeros((n, 3))))
    o.set_soils(TO.SOIL, 1, 0.1, 0.0, tp.tkdry, 0.0)
    o.set_options(cnfac=tp.cnfac, ksp_type=C.KSP_GMRES, pc_type=C.PC_ILU0, ksp_rtol=ksp_rtol)
    o.set_soln_prev_clm(np.full(n, tp.t_init))
    o.set_cell_data(TO.VAR_TUNING_FACTOR, 1.0)
    o.set_cell_data(TO.VAR_LIQ_AREAL_DEN, 0.0)
    o.set_condition_values(TO.COND_BC, 1, tp.bc_temperature)
    o.set_condition_aux(1, "active", np.ones(tp.bc.num_connections))   # thermal_mms_problem.F90:664-668
    o.set_condition_aux(1, "frac", np.ones(tp.bc.num_connections))
    o.set_condition_values(TO.COND_SS, 1, tp.heat_source)
    o.pre_step_dt()
    return o


def to_th_oracle(tp: THProblem, ksp_rtol=0.0):
    from oracle import th_oracle as TO
    o = TO.THOracle(tp.mesh.ncells_local)
    ms = o.mass
    ms.set_mesh(tp.mesh.vol, tp.mesh.is_active)
    ms.set_connections(tp.mesh.intrn_conn_sets[0])
    for bc in tp.mass_bcs:
        ms.add_condition(C.COND_BC, bc[0], bc[1])
    for ss in tp.mass_sss:
        ms.add_condition(C.COND_SS, ss[0], ss[1])
    for bc in tp.energy_bcs:
        o.energy_add_condition(C.COND_BC, bc[0], bc[1])
    for ss in tp.energy_sss:
        o.energy_add_condition(C.COND_SS, ss[0], ss[1])
    ms.set_soil_permeability(tp.perm[:, 0], tp.perm[:, 1], tp.perm[:, 2])
    linear = getattr(tp, "por_slope", None) is not None
    if linear:
        ms.set_soil_porosity_linear(tp.por, tp.por_pressure_reference, tp.por_slope)
    else:
        ms.set_soil_porosity(tp.por)
    ms.set_saturation_function(tp.satfunc_type, tp.alpha, tp.lam, tp.sat_res)
    o.set_eos(tp.density_type, tp.ie_type)
    o.energy_set_soils(tp.por, tp.satfunc_type, tp.alpha, tp.lam, tp.sat_res, tp.heat_cap, tp.tcond_wet, tp.tcond_dry,
                       tp.talpha, tp.den_soil, perm=tp.energy_perm)
    if linear:
        o.energy_set_porosity_linear(tp.por_pressure_reference, tp.por_slope)
    o.setup()
    o.restart(tp.press_ic, tp.temp_ic)
    if ksp_rtol:
        o.glob.set_solver(C.KSP_GMRES, C.PC_ILU0, ksp_rtol=ksp_rtol)
    th_set_conditions(tp, o)
    return o


def th_set_conditions(tp: THProblem, o):
    """what the reference drivers do before every StepDT (set_bondary_conditions / set_source_sink_conditions)"""
    for i, bc in enumerate(tp.mass_bcs):
        o.mass.set_condition_values(C.COND_BC, i + 1, bc[2])
        if len(bc) > 3 and bc[3] is not None:
            o.mass_set_bc_temperature(i + 1, bc[3])
    for i, ss in enumerate(tp.mass_sss):
        o.mass.set_condition_values(C.COND_SS, i + 1, ss[2])
    for i, bc in enumerate(tp.energy_bcs):
        o.energy_set_condition_values(C.COND_BC, i + 1, bc[2])
        o.energy_set_bc_pressure(i + 1, bc[3])
    for i, ss in enumerate(tp.energy_sss):
        o.energy_set_condition_values(C.COND_SS, i + 1, ss[2])


def th_mms(nx=20) -> THProblem:
    """src/driver/standalone/thermal-e/th_mms_problem.F90 (STEADY_STATE_SOIL_ONLY_1D): manufactured steady pressure
    and temperature on x in [0,10], zero porosity and heat capacity (no storage), constant density + IFC67 enthalpy,
    Dirichlet P and T at both ends of both equations, manufactured mass and heat sources in every cell, one step."""
    from oracle import th_oracle as TO
    from oracle import vsfm_oracle as VO
    x_min, x_max = 0.0, 10.0
    xlim = x_max - x_min
    dx = (x_max - x_min) / nx
    m = M.mesh_create("Soil mesh", x_max, 1.0, 1.0, nx, 1, 1, C.CONN_IN_XYZ_DIR, x_min=x_min)
    n = m.ncells_local
    xc = m.x
    pi = 4 * np.arctan(1.0)
    FMW = 18.01534
    density_type, ie_type = C.DENSITY_CONSTANT, 2     # set_problem :960-962 (the second assignment wins)

    def pres(x):      # compute_pressure_or_deriv :991-1021
        a0, a1 = 15000.0, -20000.0
        s, c = np.sin((x - x_min) / xlim * pi), np.cos((x - x_min) / xlim * pi)
        return a0 * s + a1 + 101325.0, a0 * pi / xlim * c, -a0 * pi * pi / xlim / xlim * s

    def temp(x):      # compute_temperature_or_deriv :1023-1044 (the second set of assignments wins)
        a0, a1 = 5.0, 290.0
        s, c = np.sin((x - x_min) / xlim * pi), np.cos((x - x_min) / xlim * pi)
        return a0 * s * 1.0 + a1, a0 * pi / xlim * c * 1.0, -a0 * pi * pi / xlim / xlim * s * 1.0

    def permf(x):     # compute_permeability_or_deriv :1046-1060
        p0 = 1e-11
        return p0 * (2.0 - np.cos((x - x_min) / xlim * pi)), p0 * pi / xlim * np.sin((x - x_min) / xlim * pi)

    alpha0, lam0, sr0, kdry0, kwet0, talpha0 = 1.0 / 4000.0, 0.5, 0.0, 0.25, 1.3, 0.45
    pert = 1e-6
    mu = 8.904156e-4

    def rho_of(P, T):
        den, ddp, ddt = VO.density(P, T, density_type)
        return den * FMW, ddp * FMW, ddt * FMW

    def sat_of(P):
        a = VO.auxvar_compute(C.SAT_FUNC_VAN_GENUCHTEN, sr0, alpha0, lam0, 0.0, density_type, P)
        return a["sat"], a["dsat_dP"], a["kr"], a["dkr_dP"]

    mass_src, heat_src = np.zeros(n), np.zeros(n)
    for i in range(n):
        x = xc[i]
        k, dk_dx = permf(x)
        P, dP_dx, d2P_dx2 = pres(x)
        T, dT_dx, d2T_dx2 = temp(x)
        rho, drho_dP, drho_dT = rho_of(P, T)
        se, dse_dP, kr, dkr_dP = sat_of(P)
        dkr_dx = dkr_dP * dP_dx
        Pp, _, _ = pres(x + pert); Tp, _, _ = temp(x + pert)
        Pn, _, _ = pres(x - pert); Tn, _, _ = temp(x - pert)
        rho_p, drho_p_dP, drho_p_dT = rho_of(Pp, Tp)
        rho_n, drho_n_dP, drho_n_dT = rho_of(Pn, Tn)
        drho_dx = (rho_p - rho_n) / pert / 2.0
        drhoq_dx = -((k * kr / mu) * drho_dx + (rho * kr / mu) * dk_dx + (rho * k / mu) * dkr_dx) * dP_dx - (rho * k * kr / mu) * d2P_dx2
        mass_src[i] = drhoq_dx * dx                                        # DATA_MASS_SOURCE :1262-1316
        U, H, dU_dT, dH_dT, dU_dP, dH_dP = TO.internal_energy_enthalpy(P, T, ie_type, rho, drho_dT, drho_dP)
        rhoq = -rho * (k * kr / mu * dP_dx)
        se_p = sat_of(Pp)[0]; se_n = sat_of(Pn)[0]
        Ke = (se + 1e-6) ** talpha0
        dKe_dx = ((se_p + 1e-6) ** talpha0 - (se_n + 1e-6) ** talpha0) / pert / 2.0
        kappa = kwet0 * Ke + kdry0 * (1.0 - Ke)
        dkappa_dx = 0.0 * Ke + 0.0 * (1.0 - Ke) + (kwet0 - kdry0) * dKe_dx
        H_p = TO.internal_energy_enthalpy(Pp, Tp, ie_type, rho_p, drho_p_dT, drho_p_dP)[1]
        H_n = TO.internal_energy_enthalpy(Pn, Tn, ie_type, rho_n, drho_n_dT, drho_n_dP)[1]
        dH_dx = (H_p - H_n) / pert / 2.0
        heat_src[i] = -(drhoq_dx * H / FMW + rhoq * dH_dx / FMW - dkappa_dx * dT_dx - kappa * d2T_dx2) * dx   # :1365-1452

    P_all = np.array([pres(x)[0] for x in xc]); T_all = np.array([temp(x)[0] for x in xc])
    P0 = 0.0; T0 = 0.0
    for i in range(n):      # DATA_INITIAL_PRESSURE / _TEMPERATURE: running mean, same summation order
        P0 = P0 + 1.0 / nx * P_all[i]; T0 = T0 + 1.0 / nx * T_all[i]
    pres_bc = np.array([pres(xc[0] - dx / 2.0)[0], pres(xc[-1] + dx / 2.0)[0]])
    temp_bc = np.array([temp(xc[0] - dx / 2.0)[0], temp(xc[-1] + dx / 2.0)[0]])
    ends = M.ConnectionSet(np.array([0, 0]), np.array([1, nx]), np.zeros(2), np.full(2, dx / 2.0), np.full(2, 1.0 * 1.0),
                           np.array([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]]))      # ComputeBoundaryDomainConnection, ny = nz = 1
    ids = np.arange(1, n + 1)
    z = np.zeros(n)
    allcells = M.ConnectionSet(np.zeros(n, dtype=np.int64), ids, z, z, np.ones(n), np.zeros((n, 3)))
    perm = np.array([permf(x)[0] for x in xc])
    return THProblem(
        name="th_mms", mesh=m, perm=np.stack([perm] * 3, axis=1), por=np.zeros(n),
        satfunc_type=np.full(n, C.SAT_FUNC_VAN_GENUCHTEN, dtype=np.int32), alpha=np.full(n, alpha0), lam=np.full(n, lam0),
        sat_res=np.full(n, sr0), density_type=density_type, ie_type=ie_type, heat_cap=np.zeros(n),
        tcond_wet=np.full(n, kwet0), tcond_dry=np.full(n, kdry0), talpha=np.full(n, talpha0), den_soil=np.zeros(n),
        energy_perm=(perm, perm, perm), press_ic=np.full(n, P0), temp_ic=np.full(n, T0),
        mass_bcs=[(C.COND_DIRICHLET, ends, pres_bc, temp_bc)], mass_sss=[(C.COND_MASS_RATE, allcells, mass_src)],
        energy_bcs=[(C.COND_DIRICHLET, ends, temp_bc, pres_bc)], energy_sss=[(C.COND_HEAT_RATE, allcells, heat_src)],
        dt=1.0, nstep=1)


def vsfm_mms(nx=20, fully_saturated=False) -> Problem:
    """src/driver/standalone/vsfm/vsfm_mms_problem.F90 + vsfm_mms_steady_state_soil_only_1D.F90 (STEADY_STATE_SOIL_ONLY_1D), built
    with the ORACLE's constitutive laws (the product's builder, mpp_amd.workloads.vsfm_mms, uses the library's host laws: the two
    are compared input by input in tests/test_oracle_vsfm_mms.py): steady 1-D horizontal column x in [0, 10], porosity 0,
    P(x) = 15000 sin(pi x / 10) -+ 20000 + 101325 (:52-60), k(x) = 1e-11 (2 + cos(pi x / 10)) (:74-75), van Genuchten
    alpha = 1/4000, m = 1/2, sr = 1/2 (:89-121), TGDPB01, Dirichlet ends, mass source -d/dx(rho k kr / mu dP/dx) dx (:263-302)."""
    from oracle import vsfm_oracle as VO
    pi = 4.0 * np.arctan(1.0)
    x_min, x_max = 0.0, 10.0
    xlim = x_max - x_min
    dx = (x_max - x_min) / nx
    m = M.mesh_create("Soil mesh", x_max, 1.0, 1.0, nx, 1, 1, C.CONN_IN_XYZ_DIR, x_min=x_min)
    n = m.ncells_local
    xc = m.x
    FMW = 18.01534
    a0, a1 = 15000.0, -20000.0
    alpha0, lam0, sr0 = 1.0 / 4000.0, 0.5, 0.5
    mu = 8.904156e-4

    def pres(x):
        s, c = np.sin((x - x_min) / xlim * pi), np.cos((x - x_min) / xlim * pi)
        return a0 * s + (-a1 if fully_saturated else a1) + 101325.0, a0 * pi / xlim * c, -a0 * pi * pi / xlim / xlim * s

    def permf(x):
        p0 = 1e-11
        return p0 * (2.0 + np.cos((x - x_min) / xlim * pi)), p0 * pi / xlim * (-np.sin((x - x_min) / xlim * pi))

    src, P_all, perm = np.zeros(n), np.zeros(n), np.zeros(n)
    for i in range(n):
        x = xc[i]
        k, dk_dx = permf(x)
        P, dP_dx, d2P_dx2 = pres(x)
        den, ddp, _ddt = VO.density(P, 298.15, C.DENSITY_TGDPB01)
        rho, drho_dP = den * FMW, ddp * FMW
        a = VO.auxvar_compute(C.SAT_FUNC_VAN_GENUCHTEN, sr0, alpha0, lam0, 0.0, C.DENSITY_TGDPB01, P)
        kr, dkr_dP = a["kr"], a["dkr_dP"]
        dkr_dx = dkr_dP * dP_dx
        drho_dx = drho_dP * dP_dx
        src[i] = (-((k * kr / mu) * drho_dx + (rho * kr / mu) * dk_dx + (rho * k / mu) * dkr_dx) * dP_dx - (rho * k * kr / mu) * d2P_dx2) * dx
        P_all[i], perm[i] = P, k
    P0 = 0.0
    for i in range(n):
        P0 = P0 + 1.0 / nx * P_all[i]
    pres_bc = np.array([pres(xc[0] - dx / 2.0)[0], pres(xc[-1] + dx / 2.0)[0]])
    ends = M.ConnectionSet(np.array([0, 0]), np.array([1, nx]), np.zeros(2), np.full(2, dx / 2.0), np.full(2, 1.0 * 1.0),
                           np.array([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]]))
    z = np.zeros(n)
    allcells = M.ConnectionSet(np.full(n, -1), np.arange(1, n + 1), z, z, np.ones(n), np.zeros((n, 3)))
    prob = Problem(
        name="vsfm_mms", mesh=m, perm=np.stack([perm] * 3, axis=1), por=np.zeros(n),
        satfunc_type=np.full(n, C.SAT_FUNC_VAN_GENUCHTEN, dtype=np.int32), alpha=np.full(n, alpha0), lam=np.full(n, lam0),
        sat_res=np.full(n, sr0), density_type=C.DENSITY_TGDPB01, press_ic=np.full(n, P0),
        bcs=[(C.COND_DIRICHLET, ends, pres_bc)], sss=[(C.COND_MASS_RATE, allcells, src)], dt=1.0, nstep=1)
    prob.exact = P_all
    return prob
