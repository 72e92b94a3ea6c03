"""Problem definitions shared by the parity tests: the workload builders of the product package (mpp_amd.workloads)
plus the adapters that build the same problems on the CPU oracle (oracle/adapters.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mpp_amd.workloads import *  # noqa: E402,F401,F403
from mpp_amd.workloads import _mms_funcs, _restrict_conn_set  # noqa: E402,F401
from oracle.adapters import th_mms, th_set_conditions, to_oracle, to_th_oracle, to_thermal_oracle  # noqa: E402,F401


# ---- thermal problem with a COND_HEAT_FLUX boundary (CLM's linearised surface flux H + dH/dT (T - T_old)) ----
def thermal_flux_case(nx=7, ny=5, nz=6, bc_dist_up=0.0):
    """thermal_mms_3d's mesh and soils; the x-max face carries a heat flux with frac and dhsdT, the other faces stay Dirichlet.
    Returns (tp, dirichlet set, its temperatures, flux set, H, frac, dhsdT)."""
    import numpy as np
    from mpp_amd import mesh as M
    tp = thermal_mms(3, nx, ny, nz, replicate_reference_soil_indexing=False)
    tp.cnfac = 0.5
    uv = np.asarray(tp.bc.unit_vec)
    zc = np.asarray(tp.mesh.x)[np.asarray(tp.bc.id_dn) - 1]
    top = np.isclose(np.asarray(tp.bc.area), 1.0 / (ny * nz)) & (zc > 0.5 * (zc.min() + zc.max()))   # the x-max face
    sub = lambda keep: M.ConnectionSet(np.asarray(tp.bc.id_up)[keep], np.asarray(tp.bc.id_dn)[keep], np.asarray(tp.bc.dist_up)[keep],
                                       np.asarray(tp.bc.dist_dn)[keep], np.asarray(tp.bc.area)[keep], uv[keep])
    flux, diri = sub(top), sub(~top)
    if bc_dist_up > 0.0:     # the reference's MMS driver puts the boundary value on the face (dist_up = 0); CLM-style set-ups do not
        diri.dist_up = bc_dist_up * np.asarray(diri.dist_dn)
        flux.dist_up = bc_dist_up * np.asarray(flux.dist_dn)
    nf = flux.num_connections
    rng = np.random.default_rng(7)
    H = 40.0 + 20.0 * rng.random(nf)
    frac = 0.25 + 0.75 * rng.random(nf)
    dhsdT = 2.0 + 3.0 * rng.random(nf)          # positive: `dhsdT**area` of the defect mode stays real
    return tp, diri, tp.bc_temperature[~top], flux, H, frac, dhsdT


def thermal_land_units(n, seed=11):
    """per-cell land-unit data exercising every branch of ThermKSPTempSoilAuxVarCompute (auxvar :66-168): soil / wetland / land
    ice, shallow and deep (bedrock) layers, liquid and ice contents from dry to saturated, snow water with and without snow layers"""
    import numpy as np
    rng = np.random.default_rng(seed)
    land = dict(itype=rng.choice([1, 1, 1, 2, 3], n), shallow=(rng.random(n) < 0.8).astype(np.int32), por=0.3 + 0.3 * rng.random(n),
                csol=2.0e6 + 1.0e6 * rng.random(n), tkmg=1.5 + 2.0 * rng.random(n), tkdry=0.15 + 0.2 * rng.random(n),
                liq=np.where(rng.random(n) < 0.15, 0.0, 30.0 * rng.random(n)), ice=np.where(rng.random(n) < 0.5, 0.0, 20.0 * rng.random(n)),
                snow=5.0 * rng.random(n), nsnow=rng.choice([0, 0, 1, 3], n), tuning=0.5 + rng.random(n))
    return land


def to_thermal_flux_oracle(case, flux_first, defects=False, ksp_rtol=0.0, land=None, t_prev=None):
    import numpy as np
    from mpp_amd import constants as C
    from mpp_amd import mesh as M
    from oracle import thermal_oracle as TO
    tp, diri, tdir, flux, H, frac, dhsdT = case
    n = tp.mesh.ncells_local
    o = TO.ThermalOracle(n)
    o.set_mesh(tp.mesh.vol, tp.mesh.dz, tp.mesh.is_active)
    o.set_connections(tp.mesh.internal_connections())
    order = [("f", flux), ("d", diri)] if flux_first else [("d", diri), ("f", flux)]
    for kind, cs in order:
        o.add_condition(TO.COND_BC, TO.COND_HEAT_FLUX if kind == "f" else TO.COND_DIRICHLET, cs)
    ids = np.arange(1, n + 1); z = np.zeros(n)
    o.add_condition(TO.COND_SS, TO.COND_HEAT_RATE, M.ConnectionSet(np.zeros(n, dtype=np.int64), ids, z, z, z, np.zeros((n, 3))))
    if land is None:
        o.set_soils(TO.SOIL, 1, 0.1, 0.0, tp.tkdry, 0.0)
    else:
        o.set_soils(land["itype"], land["shallow"], land["por"], land["tkmg"], land["tkdry"], land["csol"])
    o.set_options(cnfac=tp.cnfac, ksp_type=C.KSP_GMRES, pc_type=C.PC_ILU0, ksp_rtol=ksp_rtol)
    o.set_soln_prev_clm(tp.exact if t_prev is None else t_prev)
    o.set_cell_data(TO.VAR_TUNING_FACTOR, 1.0 if land is None else land["tuning"])
    o.set_cell_data(TO.VAR_LIQ_AREAL_DEN, 0.0 if land is None else land["liq"])
    if land is not None:
        o.set_cell_data(TO.VAR_ICE_AREAL_DEN, land["ice"]); o.set_cell_data(TO.VAR_SNOW_WATER, land["snow"])
        o.set_cell_data(TO.VAR_NUM_SNOW_LYR, land["nsnow"])
    fid, did = (1, 2) if flux_first else (2, 1)
    o.set_condition_values(TO.COND_BC, did, tdir)
    o.set_condition_values(TO.COND_BC, fid, H)
    o.set_condition_aux(did, "active", np.ones(tdir.size)); o.set_condition_aux(fid, "active", np.ones(H.size))
    o.set_condition_aux(did, "frac", np.ones(tdir.size))
    o.set_condition_aux(fid, "frac", frac)
    o.set_condition_aux(fid, "dhsdT", dhsdT)
    o.set_condition_values(TO.COND_SS, 1, tp.heat_source)
    if defects:
        o.set_reference_defects(True)
    o.pre_step_dt()
    return o


def to_thermal_flux_gpu(case, flux_first, options=None, device_id=0, land=None, t_prev=None):
    import numpy as np
    from mpp_amd import constants as C
    from mpp_amd import thermal as T
    tp, diri, tdir, flux, H, frac, dhsdT = case
    n = tp.mesh.ncells_local
    mpp = T.MultiPhysicsProbThermal(device_id)
    mpp.Init(); mpp.SetName("thermal with a heat-flux boundary"); mpp.SetID(T.MPP_THERMAL_TBASED_KSP_CLM); mpp.SetMPIRank(0)
    mpp.SetNumMeshes(1)
    mpp.AddMesh(1, tp.mesh)
    mpp.AddGovEqn(T.GE_THERM_SOIL_TBASED, "soil temperature", 302)
    mpp.SetMeshesOfGoveqns()
    order = [("f", flux), ("d", diri)] if flux_first else [("d", diri), ("f", flux)]
    for kind, cs in order:
        mpp.soe.AddConditionInGovEqn(1, C.COND_BC, "bc " + kind, "-", T.COND_HEAT_FLUX if kind == "f" else C.COND_DIRICHLET, conn_set=cs)
    mpp.soe.AddConditionInGovEqn(1, C.COND_SS, "source", "W/m^2", T.COND_HEAT_RATE, region_type=T.ALL_CELLS)
    mpp.soe.SetOption("cnfac", tp.cnfac)
    for k, v in (options or {}).items():
        mpp.soe.SetOption(k, v)
    mpp.AllocateAuxVars()
    mpp.SetupProblem()
    if land is None:
        T.MPPThermalSetSoils(mpp, T.ISTSOIL, 1, 0.1, 0.0, 0.0, tp.tkdry)
    else:
        T.MPPThermalSetSoils(mpp, land["itype"], land["shallow"], land["por"], land["csol"], land["tkmg"], land["tkdry"])
    mpp.soe.SetSolnPrevCLM(tp.exact if t_prev is None else t_prev)
    mpp.soe.SetRDataFromCLM(C.AUXVAR_INTERNAL, T.VAR_TUNING_FACTOR, 1, np.ones(n) if land is None else land["tuning"])
    mpp.soe.SetRDataFromCLM(C.AUXVAR_INTERNAL, T.VAR_LIQ_AREAL_DEN, 1, np.zeros(n) if land is None else land["liq"])
    if land is not None:
        mpp.soe.SetRDataFromCLM(C.AUXVAR_INTERNAL, T.VAR_ICE_AREAL_DEN, 1, land["ice"])
        mpp.soe.SetRDataFromCLM(C.AUXVAR_INTERNAL, T.VAR_SNOW_WATER, 1, land["snow"])
        mpp.soe.SetRDataFromCLM(C.AUXVAR_INTERNAL, T.VAR_NUM_SNOW_LYR, 1, land["nsnow"].astype(float))
    fid, did = (1, 2) if flux_first else (2, 1)
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, did, tdir)
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, fid, H)
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, T.VAR_ACTIVE, did, np.ones(tdir.size)); mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, T.VAR_ACTIVE, fid, np.ones(H.size))
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, T.VAR_FRAC, did, np.ones(tdir.size))
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, T.VAR_FRAC, fid, frac)
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, T.VAR_DHS_DT, fid, dhsdT)
    mpp.soe.SetRDataFromCLM(C.AUXVAR_SS, C.VAR_BC_SS_CONDITION, 1, tp.heat_source)
    mpp.soe.PreStepDT()
    return mpp
