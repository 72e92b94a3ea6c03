"""Host-side mirror of the reference's coupled thermal-hydrology operator API over the C ABI.

  mpp_th_type          src/mpp/mpp/MultiPhysicsProbTH.F90            -> MultiPhysicsProbTH
  sysofeqns_th_type    src/mpp/soe/SystemOfEquationsTHType.F90       -> SystemOfEquationsTH
  MPPTHSetSoils        src/mpp/mpp/MultiPhysicsProbTH.F90:75-176
Governing equation 1 is the Richards mass equation (GE_RE), 2 the enthalpy-based heat equation
(GE_THERM_SOIL_EBASED); unknowns are field-blocked [P(1..N); T(1..N)].  Call order as in
src/driver/standalone/thermal-e/mass_and_heat_model_problem.F90:112-139.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import constants as C
from . import mesh as M
from .vsfm import (MPPError, MultiPhysicsProbVSFM, SystemOfEquationsVSFM, VSFMMPPSetDensityType,
                   VSFMMPPSetSaturationFunction, VSFMMPPSetSoilPermeability, VSFMMPPSetSoilPorosity, _SATFUNC_NAMES, _d, _i)

GE_THERM_SOIL_EBASED = 205
MPP_TH_SNES_CLM = 14
VAR_TEMPERATURE = 605
INT_ENERGY_ENTHALPY_CONSTANT, INT_ENERGY_ENTHALPY_IFC67 = 1, 2


class SystemOfEquationsTH(SystemOfEquationsVSFM):
    """sysofeqns_th_type."""

    def __init__(self, device_id=0):
        super().__init__(device_id)
        self._chk(self.L.mppgpu_set_option(self.h, b"problem_type", b"th"))

    def AddConditionInGovEqn(self, igoveqn, ss_or_bc_type, name, unit, cond_type, region_type=None, conn_set=None,
                             swap_order=False):
        """SOEBaseAddConditionInGovEqn (SystemOfEquationsBaseType.F90:1260-1290); returns the SoE-level id."""
        if igoveqn not in (1, 2):
            raise MPPError("Attempting to add condition for governing equation that is not in the list")
        if conn_set is None:
            if region_type in (C.SOIL_CELLS, 408):   # ALL_CELLS
                n = self.mesh.ncells_local
                z = np.zeros(n)
                conn_set = M.ConnectionSet(np.full(n, -1), np.arange(1, n + 1), z, z, self.mesh.area_xy.copy(), np.zeros((n, 3)))
            else:
                conn_set = M.create_region_connection_set(self.mesh, region_type)
        dn, pdn = _i(conn_set.id_dn)
        du, pdu = _d(conn_set.dist_up); dd, pdd = _d(conn_set.dist_dn); ar, par = _d(conn_set.area)
        uv, puv = _d(conn_set.unit_vec)
        cid = ct.c_int()
        self._chk(self.L.mppgpu_th_add_condition(self.h, int(igoveqn), ss_or_bc_type, cond_type, conn_set.num_connections,
                                                 pdn, pdu, pdd, par, puv, int(swap_order), ct.byref(cid)))
        sizes = self._bc_sizes if ss_or_bc_type == C.COND_BC else self._ss_sizes
        sizes.insert(cid.value - 1, conn_set.num_connections)
        return cid.value

    def SetBoundaryAuxVarPressure(self, soe_auxvar_id, data_1d):
        """cur_goveq%aux_vars_bc(:)%pressure of the heat equation (mass_and_heat_model_problem.F90:609-617)"""
        self.SetDataFromCLM(C.AUXVAR_BC, C.VAR_PRESSURE, soe_auxvar_id, data_1d)

    def SetBoundaryAuxVarTemperature(self, soe_auxvar_id, data_1d):
        """cur_goveq%aux_vars_bc(:)%temperature of the mass equation (th_mms_problem.F90:862-875)"""
        self.SetDataFromCLM(C.AUXVAR_BC, VAR_TEMPERATURE, soe_auxvar_id, data_1d)

    def GetDataForCLM(self, soe_auxvar_type, var_type, soe_auxvar_id=-1, data_1d=None):
        """SOETHGetDataForCLM (SystemOfEquationsTHType.F90:1066-1123): the reference returns 2N values (the SoE auxvars
        of both equations) of which the drivers keep the first N for VAR_PRESSURE and the last N for VAR_TEMPERATURE;
        this returns that N-vector."""
        if data_1d is None:
            data_1d = np.zeros(self.mesh.ncells_local)
        return super().GetDataForCLM(soe_auxvar_type, var_type, soe_auxvar_id, data_1d)


class MultiPhysicsProbTH(MultiPhysicsProbVSFM):
    """mpp_th_type"""

    def __init__(self, device_id=0):
        self.name = ""
        self.id = 0
        self.mpi_rank = 0
        self.meshes = []
        self.soe = SystemOfEquationsTH(device_id)
        self._setup = False
        self._goveqns = []

    def SetID(self, mpp_id):
        if mpp_id != MPP_TH_SNES_CLM:
            raise MPPError("Unknown mpp_type")
        self.id = mpp_id

    def AddGovEqn(self, geq_type, name, mesh_itype):
        want = (C.GE_RE, GE_THERM_SOIL_EBASED)
        k = len(self._goveqns)
        if k >= 2 or geq_type != want[k]:
            raise MPPError("the coupled system is GE_RE followed by GE_THERM_SOIL_EBASED")
        self._goveqns.append(geq_type)
        self.soe.ngoveqns = len(self._goveqns)

    def GovEqnSetInternalCouplingVars(self, ieqn, nvars, var_ids, goveqn_ids):
        """th_mpp%GovEqnSetInternalCouplingVars: the mass equation takes VAR_TEMPERATURE from equation 2, the heat
        equation VAR_PRESSURE from equation 1 -- the only coupling this path implements."""
        ok = (ieqn == 1 and list(var_ids) == [VAR_TEMPERATURE] and list(goveqn_ids) == [2]) or \
             (ieqn == 2 and list(var_ids) == [C.VAR_PRESSURE] and list(goveqn_ids) == [1])
        if not ok or nvars != 1:
            raise MPPError("unsupported coupling variables")

    def Restart(self, pressure, temperature):
        """the drivers' set_initial_conditions: soln = soln_prev = [P; T]"""
        n = self.meshes[0].ncells_local
        d, pd = _d(np.concatenate([np.asarray(pressure, dtype=np.float64), np.asarray(temperature, dtype=np.float64)]))
        if d.size != 2 * n:
            raise MPPError("size(data_1d) /= 2 ncells_local")
        self.soe._chk(self.soe.L.mppgpu_restart(self.soe.h, d.size, pd))


def MPPTHSetSoilsSI(th_mpp, perm, por, satfunc_type, alpha, lam, sat_res, density_type, int_energy_enthalpy_type, heat_cap,
                    therm_cond_wet, therm_cond_dry, therm_alpha, den_soil, energy_perm=None):
    """the per-equation setters of th_mms_problem.F90:638-755 (SI units, per-cell arrays); saturation function and
    porosity are shared by both equations"""
    n = th_mpp.meshes[0].ncells_local
    b = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (n,)))
    VSFMMPPSetDensityType(th_mpp, 1, density_type)
    th_mpp.soe.SetOption("int_energy_enthalpy_type", int(int_energy_enthalpy_type))
    perm = np.asarray(perm, dtype=np.float64)
    if perm.ndim == 1:
        perm = np.stack([perm] * 3, axis=1)
    VSFMMPPSetSoilPermeability(th_mpp, 1, perm[:, 0], perm[:, 1], perm[:, 2])
    VSFMMPPSetSoilPorosity(th_mpp, 1, b(por))
    VSFMMPPSetSaturationFunction(th_mpp, 1, np.ascontiguousarray(np.broadcast_to(satfunc_type, (n,)), dtype=np.int32), b(alpha), b(lam), b(sat_res))
    args = [_d(b(a)) for a in (heat_cap, therm_cond_wet, therm_cond_dry, therm_alpha, den_soil)]
    if energy_perm is None:
        pp = [None, None, None]
        keep = []
    else:
        keep = [_d(b(a)) for a in energy_perm]
        pp = [k[1] for k in keep]
    th_mpp.soe._chk(th_mpp.soe.L.mppgpu_th_set_soils(th_mpp.soe.h, *[a[1] for a in args], *pp))


def MPPTHSetSoils(th_mpp, filter_thermal, watsat, csol, tkdry, hksat, bsw, sucsat, residual_sat, vsfm_satfunc_type,
                  density_type, int_energy_enthalpy_type):
    """MPPTHSetSoils (MultiPhysicsProbTH.F90:75-601): CLM units; (ncolumns, nlevgrnd) arrays, both equations.  The heat
    equation gets therm_alpha 0.45, wet conductivity 1.3, soil density 2700 and keeps its default permeability."""
    mesh = th_mpp.meshes[0]
    nlev = mesh.nlev
    n = mesh.ncells_local
    ncol = n // nlev
    filt = np.asarray(filter_thermal)[:ncol]
    act = np.nonzero(filt == 1)[0]
    if act.size == 0:
        raise MPPError("No active soil column found")
    col_id = np.where(filt == 1, np.arange(ncol), act[0])

    def pick(a):
        return np.asarray(a, dtype=np.float64).reshape(-1, nlev)[col_id, :].ravel()

    vish2o = 0.001002
    perm = pick(hksat) * vish2o / (C.DENH2O * C.GRAV_CLM) * 0.001     # :505
    alpha = 1.0 / (pick(sucsat) * C.GRAV_CLM)                         # :509
    lam = 1.0 / pick(bsw)                                             # :512
    if vsfm_satfunc_type not in _SATFUNC_NAMES:
        raise MPPError("ERROR:: Unknown vsfm_satfunc_type = " + str(vsfm_satfunc_type))
    MPPTHSetSoilsSI(th_mpp, perm, pick(watsat), _SATFUNC_NAMES[vsfm_satfunc_type], alpha, lam, pick(residual_sat), density_type,
                    int_energy_enthalpy_type, pick(csol), 1.3, pick(tkdry), 0.45, 2700.0, energy_perm=None)
