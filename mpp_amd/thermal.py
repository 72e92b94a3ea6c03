"""Host-side mirror of the reference's thermal (soil heat diffusion, KSP) operator API over the C ABI.

  mpp_thermal_type          src/mpp/mpp/MultiPhysicsProbThermal.F90           -> MultiPhysicsProbThermal
  sysofeqns_thermal_type    src/mpp/soe/SystemOfEquationsThermalType.F90      -> SystemOfEquationsThermal
  MPPThermalSetSoils        src/mpp/mpp/MultiPhysicsProbThermal.F90:76-208
Call order as in src/driver/standalone/thermal/thermal_mms_problem.F90:236-262.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import constants as C
from . import mesh as M
from .vsfm import MPPError, SystemOfEquationsVSFM, _d, _i

# ids of MultiPhysicsProbConstants.F90 used on this path
COND_HEAT_FLUX, COND_HEAT_RATE = 507, 511
VAR_LIQ_AREAL_DEN, VAR_ICE_AREAL_DEN, VAR_FRAC, VAR_SNOW_WATER, VAR_NUM_SNOW_LYR = 615, 617, 618, 619, 620
VAR_DHS_DT, VAR_ACTIVE, VAR_TUNING_FACTOR = 621, 624, 630
GE_THERM_SOIL_TBASED, MPP_THERMAL_TBASED_KSP_CLM, ALL_CELLS = 202, 12, 408
ISTSOIL, ISTWET, ISTICE = 1, 2, 3


class SystemOfEquationsThermal(SystemOfEquationsVSFM):
    """sysofeqns_thermal_type with one goveqn_thermal_ksp_temp_soil_type."""

    def __init__(self, device_id=0):
        super().__init__(device_id)
        self._chk(self.L.mppgpu_set_option(self.h, b"problem_type", b"thermal"))

    def AddConditionInGovEqn(self, igoveqn, ss_or_bc_type, name, unit, cond_type, region_type=None, conn_set=None,
                             swap_order=False):
        if conn_set is None and region_type == ALL_CELLS:
            region_type = C.SOIL_CELLS
        return super().AddConditionInGovEqn(igoveqn, ss_or_bc_type, name, unit, cond_type, region_type, conn_set, swap_order)

    def SetSolnPrevCLM(self, data_1d):
        d, pd = _d(data_1d)
        self._chk(self.L.mppgpu_thermal_set_soln_prev_clm(self.h, d.size, pd))

    def SetRDataFromCLM(self, soe_auxvar_type, var_type, soe_auxvar_id, data_1d):
        d, pd = _d(data_1d)
        self._chk(self.L.mppgpu_set_data(self.h, soe_auxvar_type, var_type, soe_auxvar_id, d.size, pd))

    def PreStepDT(self):
        self._chk(self.L.mppgpu_thermal_pre_step_dt(self.h))

    def StepDT(self, dt, nstep=1):
        conv, its = ct.c_int(), ct.c_int()
        self._chk(self.L.mppgpu_thermal_step_dt(self.h, float(dt), ct.byref(conv), ct.byref(its)))
        self.last_linear_iterations = its.value
        return bool(conv.value), (2 if conv.value else -3)

    def GetSoln(self, data_1d=None):
        if data_1d is None:
            data_1d = np.zeros(self.mesh.ncells_local)
        self._chk(self.L.mppgpu_thermal_get_soln(self.h, data_1d.size, data_1d.ctypes.data_as(ct.POINTER(ct.c_double))))
        return data_1d

    def ComputeRHS(self, dt):
        b = np.zeros(self.mesh.ncells_local)
        self._chk(self.L.mppgpu_thermal_compute_rhs(self.h, float(dt), b.ctypes.data_as(ct.POINTER(ct.c_double))))
        return b

    def ComputeOperators(self, dt):
        nnz = ct.c_int()
        self._chk(self.L.mppgpu_jacobian_nnz(self.h, ct.byref(nnz)))
        n = self.mesh.ncells_local
        rp = np.zeros(n + 1, dtype=np.int32); cj = np.zeros(nnz.value, dtype=np.int32); val = np.zeros(nnz.value)
        self._chk(self.L.mppgpu_thermal_compute_operators(self.h, float(dt), rp.ctypes.data_as(ct.POINTER(ct.c_int)),
                                                          cj.ctypes.data_as(ct.POINTER(ct.c_int)),
                                                          val.ctypes.data_as(ct.POINTER(ct.c_double))))
        return rp, cj, val


class MultiPhysicsProbThermal:
    """mpp_thermal_type"""

    def __init__(self, device_id=0):
        self.soe = SystemOfEquationsThermal(device_id)
        self.meshes = []
        self.name, self.id = "", 0

    def Init(self):
        return self

    def SetName(self, name):
        self.name = name

    def SetID(self, mpp_id):
        if mpp_id != MPP_THERMAL_TBASED_KSP_CLM:
            raise MPPError("Unknown mpp_type")
        self.id = mpp_id

    def SetMPIRank(self, rank):
        self.mpi_rank = rank

    def SetNumMeshes(self, n):
        if n != 1:
            raise MPPError("only one mesh is supported")

    def AddMesh(self, imesh, mesh: M.Mesh):
        self.meshes = [mesh]

    def AddGovEqn(self, geq_type, name, mesh_itype):
        if geq_type != GE_THERM_SOIL_TBASED:
            raise MPPError("only the soil temperature equation (GE_THERM_SOIL_TBASED) is on this path")
        self.soe.ngoveqns += 1

    def SetMeshesOfGoveqns(self):
        self.soe._set_mesh(self.meshes[0])

    def AllocateAuxVars(self):
        pass

    def SetupProblem(self):
        self.soe._chk(self.soe.L.mppgpu_setup(self.soe.h))


def MPPThermalSetSoils(therm_mpp, itype, is_soil_shallow, watsat, csol, tkmg, tkdry):
    """per-cell form of MPPThermalSetSoils (the (column, level) -> cell association is the caller's, see
    tests/problems.py: thermal_mms for the reference driver's)"""
    mesh = therm_mpp.meshes[0]
    n = mesh.ncells_local
    a, pa = _i(np.broadcast_to(itype, (n,))); b, pb = _i(np.broadcast_to(is_soil_shallow, (n,)))
    c, pc = _d(np.broadcast_to(watsat, (n,))); d, pd = _d(np.broadcast_to(csol, (n,)))
    e, pe = _d(np.broadcast_to(tkmg, (n,))); f, pf = _d(np.broadcast_to(tkdry, (n,))); g, pg = _d(mesh.dz)
    therm_mpp.soe._chk(therm_mpp.soe.L.mppgpu_thermal_set_soils(therm_mpp.soe.h, pa, pb, pc, pd, pe, pf, pg))
