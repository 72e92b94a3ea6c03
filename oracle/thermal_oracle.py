"""ctypes binding of the thermal CPU oracle (oracle/thermal_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import vsfm_oracle as VO

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int)
SOIL, WETLAND, ICE = 1, 2, 3
COND_BC, COND_SS, COND_DIRICHLET, COND_HEAT_FLUX, COND_HEAT_RATE = 501, 502, 505, 507, 511
VAR_LIQ_AREAL_DEN, VAR_ICE_AREAL_DEN, VAR_TUNING_FACTOR, VAR_SNOW_WATER, VAR_NUM_SNOW_LYR = 1, 2, 3, 4, 5


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class ThermalOracle:
    def __init__(self, n):
        self.L = VO.lib()
        self.L.to_create.restype = ct.c_void_p
        self.L.to_create.argtypes = [ct.c_int]
        self.L.to_set_options.argtypes = [ct.c_void_p, ct.c_double, ct.c_int, ct.c_int, ct.c_double]
        self.L.to_step_dt.argtypes = [ct.c_void_p, ct.c_double, _ip]
        self.L.to_set_dtime.argtypes = [ct.c_void_p, ct.c_double]
        self.L.to_linear_iterations.restype = ct.c_long
        self.n = int(n)
        self.h = ct.c_void_p(self.L.to_create(self.n))

    def __del__(self):
        try:
            if self.h:
                self.L.to_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("thermal oracle call failed")

    def set_mesh(self, vol, dz, is_active=None):
        v, pv = _d(vol); z, pz = _d(dz)
        if is_active is None:
            self._chk(self.L.to_set_mesh(self.h, pv, pz, None))
        else:
            a, pa = _i(is_active)
            self._chk(self.L.to_set_mesh(self.h, pv, pz, pa))

    def set_connections(self, cs):
        up, pup = _i(cs.id_up); dn, pdn = _i(cs.id_dn)
        du, pdu = _d(cs.dist_up); dd, pdd = _d(cs.dist_dn); ar, par = _d(cs.area)
        self._chk(self.L.to_set_connections(self.h, cs.num_connections, pup, pdn, pdu, pdd, par))

    def add_condition(self, kind, cond_type, cs):
        dn, pdn = _i(cs.id_dn); du, pdu = _d(cs.dist_up); dd, pdd = _d(cs.dist_dn); ar, par = _d(cs.area)
        rc = self.L.to_add_condition(self.h, kind, cond_type, cs.num_connections, pdn, pdu, pdd, par)
        if rc < 0:
            raise RuntimeError("bad condition")
        return rc

    def set_soils(self, itype, is_soil_shallow, por, tkmg, tkdry, csol):
        a, pa = _i(np.broadcast_to(itype, (self.n,))); b, pb = _i(np.broadcast_to(is_soil_shallow, (self.n,)))
        c, pc = _d(np.broadcast_to(por, (self.n,))); d, pd = _d(np.broadcast_to(tkmg, (self.n,)))
        e, pe = _d(np.broadcast_to(tkdry, (self.n,))); f, pf = _d(np.broadcast_to(csol, (self.n,)))
        self._chk(self.L.to_set_soils(self.h, pa, pb, pc, pd, pe, pf))

    def set_cell_data(self, var, values):
        v, pv = _d(np.broadcast_to(values, (self.n,)))
        self._chk(self.L.to_set_cell_data(self.h, var, pv))

    def set_condition_values(self, kind, cid, values):
        v, pv = _d(values)
        self._chk(self.L.to_set_condition_values(self.h, kind, cid, v.size, pv))

    def set_options(self, cnfac=-1.0, ksp_type=VO.lib and 1, pc_type=1, ksp_rtol=0.0):
        self._chk(self.L.to_set_options(self.h, float(cnfac), int(ksp_type), int(pc_type), float(ksp_rtol)))

    def set_condition_aux(self, cid, var, values):
        """frac / dhsdT / is_active of a boundary condition's auxvars (var = 'frac' | 'dhsdT' | 'active')"""
        v, pv = _d(values)
        self._chk(self.L.to_set_condition_aux(self.h, cid, {"frac": 1, "dhsdT": 2, "active": 3}[var], v.size, pv))

    def set_reference_defects(self, on=True):
        """Reproduce GoveqnThermalKSPTemperatureSoilType.F90:906-908 (stale area) and :1215 (dhsdT**area) as written."""
        self._chk(self.L.to_set_reference_defects(self.h, int(bool(on))))

    def set_soln_prev_clm(self, T):
        v, pv = _d(T)
        self._chk(self.L.to_set_soln_prev_clm(self.h, pv))

    def pre_step_dt(self):
        self._chk(self.L.to_pre_step_dt(self.h))

    def step_dt(self, dt):
        conv = ct.c_int()
        self._chk(self.L.to_step_dt(self.h, float(dt), ct.byref(conv)))
        return bool(conv.value)

    def compute_rhs(self, dt):
        self.L.to_set_dtime(self.h, float(dt))
        b = np.zeros(self.n)
        self._chk(self.L.to_compute_rhs(self.h, b.ctypes.data_as(_dp)))
        return b

    def compute_operators(self):
        self._chk(self.L.to_compute_operators(self.h))
        nnz = self.L.to_nnz(self.h)
        rp = np.zeros(self.n + 1, dtype=np.int32); cj = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
        self._chk(self.L.to_get_operator(self.h, rp.ctypes.data_as(_ip), cj.ctypes.data_as(_ip), val.ctypes.data_as(_dp)))
        return rp, cj, val

    def get_soln(self):
        T = np.zeros(self.n)
        self._chk(self.L.to_get_soln(self.h, T.ctypes.data_as(_dp)))
        return T

    def linear_iterations(self):
        return int(self.L.to_linear_iterations(self.h))
