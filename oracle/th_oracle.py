"""ctypes binding of the coupled thermal-hydrology CPU oracle (oracle/th_oracle.c).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import vsfm_oracle as VO
from .vsfm_oracle import _chk, _d, _dp, _i, _ip

COND_HEAT_RATE = 511
INT_ENERGY_ENTHALPY_CONSTANT = 1
INT_ENERGY_ENTHALPY_IFC67 = 2

_ready = False


def lib():
    global _ready
    L = VO.lib()
    if not _ready:
        L.tho_create.restype = ct.c_void_p
        L.tho_create.argtypes = [ct.c_int]
        L.tho_mass.restype = ct.c_void_p
        L.tho_mass.argtypes = [ct.c_void_p]
        L.tho_global.restype = ct.c_void_p
        L.tho_global.argtypes = [ct.c_void_p]
        L.tho_destroy.argtypes = [ct.c_void_p]
        L.tho_set_eos.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
        L.tho_energy_add_condition.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, _ip, _dp, _dp, _dp, _dp, _ip]
        L.tho_energy_set_soils.argtypes = [ct.c_void_p, _dp, _ip] + [_dp] * 11
        L.tho_energy_set_porosity_linear.argtypes = [ct.c_void_p, _dp, _dp]
        L.tho_energy_set_condition_values.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, _dp]
        L.tho_energy_set_bc_pressure.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, _dp]
        L.tho_mass_set_bc_temperature.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, _dp]
        L.tho_restart.argtypes = [ct.c_void_p, _dp, _dp]
        L.tho_setup.argtypes = [ct.c_void_p]
        L.tho_set_dtime.argtypes = [ct.c_void_p, ct.c_double]
        L.tho_pre_solve.argtypes = [ct.c_void_p]
        L.tho_residual.argtypes = [ct.c_void_p, _dp, _dp]
        L.tho_jacobian_nnz.argtypes = [ct.c_void_p]
        L.tho_jacobian.argtypes = [ct.c_void_p, _dp, _ip, _ip, _dp]
        L.tho_step_dt.argtypes = [ct.c_void_p, ct.c_double, _ip, _ip]
        L.tho_get_solution.argtypes = [ct.c_void_p, _dp, _dp]
        L.tho_internal_energy_enthalpy.argtypes = [ct.c_double, ct.c_double, ct.c_int, ct.c_double, ct.c_double, ct.c_double, _dp]
        L.tho_auxvar_compute.argtypes = [ct.c_int, ct.c_double, ct.c_double, ct.c_double, ct.c_double, ct.c_int, ct.c_int,
                                         ct.c_double, ct.c_double, ct.c_double, ct.c_double, ct.c_double, _dp]
        _ready = True
    return L


def internal_energy_enthalpy(P, t_K, itype, den, dden_dT, dden_dP):
    """(U, H, dU_dT, dH_dT, dU_dP, dH_dP); den in kg m^-3 (EOSWaterMod.F90:589-707)"""
    out = np.zeros(6)
    lib().tho_internal_energy_enthalpy(P, t_K, itype, den, dden_dT, dden_dP, out.ctypes.data_as(_dp))
    return tuple(out)


AUX_FIELDS = ("sat", "dsat_dP", "kr", "dkr_dP", "den", "dden_dP", "dden_dT", "ul", "hl", "dul_dP", "dhl_dP", "dul_dT",
              "dhl_dT", "therm_cond", "dtherm_cond_dP")


def auxvar_compute(sat_func_type, sat_res, alpha, lam, por, density_type, ie_type, tcond_wet, tcond_dry, talpha,
                   pressure, temperature):
    out = np.zeros(15)
    lib().tho_auxvar_compute(sat_func_type, sat_res, alpha, lam, por, density_type, ie_type, tcond_wet, tcond_dry, talpha,
                             pressure, temperature, out.ctypes.data_as(_dp))
    return dict(zip(AUX_FIELDS, out))


class THOracle:
    """th_mpp of the reference drivers (mass_and_heat_model_problem.F90, th_mms_problem.F90): `mass` is the Richards
    equation (an OracleProblem: mesh, connections, its conditions and soils), the heat equation is configured here."""

    def __init__(self, ncells: int):
        self.L = lib()
        self.n = int(ncells)
        self.h = ct.c_void_p(self.L.tho_create(self.n))
        self.mass = VO.OracleProblem(self.n, handle=ct.c_void_p(self.L.tho_mass(self.h)))
        self.glob = VO.OracleProblem(2 * self.n, handle=ct.c_void_p(self.L.tho_global(self.h)))

    def close(self):
        if self.h:
            self.mass.close(); self.glob.close()
            self.L.tho_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_eos(self, density_type, ie_type):
        _chk(self.L.tho_set_eos(self.h, int(density_type), int(ie_type)))

    def energy_add_condition(self, kind, cond_type, cs):
        dn, pdn = _i(cs.id_dn)
        du, pdu = _d(cs.dist_up); dd, pdd = _d(cs.dist_dn); ar, par = _d(cs.area); uv, puv = _d(cs.unit_vec)
        cid = ct.c_int(0)
        _chk(self.L.tho_energy_add_condition(self.h, kind, cond_type, cs.num_connections, pdn, pdu, pdd, par, puv, ct.byref(cid)))
        return cid.value

    def energy_set_soils(self, por, satfunc_type, alpha, lam, sat_res, heat_cap, tcond_wet, tcond_dry, talpha, den_soil,
                         perm=None):
        n = self.n
        b = lambda a: _d(np.broadcast_to(a, (n,)))
        a_por, p_por = b(por); t, pt = _i(np.broadcast_to(satfunc_type, (n,)))
        a_al, p_al = b(alpha); a_l, p_l = b(lam); a_s, p_s = b(sat_res)
        a_hc, p_hc = b(heat_cap); a_w, p_w = b(tcond_wet); a_d, p_d = b(tcond_dry); a_ta, p_ta = b(talpha); a_ds, p_ds = b(den_soil)
        if perm is None:
            px = py = pz = None
        else:
            (ax, px), (ay, py), (az, pz) = b(perm[0]), b(perm[1]), b(perm[2])
        _chk(self.L.tho_energy_set_soils(self.h, p_por, pt, p_al, p_l, p_s, px, py, pz, p_hc, p_w, p_d, p_ta, p_ds))

    def energy_set_porosity_linear(self, press_ref, slope):
        """POROSITY_LINEAR for the heat equation's auxvars (por of energy_set_soils is then porosity_base)"""
        (a, pa), (b, pb) = _d(press_ref), _d(slope)
        _chk(self.L.tho_energy_set_porosity_linear(self.h, pa, pb))

    def energy_set_condition_values(self, kind, cond_id, values):
        v, pv = _d(values)
        _chk(self.L.tho_energy_set_condition_values(self.h, kind, cond_id, v.size, pv))

    def energy_set_bc_pressure(self, cond_id, values):
        v, pv = _d(values)
        _chk(self.L.tho_energy_set_bc_pressure(self.h, cond_id, v.size, pv))

    def mass_set_bc_temperature(self, cond_id, values):
        v, pv = _d(values)
        _chk(self.L.tho_mass_set_bc_temperature(self.h, cond_id, v.size, pv))

    def restart(self, pressure, temperature):
        a, pa = _d(pressure); b, pb = _d(temperature)
        _chk(self.L.tho_restart(self.h, pa, pb))

    def setup(self):
        _chk(self.L.tho_setup(self.h))

    def set_dtime(self, dt):
        _chk(self.L.tho_set_dtime(self.h, float(dt)))

    def pre_solve(self):
        _chk(self.L.tho_pre_solve(self.h))

    def residual(self, x):
        xx, px = _d(x)
        f = np.zeros(2 * self.n)
        _chk(self.L.tho_residual(self.h, px, f.ctypes.data_as(_dp)))
        return f

    def jacobian(self, x):
        xx, px = _d(x)
        nnz = self.L.tho_jacobian_nnz(self.h)
        rp = np.zeros(2 * self.n + 1, dtype=np.int32); cj = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
        _chk(self.L.tho_jacobian(self.h, px, rp.ctypes.data_as(_ip), cj.ctypes.data_as(_ip), val.ctypes.data_as(_dp)))
        return rp, cj, val

    def step_dt(self, dt):
        conv, reason = ct.c_int(0), ct.c_int(0)
        _chk(self.L.tho_step_dt(self.h, float(dt), ct.byref(conv), ct.byref(reason)))
        return bool(conv.value), reason.value

    def solution(self):
        p = np.zeros(self.n); t = np.zeros(self.n)
        _chk(self.L.tho_get_solution(self.h, p.ctypes.data_as(_dp), t.ctypes.data_as(_dp)))
        return p, t

    def counters(self):
        return self.glob.counters()
