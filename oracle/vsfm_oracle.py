"""ctypes binding of the CPU oracle (oracle/vsfm_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg.  Nothing under mpp_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvsfm_oracle.so")

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("vsfm_oracle.c", "thermal_oracle.c", "th_oracle.c", "vsfm_oracle.h",
                                             "vsfm_oracle_internal.h", "th_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # one thread by default: deterministic sums, and no spinning OpenMP team on boxes whose visible CPU
        # count is far above their CPU share; callers that want a threaded baseline call set_num_threads().
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = ct.CDLL(_SO)
        L.vo_create.restype = ct.c_void_p
        L.vo_create.argtypes = [ct.c_int]
        L.vo_last_error.restype = ct.c_char_p
        for name in ("vo_destroy", "vo_set_mesh", "vo_set_connections", "vo_add_condition", "vo_set_soil_permeability",
                     "vo_set_soil_porosity", "vo_set_soil_porosity_linear", "vo_set_saturation_function", "vo_set_density_type",
                     "vo_set_condition_values", "vo_set_source_sink_auxvar", "vo_set_lateral_connections", "vo_get_condition_values", "vo_pre_step_dt", "vo_post_step_dt", "vo_set_frac_liq_sat", "vo_restart", "vo_set_gmres_restart", "vo_set_blocks", "vo_set_solver", "vo_set_linesearch", "vo_set_dtime",
                     "vo_pre_solve", "vo_residual", "vo_jacobian_nnz", "vo_jacobian", "vo_pc_setup_apply",
                     "vo_ksp_solve", "vo_step_dt", "vo_get", "vo_get_counters"):
            getattr(L, name).argtypes = None
        L.vo_set_dtime.argtypes = [ct.c_void_p, ct.c_double]
        L.vo_compute_lateral_flux.argtypes = [ct.c_void_p, ct.c_double, ct.c_int]
        L.vo_step_dt.argtypes = [ct.c_void_p, ct.c_double, ct.c_int, _ip, _ip]
        L.vo_set_solver.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_double, ct.c_double, ct.c_double,
                                    ct.c_double, ct.c_int]
        L.vo_ksp_solve.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_double, _ip, _ip, _dp, _dp, _dp, _ip, _ip]
        L.vo_density.argtypes = [ct.c_double, ct.c_double, ct.c_int, _dp, _dp, _dp]
        L.vo_auxvar_compute.argtypes = [ct.c_int, ct.c_double, ct.c_double, ct.c_double, ct.c_double, ct.c_int,
                                        ct.c_double, ct.c_double, ct.c_double, _dp]
        L.vo_richards_flux.argtypes = [_dp, _dp, _dp, _dp, ct.c_double, ct.c_double, ct.c_double, _dp, ct.c_int,
                                       ct.c_int, ct.c_int, _dp, _dp, _dp]
        L.vo_linesearch_switches.restype = ct.c_longlong
        L.vo_linesearch_switches.argtypes = [ct.c_void_p]
        L.vo_set_num_threads.argtypes = [ct.c_int]
        L.vo_set_num_threads(1)
        _lib = L
    return _lib


def set_num_threads(n: int):
    lib().vo_set_num_threads(int(n))


class OracleError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise OracleError(lib().vo_last_error().decode())


def density(p, t_K, density_type):
    den, dp_, dT = ct.c_double(), ct.c_double(), ct.c_double()
    lib().vo_density(p, t_K, density_type, ct.byref(den), ct.byref(dp_), ct.byref(dT))
    return den.value, dp_.value, dT.value


def auxvar_compute(sat_func_type, sat_res, alpha, lambda_or_m, por, density_type, pressure,
                   temperature=298.15, frac_liq_sat=1.0):
    out = np.zeros(8)
    lib().vo_auxvar_compute(sat_func_type, sat_res, alpha, lambda_or_m, por, density_type, temperature,
                            frac_liq_sat, pressure, out.ctypes.data_as(_dp))
    return dict(zip(("sat", "dsat_dP", "kr", "dkr_dP", "den", "dden_dP", "por", "dpor_dP"), out))


def richards_flux(aux_up, perm_up, aux_dn, perm_dn, area, dist_up, dist_dn, unitvec, internal_conn, swap_order,
                  cond_type):
    f, a, b = ct.c_double(), ct.c_double(), ct.c_double()
    au, pau = _d(aux_up); pu, ppu = _d(perm_up); ad, pad = _d(aux_dn); pd_, ppd = _d(perm_dn); uv, puv = _d(unitvec)
    lib().vo_richards_flux(pau, ppu, pad, ppd, area, dist_up, dist_dn, puv, int(internal_conn), int(swap_order),
                           cond_type, ct.byref(f), ct.byref(a), ct.byref(b))
    return f.value, a.value, b.value


class OracleProblem:
    """Mirrors the calls the reference drivers make (vsfm_celia1990_problem.F90:106-137)."""

    def __init__(self, ncells: int, handle=None):
        self.L = lib()
        self.n = int(ncells)
        self._owned = handle is None      # a borrowed handle belongs to a THOracle (oracle/th_oracle.py)
        self.h = ct.c_void_p(self.L.vo_create(self.n)) if handle is None else handle
        self._keep = []

    def close(self):
        if self.h and self._owned:
            self.L.vo_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mesh(self, vol, is_active=None):
        v, pv = _d(vol)
        if is_active is None:
            _chk(self.L.vo_set_mesh(self.h, pv, None))
        else:
            a, pa = _i(is_active)
            _chk(self.L.vo_set_mesh(self.h, pv, pa))

    def set_connections(self, cs):
        up, pup = _i(cs.id_up); dn, pdn = _i(cs.id_dn)
        du, pdu = _d(cs.dist_up); dd, pdd = _d(cs.dist_dn); ar, par = _d(cs.area); uv, puv = _d(cs.unit_vec)
        _chk(self.L.vo_set_connections(self.h, cs.num_connections, pup, pdn, pdu, pdd, par, puv))

    def add_condition(self, kind, cond_type, cs, swap_order=False):
        dn, pdn = _i(cs.id_dn)
        du, pdu = _d(cs.dist_up); dd, pdd = _d(cs.dist_dn); ar, par = _d(cs.area); uv, puv = _d(cs.unit_vec)
        rc = self.L.vo_add_condition(self.h, kind, cond_type, cs.num_connections, pdn, pdu, pdd, par, puv,
                                     int(swap_order))
        if rc < 0:
            raise OracleError(self.L.vo_last_error().decode())
        return rc

    def set_soil_permeability(self, px, py, pz):
        a, pa = _d(px); b, pb = _d(py); c, pc = _d(pz)
        _chk(self.L.vo_set_soil_permeability(self.h, pa, pb, pc))

    def set_soil_porosity(self, por):
        a, pa = _d(por)
        _chk(self.L.vo_set_soil_porosity(self.h, pa))

    def set_soil_porosity_linear(self, por_base, press_ref, slope):
        a, pa = _d(por_base); b, pb = _d(press_ref); c, pc = _d(slope)
        _chk(self.L.vo_set_soil_porosity_linear(self.h, pa, pb, pc))

    def set_saturation_function(self, satfunc_type, alpha, lam, sat_res):
        t, pt = _i(np.broadcast_to(satfunc_type, (self.n,)))
        a, pa = _d(np.broadcast_to(alpha, (self.n,))); l, pl = _d(np.broadcast_to(lam, (self.n,)))
        s, ps = _d(np.broadcast_to(sat_res, (self.n,)))
        _chk(self.L.vo_set_saturation_function(self.h, pt, pa, pl, ps))

    def set_density_type(self, t):
        _chk(self.L.vo_set_density_type(self.h, int(t)))

    def set_condition_values(self, kind, cond_id, values):
        v, pv = _d(values)
        _chk(self.L.vo_set_condition_values(self.h, kind, cond_id, v.size, pv))

    def set_lateral_connections(self, cs):
        up, pup = _i(cs.id_up); dn, pdn = _i(cs.id_dn)
        du, pdu = _d(cs.dist_up); dd, pdd = _d(cs.dist_dn); ar, par = _d(cs.area); uv, puv = _d(cs.unit_vec)
        _chk(self.L.vo_set_lateral_connections(self.h, cs.num_connections, pup, pdn, pdu, pdd, par, puv))

    def compute_lateral_flux(self, dt, ss_cond_id):
        _chk(self.L.vo_compute_lateral_flux(self.h, ct.c_double(dt), int(ss_cond_id)))

    def get_condition_values(self, kind, cond_id, n):
        out = np.zeros(n)
        _chk(self.L.vo_get_condition_values(self.h, kind, cond_id, n, out.ctypes.data_as(_dp)))
        return out

    def pre_step_dt(self):
        _chk(self.L.vo_pre_step_dt(self.h))

    def post_step_dt(self):
        _chk(self.L.vo_post_step_dt(self.h))

    def set_frac_liq_sat(self, frac):
        v, pv = _d(np.broadcast_to(frac, (self.n,)))
        _chk(self.L.vo_set_frac_liq_sat(self.h, pv))

    def set_source_sink_auxvar(self, var_type, values):
        v, pv = _d(values)
        _chk(self.L.vo_set_source_sink_auxvar(self.h, int(var_type), v.size, pv))

    def restart(self, pressure):
        v, pv = _d(pressure)
        _chk(self.L.vo_restart(self.h, pv))

    def set_blocks(self, block_ptr, block_cells):
        a, pa = _i(block_ptr); b, pb = _i(block_cells)
        _chk(self.L.vo_set_blocks(self.h, a.size - 1, pa, pb))

    def set_solver(self, ksp_type, pc_type, ksp_rtol=0.0, snes_rtol=0.0, snes_atol=0.0, snes_stol=0.0, snes_max_it=0):
        _chk(self.L.vo_set_solver(self.h, ksp_type, pc_type, ksp_rtol, snes_rtol, snes_atol, snes_stol, snes_max_it))

    def set_gmres_restart(self, restart):
        _chk(self.L.vo_set_gmres_restart(self.h, int(restart)))

    # operator level
    LINESEARCH = {"basic": 1, "bt": 2, "l2": 3, "cp": 4}

    def set_linesearch(self, name="bt", use_dynamic=False):
        _chk(self.L.vo_set_linesearch(self.h, self.LINESEARCH[name], int(bool(use_dynamic))))

    def linesearch_switches(self):
        return int(self.L.vo_linesearch_switches(self.h))

    def set_dtime(self, dt):
        _chk(self.L.vo_set_dtime(self.h, float(dt)))

    def pre_solve(self):
        _chk(self.L.vo_pre_solve(self.h))

    def residual(self, x):
        xx, px = _d(x)
        f = np.zeros(self.n)
        _chk(self.L.vo_residual(self.h, px, f.ctypes.data_as(_dp)))
        return f

    def jacobian(self, x):
        xx, px = _d(x)
        nnz = self.L.vo_jacobian_nnz(self.h)
        rp = np.zeros(self.n + 1, dtype=np.int32); cj = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
        _chk(self.L.vo_jacobian(self.h, px, rp.ctypes.data_as(_ip), cj.ctypes.data_as(_ip), val.ctypes.data_as(_dp)))
        return rp, cj, val

    def pc_apply(self, pc_type, rp, cj, val, r):
        a, pa = _i(rp); b, pb = _i(cj); v, pv = _d(val); rr, pr = _d(r)
        z = np.zeros(self.n)
        _chk(self.L.vo_pc_setup_apply(self.h, pc_type, pa, pb, pv, pr, z.ctypes.data_as(_dp)))
        return z

    def ksp_solve(self, ksp_type, pc_type, rp, cj, val, b, rtol=0.0):
        a, pa = _i(rp); c, pc = _i(cj); v, pv = _d(val); bb, pb = _d(b)
        x = np.zeros(self.n)
        its, reason = ct.c_int(), ct.c_int()
        _chk(self.L.vo_ksp_solve(self.h, ksp_type, pc_type, rtol, pa, pc, pv, pb, x.ctypes.data_as(_dp),
                                 ct.byref(its), ct.byref(reason)))
        return x, its.value, reason.value

    # time step level
    def step_dt(self, dt, nstep=1):
        conv, reason = ct.c_int(), ct.c_int()
        _chk(self.L.vo_step_dt(self.h, float(dt), int(nstep), ct.byref(conv), ct.byref(reason)))
        return bool(conv.value), reason.value

    def get(self, auxvar_type, var_type, cond_id=-1, n=None):
        n = self.n if n is None else n
        out = np.zeros(n)
        _chk(self.L.vo_get(self.h, auxvar_type, var_type, cond_id, n, out.ctypes.data_as(_dp)))
        return out

    def counters(self):
        a, b, c, d = ct.c_long(), ct.c_long(), ct.c_long(), ct.c_long()
        self.L.vo_get_counters(self.h, ct.byref(a), ct.byref(b), ct.byref(c), ct.byref(d))
        return dict(newton=a.value, linear=b.value, time_cuts=c.value, func_evals=d.value)


def spmv(rp, cj, val, x):
    a, pa = _i(rp); c, pc = _i(cj); v, pv = _d(val); xx, px = _d(x)
    y = np.zeros(a.size - 1)
    lib().vo_spmv(a.size - 1, pa, pc, pv, px, y.ctypes.data_as(_dp))
    return y
