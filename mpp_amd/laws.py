"""The constitutive laws on the host, array by array: what a driver calls where the reference's drivers call `Density`,
`InternalEnergyAndEnthalpy`, `SatFunc_PressToSat` / `SatFunc_PressToRelPerm` directly to build their inputs
(src/driver/standalone/thermal-e/th_mms_problem.F90:1262-1452 forms manufactured sources that way).  Thin ctypes front of
mppgpu_host_density / _internal_energy_enthalpy / _sat_relperm (include/mppgpu.h): the arithmetic is the library's, the same
inlined functions its kernels use.  Nothing here runs in a solve."""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib


class LawError(RuntimeError):
    pass


def _arr(x, n=None, dtype=np.float64):
    a = np.ascontiguousarray(np.atleast_1d(x), dtype=dtype)
    if n is not None and a.size == 1 and n != 1:
        a = np.full(n, a[0], dtype=dtype)
    return a


def _p(a):
    return a.ctypes.data_as(ct.POINTER(ct.c_double if a.dtype == np.float64 else ct.c_int))


def _chk(rc):
    if rc != 0:
        raise LawError(_lib.lib().mppgpu_last_error(None).decode())


def density(p, t_K, density_type):
    """Density (EOSWaterMod.F90:38-77): den [kmol/m^3], dden_dp, dden_dT"""
    p = _arr(p)
    t = _arr(t_K, p.size)
    den, ddp, ddt = np.empty_like(p), np.empty_like(p), np.empty_like(p)
    _chk(_lib.lib().mppgpu_host_density(int(density_type), p.size, _p(p), _p(t), _p(den), _p(ddp), _p(ddt)))
    return den, ddp, ddt


def internal_energy_enthalpy(p, t_K, ie_type, den, dden_dt, dden_dp):
    """InternalEnergyAndEnthalpy (EOSWaterMod.F90:589-707); den and its derivatives in kg/m^3: U, H [J/kmol], dU_dT, dH_dT,
    dU_dP, dH_dP"""
    p = _arr(p)
    n = p.size
    t, d, ddt, ddp = _arr(t_K, n), _arr(den, n), _arr(dden_dt, n), _arr(dden_dp, n)
    out = [np.empty_like(p) for _ in range(6)]
    _chk(_lib.lib().mppgpu_host_internal_energy_enthalpy(int(ie_type), n, _p(p), _p(t), _p(d), _p(ddt), _p(ddp), *[_p(o) for o in out]))
    return tuple(out)


def sat_relperm(p, satfunc_type, sat_res, alpha, lam):
    """SatFunc_PressToSat + SatFunc_PressToRelPerm (SaturationFunction.F90:564-650): sat, dsat_dP, kr, dkr_dP"""
    p = _arr(p)
    n = p.size
    sf = _arr(satfunc_type, n, np.int32)
    sr, al, la = _arr(sat_res, n), _arr(alpha, n), _arr(lam, n)
    out = [np.empty_like(p) for _ in range(4)]
    _chk(_lib.lib().mppgpu_host_sat_relperm(n, _p(sf), _p(p), _p(sr), _p(al), _p(la), *[_p(o) for o in out]))
    return tuple(out)
