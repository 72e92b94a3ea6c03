"""ctypes loader of libmppgpu.so (the C ABI declared in include/mppgpu.h).

The shared library is built in-tree (mpp_amd/_build/libmppgpu.so) by `build()`; loading fails loudly
when it is missing -- there is no Python or CPU fallback for the compute path.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "_build", "libmppgpu.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mppgpu.h")

_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int)
_llp = ct.POINTER(ct.c_longlong)
_h = ct.c_void_p

# name -> (restype, argtypes): every symbol include/mppgpu.h declares
SIGNATURES = {
    "mppgpu_create": (ct.c_int, [ct.c_int, ct.POINTER(_h)]),
    "mppgpu_destroy": (ct.c_int, [_h]),
    "mppgpu_last_error": (ct.c_char_p, [_h]),
    "mppgpu_synchronize": (ct.c_int, [_h]),
    "mppgpu_set_mesh": (ct.c_int, [_h, ct.c_int, ct.c_int, _dp, _ip]),
    "mppgpu_set_connections": (ct.c_int, [_h, ct.c_int, _ip, _ip, _dp, _dp, _dp, _dp]),
    "mppgpu_add_condition": (ct.c_int, [_h, ct.c_int, ct.c_int, ct.c_int, _ip, _dp, _dp, _dp, _dp, ct.c_int, _ip]),
    "mppgpu_set_soil_permeability": (ct.c_int, [_h, _dp, _dp, _dp]),
    "mppgpu_set_soil_porosity": (ct.c_int, [_h, _dp]),
    "mppgpu_set_soil_porosity_linear": (ct.c_int, [_h, _dp, _dp, _dp]),
    "mppgpu_set_saturation_function": (ct.c_int, [_h, _ip, _dp, _dp, _dp]),
    "mppgpu_set_density_type": (ct.c_int, [_h, ct.c_int]),
    "mppgpu_pre_step_dt": (ct.c_int, [_h]),
    "mppgpu_post_step_dt": (ct.c_int, [_h]),
    "mppgpu_set_ghost_cells": (ct.c_int, [_h, ct.c_int]),
    "mppgpu_set_halo": (ct.c_int, [_h, ct.c_int, _ip, _ip, _ip, _ip]),
    "mppgpu_set_lateral_connections": (ct.c_int, [_h, ct.c_int, _ip, _ip, _dp, _dp, _dp, _dp]),
    "mppgpu_compute_lateral_flux": (ct.c_int, [_h, ct.c_double, ct.c_int]),
    "mppgpu_get_condition_values": (ct.c_int, [_h, ct.c_int, ct.c_int, ct.c_int, _dp]),
    "mppgpu_set_source_sink_auxvar": (ct.c_int, [_h, ct.c_int, ct.c_int, _dp]),
    "mppgpu_set_option": (ct.c_int, [_h, ct.c_char_p, ct.c_char_p]),
    "mppgpu_setup": (ct.c_int, [_h]),
    "mppgpu_restart": (ct.c_int, [_h, ct.c_int, _dp]),
    "mppgpu_set_data": (ct.c_int, [_h, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _dp]),
    "mppgpu_step_dt": (ct.c_int, [_h, ct.c_double, ct.c_int, _ip, _ip]),
    "mppgpu_get_data": (ct.c_int, [_h, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _dp]),
    "mppgpu_get_step_history": (ct.c_int, [_h, ct.c_int, _dp, _ip]),
    "mppgpu_get_counters": (ct.c_int, [_h, _llp, _llp, _llp, _llp]),
    "mppgpu_set_dtime": (ct.c_int, [_h, ct.c_double]),
    "mppgpu_pre_solve": (ct.c_int, [_h]),
    "mppgpu_residual": (ct.c_int, [_h, _dp, _dp]),
    "mppgpu_jacobian_nnz": (ct.c_int, [_h, _ip]),
    "mppgpu_jacobian": (ct.c_int, [_h, _dp, _ip, _ip, _dp]),
    "mppgpu_spmv": (ct.c_int, [_h, _dp, _dp]),
    "mppgpu_pc_apply": (ct.c_int, [_h, _dp, _dp]),
    "mppgpu_ksp_solve": (ct.c_int, [_h, _dp, _dp, _ip, _ip]),
    "mppgpu_thermal_set_soils": (ct.c_int, [_h, _ip, _ip, _dp, _dp, _dp, _dp, _dp]),
    "mppgpu_thermal_set_soln_prev_clm": (ct.c_int, [_h, ct.c_int, _dp]),
    "mppgpu_thermal_pre_step_dt": (ct.c_int, [_h]),
    "mppgpu_thermal_step_dt": (ct.c_int, [_h, ct.c_double, _ip, _ip]),
    "mppgpu_thermal_get_soln": (ct.c_int, [_h, ct.c_int, _dp]),
    "mppgpu_thermal_compute_rhs": (ct.c_int, [_h, ct.c_double, _dp]),
    "mppgpu_thermal_compute_operators": (ct.c_int, [_h, ct.c_double, _ip, _ip, _dp]),
    "mppgpu_th_add_condition": (ct.c_int, [_h, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _ip, _dp, _dp, _dp, _dp, ct.c_int, _ip]),
    "mppgpu_th_set_soils": (ct.c_int, [_h, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "mppgpu_kernel_stats": (ct.c_int, [_h, ct.c_char_p, _llp, _dp, _dp]),
    "mppgpu_kernel_calls": (ct.c_int, [_h, ct.c_char_p, _llp, _ip]),
    "mppgpu_reset_stats": (ct.c_int, [_h]),
    "mppgpu_get_comm_stats": (ct.c_int, [_h, _llp, _llp, _llp, _llp]),
    "mppgpu_get_halo_stats": (ct.c_int, [_h, _llp, _llp, _llp, _llp, _llp]),
    "mppgpu_get_pc_info": (ct.c_int, [_h, _ip, _ip, _ip]),
    "mppgpu_get_assembly_info": (ct.c_int, [_h, ct.c_int, _ip]),
    "mppgpu_comm_runtime_version": (ct.c_int, [_ip]),
    "mppgpu_host_density": (ct.c_int, [ct.c_int, ct.c_int, _dp, _dp, _dp, _dp, _dp]),
    "mppgpu_host_internal_energy_enthalpy": (ct.c_int, [ct.c_int, ct.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "mppgpu_host_sat_relperm": (ct.c_int, [ct.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "mppgpu_bench_kernel": (ct.c_int, [_h, ct.c_char_p, ct.c_int, ct.c_int, _dp, _dp]),
    "mppgpu_comm_get_unique_id": (ct.c_int, [ct.c_void_p]),
    "mppgpu_comm_init": (ct.c_int, [_h, ct.c_int, ct.c_int, ct.c_void_p]),
    "mppgpu_comm_init_local": (ct.c_int, [ct.POINTER(_h), ct.c_int]),
}


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-s"]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libmppgpu.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`. "
                "The VSFM solve path has no CPU fallback." % SO_PATH)
        L = ct.CDLL(SO_PATH, mode=ct.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib
