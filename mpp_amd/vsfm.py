"""Host-side mirror of the reference's VSFM operator API over the C ABI (include/mppgpu.h).

Same names, argument meaning and error behaviour as the Fortran types, so a driver written against the
reference reads the same here (compare tests/test_gpu_regression.py with
src/driver/standalone/vsfm/vsfm_celia1990_problem.F90):

  mpp_vsfm_type            src/mpp/mpp/MultiPhysicsProbVSFM.F90:33-52      -> MultiPhysicsProbVSFM
  multiphysicsprob_base    src/mpp/mpp/MultiPhysicsProbBaseType.F90:32-67  -> (same class)
  sysofeqns_vsfm_type      src/mpp/soe/SystemOfEquationsVSFMType.F90       -> SystemOfEquationsVSFM
  sysofeqns_base_type      src/mpp/soe/SystemOfEquationsBaseType.F90:34-93 -> (same class)
  VSFMMPPSetSoils & co.    src/mpp/mpp/MultiPhysicsProbVSFM.F90:211-475, 1115-1434

Everything numerical happens in libmppgpu.so on the GPU; this module only marshals arrays.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib
from . import constants as C
from . import mesh as M


class MPPError(RuntimeError):
    """Raised where the reference would `call endrun(...)` / return a non-zero PETSc ierr."""


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ct.POINTER(ct.c_double))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ct.POINTER(ct.c_int))


class SystemOfEquationsVSFM:
    """sysofeqns_vsfm_type: owns the device-resident solver behind an opaque handle."""

    def __init__(self, device_id=0):
        self.L = _lib.lib()
        self.h = ct.c_void_p()
        rc = self.L.mppgpu_create(int(device_id), ct.byref(self.h))
        if rc:
            raise MPPError(self.L.mppgpu_last_error(None).decode())
        self.mesh = None
        self.ngoveqns = 0
        self._bc_sizes, self._ss_sizes = [], []
        self.time = 0.0
        self.nstep = 0

    def _chk(self, rc):
        if rc:
            raise MPPError(self.L.mppgpu_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.mppgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- set-up ------------------------------------------------------------------------------
    def _set_mesh(self, mesh: M.Mesh):
        self.mesh = mesh
        v, pv = _d(mesh.vol)
        a, pa = _i(mesh.is_active)
        self._chk(self.L.mppgpu_set_mesh(self.h, mesh.ncells_local, mesh.nlev, pv, pa))
        cs = mesh.internal_connections()
        up, pup = _i(cs.id_up); dn, pdn = _i(cs.id_dn)
        du, pdu = _d(cs.dist_up); dd, pdd = _d(cs.dist_dn); ar, par = _d(cs.area); uv, puv = _d(cs.unit_vec)
        self._chk(self.L.mppgpu_set_connections(self.h, cs.num_connections, pup, pdn, pdu, pdd, par, puv))
        if getattr(mesh, "nghost", 0) > 0:   # a rank's piece of a split mesh (mpp_amd.parallel.partition_mesh)
            peer, send_ptr, send_cells, recv_count = mesh.halo
            self._chk(self.L.mppgpu_set_ghost_cells(self.h, int(mesh.nghost)))
            a, pa = _i(peer); b, pb = _i(send_ptr); c, pc = _i(send_cells); d, pd_ = _i(recv_count)
            self._chk(self.L.mppgpu_set_halo(self.h, a.size, pa, pb, pc, pd_))

    def AddConditionInGovEqn(self, igoveqn, ss_or_bc_type, name, unit, cond_type, region_type=None, conn_set=None,
                             swap_order=False):
        """SOEBaseAddConditionInGovEqn (SystemOfEquationsBaseType.F90:1260-1290)."""
        if igoveqn != 1:
            raise MPPError("Attempting to add condition for governing equation that is not in the list")
        if conn_set is None:
            if region_type == C.SOIL_CELLS:
                n = self.mesh.ncells_local
                ids = np.arange(1, n + 1)
                z = np.zeros(n)
                conn_set = M.ConnectionSet(np.full(n, -1), ids, z, z, self.mesh.area_xy.copy(), np.zeros((n, 3)))
            else:
                conn_set = M.create_region_connection_set(self.mesh, region_type)
        dn, pdn = _i(conn_set.id_dn)
        du, pdu = _d(conn_set.dist_up); dd, pdd = _d(conn_set.dist_dn); ar, par = _d(conn_set.area)
        uv, puv = _d(conn_set.unit_vec)
        cid = ct.c_int()
        self._chk(self.L.mppgpu_add_condition(self.h, ss_or_bc_type, cond_type, conn_set.num_connections, pdn, pdu,
                                              pdd, par, puv, int(swap_order), ct.byref(cid)))
        (self._bc_sizes if ss_or_bc_type == C.COND_BC else self._ss_sizes).append(conn_set.num_connections)
        return cid.value

    def SetOption(self, key, value):
        """PETSc options database entry (-ksp_type bcgs, -pc_type bjacobi, -snes_rtol ...)."""
        self._chk(self.L.mppgpu_set_option(self.h, str(key).lstrip("-").encode(), str(value).encode()))

    # ---- time stepping --------------------------------------------------------------------------
    def SetDataFromCLM(self, soe_auxvar_type, var_type, soe_auxvar_id, data_1d):
        """VSFMSOESetDataFromCLM (SystemOfEquationsVSFMType.F90:663-724)."""
        d, pd = _d(data_1d)
        self._chk(self.L.mppgpu_set_data(self.h, soe_auxvar_type, var_type, soe_auxvar_id, d.size, pd))

    def StepDT(self, dt, nstep):
        """SOEBaseStepDT (SystemOfEquationsBaseType.F90:334-552): returns (converged, converged_reason)."""
        conv, reason = ct.c_int(), ct.c_int()
        self._chk(self.L.mppgpu_step_dt(self.h, float(dt), int(nstep), ct.byref(conv), ct.byref(reason)))
        self.nstep = nstep
        self.time = float(dt)
        return bool(conv.value), reason.value

    def GetDataForCLM(self, soe_auxvar_type, var_type, soe_auxvar_id=-1, data_1d=None):
        """VSFMSOEGetDataForCLM (SystemOfEquationsVSFMType.F90:781-845); fills and returns data_1d."""
        if data_1d is None:
            if soe_auxvar_type == C.AUXVAR_INTERNAL:
                n = self.mesh.ncells_local
            elif soe_auxvar_type == C.AUXVAR_BC:
                n = self._bc_sizes[soe_auxvar_id - 1]
            elif soe_auxvar_type == C.AUXVAR_SS:
                n = self._ss_sizes[soe_auxvar_id - 1]
            else:
                n = self.mesh.internal_connections().num_connections
            data_1d = np.zeros(n)
        assert data_1d.dtype == np.float64 and data_1d.flags.c_contiguous
        self._chk(self.L.mppgpu_get_data(self.h, soe_auxvar_type, var_type, soe_auxvar_id, data_1d.size,
                                         data_1d.ctypes.data_as(ct.POINTER(ct.c_double))))
        return data_1d

    def PreStepDT(self):
        """VSFMSPreStepDT (SystemOfEquationsVSFMType.F90:892-923)"""
        self._chk(self.L.mppgpu_pre_step_dt(self.h))

    def PostStepDT(self):
        """VSFMSPostStepDT (:926-940)"""
        self._chk(self.L.mppgpu_post_step_dt(self.h))

    def SetLateralConnections(self, conn_set):
        """mesh%lateral_conn_set_list: connections handled by the explicit lateral-flux operator split"""
        up, pup = _i(conn_set.id_up); dn, pdn = _i(conn_set.id_dn)
        du, pdu = _d(conn_set.dist_up); dd, pdd = _d(conn_set.dist_dn); ar, par = _d(conn_set.area); uv, puv = _d(conn_set.unit_vec)
        self._chk(self.L.mppgpu_set_lateral_connections(self.h, conn_set.num_connections, pup, pdn, pdu, pdd, par, puv))

    def ComputeLateralFlux(self, dt, lateral_ss_id):
        """VSFMComputeLateralFlux (SystemOfEquationsVSFMType.F90:848-889); lateral_ss_id: the 'Lateral_flux' condition"""
        self._chk(self.L.mppgpu_compute_lateral_flux(self.h, float(dt), int(lateral_ss_id)))

    def GetConditionValues(self, kind, cond_id):
        n = (self._bc_sizes if kind == C.COND_BC else self._ss_sizes)[cond_id - 1]
        out = np.zeros(n)
        self._chk(self.L.mppgpu_get_condition_values(self.h, kind, cond_id, n, out.ctypes.data_as(ct.POINTER(ct.c_double))))
        return out

    def step_history(self):
        """sub-steps the last StepDT accepted ([dt] unless the time step was cut)"""
        n = ct.c_int()
        buf = (ct.c_double * 64)()
        self._chk(self.L.mppgpu_get_step_history(self.h, 64, buf, ct.byref(n)))
        return [buf[i] for i in range(min(n.value, 64))]

    def counters(self):
        a, b, c, d = ct.c_longlong(), ct.c_longlong(), ct.c_longlong(), ct.c_longlong()
        self._chk(self.L.mppgpu_get_counters(self.h, ct.byref(a), ct.byref(b), ct.byref(c), ct.byref(d)))
        return dict(newton=a.value, linear=b.value, time_cuts=c.value, func_evals=d.value)

    # ---- operator level (goveqn_base_type / PETSc callbacks) ------------------------------------------
    def SetDtime(self, dt):
        self._chk(self.L.mppgpu_set_dtime(self.h, float(dt)))

    def PreSolve(self):
        self._chk(self.L.mppgpu_pre_solve(self.h))

    def ComputeResidual(self, X):
        """goveqn%ComputeResidual behind SOEResidual (GoveqnRichardsODEPressureType.F90:388-421)."""
        x, px = _d(X)
        f = np.zeros(x.size)
        self._chk(self.L.mppgpu_residual(self.h, px, f.ctypes.data_as(ct.POINTER(ct.c_double))))
        return f

    def ComputeJacobian(self, X):
        """goveqn%ComputeJacobian behind SOEJacobian (:425-453): CSR (rowptr, col, val), 0-based, caller ordering."""
        x, px = _d(X)
        nnz = ct.c_int()
        self._chk(self.L.mppgpu_jacobian_nnz(self.h, ct.byref(nnz)))
        rp = np.zeros(x.size + 1, dtype=np.int32); cj = np.zeros(nnz.value, dtype=np.int32); val = np.zeros(nnz.value)
        self._chk(self.L.mppgpu_jacobian(self.h, px, rp.ctypes.data_as(ct.POINTER(ct.c_int)),
                                         cj.ctypes.data_as(ct.POINTER(ct.c_int)),
                                         val.ctypes.data_as(ct.POINTER(ct.c_double))))
        return rp, cj, val

    def MatMult(self, x):
        xx, px = _d(x)
        y = np.zeros(xx.size)
        self._chk(self.L.mppgpu_spmv(self.h, px, y.ctypes.data_as(ct.POINTER(ct.c_double))))
        return y

    def PCApply(self, r):
        rr, pr = _d(r)
        z = np.zeros(rr.size)
        self._chk(self.L.mppgpu_pc_apply(self.h, pr, z.ctypes.data_as(ct.POINTER(ct.c_double))))
        return z

    def KSPSolve(self, b):
        bb, pb = _d(b)
        x = np.zeros(bb.size)
        its, reason = ct.c_int(), ct.c_int()
        self._chk(self.L.mppgpu_ksp_solve(self.h, pb, x.ctypes.data_as(ct.POINTER(ct.c_double)), ct.byref(its),
                                          ct.byref(reason)))
        return x, its.value, reason.value

    # ---- measurement ----------------------------------------------------------------------------
    def kernel_stats(self, name):
        n, ms, by = ct.c_longlong(), ct.c_double(), ct.c_double()
        self._chk(self.L.mppgpu_kernel_stats(self.h, name.encode(), ct.byref(n), ct.byref(ms), ct.byref(by)))
        calls, stride = ct.c_longlong(), ct.c_int()
        self._chk(self.L.mppgpu_kernel_calls(self.h, name.encode(), ct.byref(calls), ct.byref(stride)))
        return dict(launches=n.value, total_ms=ms.value, bytes_per_launch=by.value, calls=calls.value, stride=stride.value)

    def reset_stats(self):
        self._chk(self.L.mppgpu_reset_stats(self.h))

    def comm_stats(self):
        a, b, c, d = ct.c_longlong(), ct.c_longlong(), ct.c_longlong(), ct.c_longlong()
        self._chk(self.L.mppgpu_get_comm_stats(self.h, ct.byref(a), ct.byref(b), ct.byref(c), ct.byref(d)))
        return dict(allreduces=a.value, halo_exchanges=b.value, chain_fallbacks=c.value, host_readbacks=d.value)

    def halo_stats(self):
        """the halo plan of this rank's piece of a split mesh (mppgpu_get_halo_stats)"""
        v = [ct.c_longlong() for _ in range(5)]
        self._chk(self.L.mppgpu_get_halo_stats(self.h, *[ct.byref(x) for x in v]))
        return dict(zip(("neighbours", "cells_sent", "cells_received", "bytes_received_per_exchange", "max_cells_from_one_neighbour"),
                        (x.value for x in v)))

    def assembly_info(self):
        """layout and kernel selection of this handle (mppgpu_get_assembly_info)"""
        v = (ct.c_int * 14)()
        self._chk(self.L.mppgpu_get_assembly_info(self.h, 14, v))
        keys = ("soil_classes", "level_geometry", "level_porosity_volume", "complete_rows", "column_segments", "spmv_variant",
                "column_solve_in_lds", "speculative_trials_discarded", "ilu_graph_replay", "ilu_graphs_instantiated",
                "setup_structures_verified", "line_search_product_index_free", "line_search_sums_folded", "columns_plus_lateral_assembly")
        return dict(zip(keys, (int(x) for x in v)))

    def pc_info(self):
        """how the ILU(0) triangular solves are applied (mppgpu_get_pc_info): path name, launches per application, box extents"""
        path, n = ct.c_int(), ct.c_int()
        ext = (ct.c_int * 3)()
        self._chk(self.L.mppgpu_get_pc_info(self.h, ct.byref(path), ct.byref(n), ext))
        names = {0: "none", 1: "levels", 2: "chained", 3: "boxes"}
        return dict(path=names.get(path.value, str(path.value)), launches=n.value, box_extents=tuple(ext))

    def bench_kernel(self, name, variant=0, reps=20):
        ms, by = ct.c_double(), ct.c_double()
        self._chk(self.L.mppgpu_bench_kernel(self.h, name.encode(), int(variant), int(reps), ct.byref(ms), ct.byref(by)))
        return ms.value, by.value

    def synchronize(self):
        self._chk(self.L.mppgpu_synchronize(self.h))

    def comm_init(self, nranks, rank, unique_id: bytes):
        buf = ct.create_string_buffer(unique_id, 128)
        self._chk(self.L.mppgpu_comm_init(self.h, int(nranks), int(rank), buf))


def comm_init_local(soes):
    """one global Newton-Krylov solve across several SoEs of this process (one host thread each): mppgpu_comm_init_local"""
    L = _lib.lib()
    arr = (ct.c_void_p * len(soes))(*[s.h for s in soes])
    if L.mppgpu_comm_init_local(arr, len(soes)):
        raise MPPError(L.mppgpu_last_error(soes[0].h).decode())


def comm_get_unique_id() -> bytes:
    buf = ct.create_string_buffer(128)
    L = _lib.lib()
    if L.mppgpu_comm_get_unique_id(buf):
        raise MPPError(L.mppgpu_last_error(None).decode())
    return buf.raw


class MultiPhysicsProbVSFM:
    """mpp_vsfm_type: the call sequence of the reference drivers (celia1990_problem.F90:106-137)."""

    def __init__(self, device_id=0):
        self.name = ""
        self.id = 0
        self.mpi_rank = 0
        self.meshes = []
        self.soe = SystemOfEquationsVSFM(device_id)
        self._setup = False

    # initialize_mpp
    def Init(self):
        return self

    def SetName(self, name):
        self.name = name

    def SetID(self, mpp_id):
        if mpp_id != C.MPP_VSFM_SNES_CLM:
            raise MPPError("VSFMMPPSetup: Unknown mpp_type")
        self.id = mpp_id

    def SetMPIRank(self, rank):
        self.mpi_rank = rank

    # add_meshes
    def SetNumMeshes(self, n):
        if n != 1:
            raise MPPError("only one mesh is supported")

    def AddMesh(self, imesh, mesh: M.Mesh):
        if imesh != 1:
            raise MPPError("mesh id out of range")
        self.meshes = [mesh]

    # add_goveqns
    def AddGovEqn(self, geq_type, name, mesh_itype):
        if geq_type != C.GE_RE:
            raise MPPError("only the Richards equation (GE_RE) is on this path")
        self.soe.ngoveqns += 1
        if self.soe.ngoveqns > 1:
            raise MPPError("a single governing equation is supported")

    def SetMeshesOfGoveqns(self):
        self.soe._set_mesh(self.meshes[0])

    def AllocateAuxVars(self):
        pass  # device arrays are allocated by SetupProblem

    def SetupProblem(self):
        self.soe._chk(self.soe.L.mppgpu_setup(self.soe.h))
        self._setup = True

    def Restart(self, data_1d):
        """VSFMMPPRestart (MultiPhysicsProbVSFM.F90:603-707)."""
        d, pd = _d(data_1d)
        if d.size != self.meshes[0].ncells_local:
            raise MPPError("VSFMMPPRestart: size(data_1d) /= goveq_richards_pres%mesh%ncells_local")
        self.soe._chk(self.soe.L.mppgpu_restart(self.soe.h, d.size, pd))


# ---- soil setters (module-level subroutines in the reference) ---------------------------------------------
def VSFMMPPSetSoilPermeability(vsfm_mpp, igoveqn, perm_x, perm_y, perm_z):
    a, pa = _d(perm_x); b, pb = _d(perm_y); c, pc = _d(perm_z)
    vsfm_mpp.soe._chk(vsfm_mpp.soe.L.mppgpu_set_soil_permeability(vsfm_mpp.soe.h, pa, pb, pc))


def VSFMMPPSetSoilPorosity(vsfm_mpp, igoveqn, por):
    a, pa = _d(por)
    vsfm_mpp.soe._chk(vsfm_mpp.soe.L.mppgpu_set_soil_porosity(vsfm_mpp.soe.h, pa))


def VSFMMPPSetSoilPorosityLinearModel(vsfm_mpp, igoveqn, por_base, pressure_reference, slope):
    """PorosityFunctionSetLinearModel for every cell (src/mpp/util/PorosityFunctionMod.F90:75-96); the reference has
    the model but no MPP-level setter for it."""
    a, pa = _d(por_base); b, pb = _d(pressure_reference); c, pc = _d(slope)
    vsfm_mpp.soe._chk(vsfm_mpp.soe.L.mppgpu_set_soil_porosity_linear(vsfm_mpp.soe.h, pa, pb, pc))


def VSFMMPPSetSaturationFunction(vsfm_mpp, igoveqn, vsfm_satfunc_type, alpha, lam, sat_res):
    t, pt = _i(vsfm_satfunc_type); a, pa = _d(alpha); l, pl = _d(lam); s, ps = _d(sat_res)
    vsfm_mpp.soe._chk(vsfm_mpp.soe.L.mppgpu_set_saturation_function(vsfm_mpp.soe.h, pt, pa, pl, ps))


def VSFMMPPSetDensityType(vsfm_mpp, igoveqn, density_type):
    vsfm_mpp.soe._chk(vsfm_mpp.soe.L.mppgpu_set_density_type(vsfm_mpp.soe.h, int(density_type)))


def VSFMMPPSetSourceSinkAuxVarRealValue(vsfm_mpp, igoveqn, var_type, var_value):
    """VSFMMPPSetSourceSinkAuxVarRealValue (MultiPhysicsProbVSFM.F90:1437-1520): VAR_POT_MASS_SINK_PRESSURE / _EXPONENT of
    the down-regulated sinks, values over the concatenated source-sink connections."""
    v, pv = _d(var_value)
    vsfm_mpp.soe._chk(vsfm_mpp.soe.L.mppgpu_set_source_sink_auxvar(vsfm_mpp.soe.h, int(var_type), v.size, pv))


_SATFUNC_NAMES = {
    "brooks_corey": C.SAT_FUNC_BROOKS_COREY,
    "smooth_brooks_corey_bz2": C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ2,
    "smooth_brooks_corey_bz3": C.SAT_FUNC_SMOOTHED_BROOKS_COREY_BZ3,
    "van_genuchten": C.SAT_FUNC_VAN_GENUCHTEN,
}


def VSFMMPPSetSoils(vsfm_mpp, begc, endc, ncols_ghost, filter_vsfmc, watsat, hksat, bsw, sucsat, residual_sat,
                    vsfm_satfunc_type, density_type):
    """VSFMMPPSetSoilsCLM (MultiPhysicsProbVSFM.F90:249-475): CLM units -> SI, cell icell=(c-begc)*nlevgrnd+j.
    watsat..residual_sat are (ncolumns, nlevgrnd) arrays indexed [c-begc, j-1]."""
    mesh = vsfm_mpp.meshes[0]
    nlev = mesh.nlev
    ncol = endc - begc + 1 + ncols_ghost
    if ncol * nlev != mesh.ncells_local:
        raise MPPError("VSFMMPPSetSoils: column range does not match the mesh")
    filt = np.asarray(filter_vsfmc)
    act = np.nonzero(filt[:endc - begc + 1] == 1)[0]
    if act.size == 0:
        raise MPPError("No active soil hydrology column found")
    col_id = np.where(filt[:ncol] == 1, np.arange(ncol), act[0])

    def pick(a):
        return np.asarray(a, dtype=np.float64).reshape(-1, nlev)[col_id, :].ravel()

    vish2o = 0.001002
    perm = pick(hksat) * vish2o / (C.DENH2O * C.GRAV_CLM) * 0.001   # :374
    alpha = 1.0 / (pick(sucsat) * C.GRAV_CLM)                       # :378
    lam = 1.0 / pick(bsw)                                           # :381
    if vsfm_satfunc_type not in _SATFUNC_NAMES:
        raise MPPError("ERROR:: Unknown vsfm_satfunc_type = " + str(vsfm_satfunc_type))
    n = mesh.ncells_local
    VSFMMPPSetDensityType(vsfm_mpp, 1, density_type)
    VSFMMPPSetSoilPermeability(vsfm_mpp, 1, perm, perm, perm)
    VSFMMPPSetSoilPorosity(vsfm_mpp, 1, pick(watsat))
    VSFMMPPSetSaturationFunction(vsfm_mpp, 1, np.full(n, _SATFUNC_NAMES[vsfm_satfunc_type], dtype=np.int32), alpha, lam,
                                 pick(residual_sat))
