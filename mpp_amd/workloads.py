"""Workload builders of the VSFM / thermal / coupled-TH solve path: inputs only, numbers quoted from the reference drivers.

celia1990 : src/driver/standalone/vsfm/vsfm_celia1990_problem.F90 (nz=100, dt=3600 s x 24, VG, TGDPB01,
            Dirichlet top 93991 Pa / bottom 3535.5 Pa, IC 3535.5 Pa); `ncols` copies = the "celia-extruded" grid of
            SURVEY.md section 8(d), which bench.py times
vchannel  : src/driver/standalone/vsfm/vsfm_vchannel_problem.F90 (20x10x30, 7-pt, no BC, dt=8640 s x 3)
thermal_mms / mass_and_heat / th_mms / th_columns : the thermal and thermal-hydrology drivers (see each function)

`to_gpu` / `to_thermal_gpu` / `to_th_gpu` build a problem on the GPU through the reference-shaped host API
(mpp_amd.vsfm / thermal / th).  Nothing here touches the CPU checker used by the tests (its adapters live beside it, outside this package).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import constants as C
from . import mesh as M


@dataclass
class Problem:
    name: str
    mesh: M.Mesh
    perm: np.ndarray            # (n,3)
    por: np.ndarray
    satfunc_type: np.ndarray
    alpha: np.ndarray
    lam: np.ndarray
    sat_res: np.ndarray
    density_type: int
    press_ic: np.ndarray
    bcs: list = field(default_factory=list)   # [(cond_type, ConnectionSet, values)]
    sss: list = field(default_factory=list)
    lateral: object = None                    # ConnectionSet of the explicit (operator-split) lateral connections
    por_pressure_reference: np.ndarray = None  # linear porosity model (PorosityFunctionMod.F90:75-96): por is then porosity_base
    por_slope: np.ndarray = None
    sink_pressure: np.ndarray = None          # down-regulated sinks: over the concatenated source-sink connections
    sink_exponent: np.ndarray = None
    dt: float = 3600.0
    nstep: int = 1
    exact: np.ndarray = None                  # manufactured solution at the cell centres (vsfm_mms)


def clm_soil_conversion(perm, alpha, lam, sucsat_grav):
    """The CLM-unit round trip every reference driver does before VSFMMPPSetSoils, then
    VSFMMPPSetSoilsCLM's inverse (MultiPhysicsProbVSFM.F90:374-381): two different gravity constants and
    two different viscosities are involved, so the numbers are reproduced step by step."""
    vish2o = 0.001002
    hksat = perm / vish2o * (C.DENH2O * C.GRAV_CLM) / 0.001          # celia1990_problem.F90:325
    bsw = 1.0 / lam
    sucsat = 1.0 / (alpha * sucsat_grav)                             # :327 uses GRAVITY_CONSTANT; vchannel uses grav
    perm_eff = hksat * vish2o / (C.DENH2O * C.GRAV_CLM) * 0.001      # MultiPhysicsProbVSFM.F90:374
    alpha_eff = 1.0 / (sucsat * C.GRAV_CLM)                          # :378
    lam_eff = 1.0 / bsw                                              # :381
    return perm_eff, alpha_eff, lam_eff


def celia1990(ncols=1, nz=100, perm_mult=None) -> Problem:
    """`ncols` copies of the celia1990 column on a CLM-style (z-fastest) mesh: the 'celia-extruded' grid of
    SURVEY.md section 8(d).  perm_mult: optional per-column permeability multiplier."""
    z_column = 1.0
    m = M.mesh_create_soil_columns("Soil mesh", ncols, nz, z_column / nz, area=1.0)
    n = m.ncells_local
    perm, alpha, lam = clm_soil_conversion(8.3913e-12, 3.4257e-4, 0.5, C.GRAVITY_CONSTANT)
    perm_arr = np.full((n, 3), perm)
    if perm_mult is not None:
        perm_arr = perm_arr * np.repeat(np.asarray(perm_mult, dtype=np.float64), nz)[:, None]
    top = M.create_region_connection_set(m, C.SOIL_TOP_CELLS)
    bot = M.create_region_connection_set(m, C.SOIL_BOTTOM_CELLS)
    return Problem(
        name="vsfm_celia1990", mesh=m, perm=perm_arr, por=np.full(n, 0.368),
        satfunc_type=np.full(n, C.SAT_FUNC_VAN_GENUCHTEN, dtype=np.int32), alpha=np.full(n, alpha),
        lam=np.full(n, lam), sat_res=np.full(n, 0.2772), density_type=C.DENSITY_TGDPB01,
        press_ic=np.full(n, 3.5355e3),
        bcs=[(C.COND_DIRICHLET, top, np.full(ncols, 9.3991e4)), (C.COND_DIRICHLET, bot, np.full(ncols, 3.5355e3))],
        dt=3600.0, nstep=24)


def _vchannel_surface(nx, ny):
    """cell-centre coordinates of the v-channel's columns (vsfm_vchannel_problem.F90: a valley along y, sloping in x towards the
    axis and in y downstream): xc2, yc2, zc2 shaped (nx, ny)"""
    dx, dy, slope_x, slope_y = 10.0, 10.0, 0.1, 0.2
    zv_x = np.zeros(nx + 1)
    for ii in range(1, nx // 2 + 2):
        zv_x[ii - 1] = slope_x * dx * (nx // 2) - (ii - 1) * slope_x * dx
    for ii in range(nx // 2 + 2, nx + 2):
        zv_x[ii - 1] = (ii - nx // 2 - 1) * slope_x * dx
    zv_y = (np.arange(1, ny + 2) - 1) * slope_y * dy
    zv = zv_x[:, None] + zv_y[None, :]                      # (nx+1, ny+1)
    xv = ((np.arange(1, nx + 2) - 1) * dx)[:, None] + np.zeros((1, ny + 1))
    yv = np.zeros((nx + 1, 1)) + ((np.arange(1, ny + 2) - 1) * dy)[None, :]

    def c4(a):
        return (a[:-1, :-1] + a[1:, :-1] + a[:-1, 1:] + a[1:, 1:]) / 4.0

    return c4(xv), c4(yv), c4(zv)


def _vchannel_block(nx, ny, nz, rows, nown_rows):
    """The v-channel mesh restricted to the grid rows `rows` (0-based jj; every ii and kk of a row): the first `nown_rows` rows
    are owned, the others are ghost rows.  Local cells: owned cells in ascending global id (kk, jj, ii), then each ghost row in
    ascending global id (kk, ii).  Internal connections with at least one owned end, in the driver's global order (vertical: kk,
    ii, jj; x: ii, kk, jj; y: jj, kk, ii), geometry by the driver's expressions in the GLOBAL indices.  With rows = all rows this
    is the driver's mesh.  Returns (mesh, l2g [0-based global ids], number of owned cells)."""
    dx, dy, dz = 10.0, 10.0, 0.5
    rows = np.asarray(rows, dtype=np.int64)
    own_rows, ghost_rows = rows[:nown_rows], rows[nown_rows:]
    xc2, yc2, zc2 = _vchannel_surface(nx, ny)
    nxy = nx * ny
    kk = np.arange(nz, dtype=np.int64)
    ii = np.arange(nx, dtype=np.int64)
    gid_own = (kk[:, None, None] * nxy + own_rows[None, :, None] * nx + ii[None, None, :])            # [kk, jj_own, ii]
    l2g = [gid_own.ravel()] + [(kk[:, None] * nxy + g * nx + ii[None, :]).ravel() for g in ghost_rows]
    l2g = np.concatenate(l2g)
    nloc, nown = l2g.size, gid_own.size
    gk, grem = np.divmod(l2g, nxy)
    gj, gi = np.divmod(grem, nx)
    m = M.Mesh(name="Soil mesh", ncells_local=int(nloc), nlev=nz, orientation=C.MESH_AGAINST_GRAVITY)
    m.x, m.y = xc2[gi, gj].copy(), yc2[gi, gj].copy()
    m.z = zc2[gi, gj] - (dz / 2.0 + (nz - (gk + 1)) * dz)
    m.dx, m.dy, m.dz = np.full(nloc, dx), np.full(nloc, dy), np.full(nloc, dz)
    m.area_xy = np.full(nloc, dx * dy)
    m.is_active = np.ones(nloc, dtype=np.int32)
    m.compute_volume()
    # local 1-based id of the cell (kk, row position p, ii); row position p indexes `rows`
    lid = np.empty((nz, rows.size, nx), dtype=np.int64)
    lid[:, :nown_rows, :] = (np.arange(nown, dtype=np.int64) + 1).reshape(nz, nown_rows, nx)
    for g in range(ghost_rows.size):
        lid[:, nown_rows + g, :] = nown + g * nz * nx + (np.arange(nz * nx, dtype=np.int64) + 1).reshape(nz, nx)
    pos = {int(j): p for p, j in enumerate(rows)}
    own_sorted = np.sort(own_rows)
    op = [pos[int(j)] for j in own_sorted]                    # owned rows in ascending jj
    ups, dns, dus, dds, areas = [], [], [], [], []

    def dist_of(u, d):
        ddx = m.x[u - 1] - m.x[d - 1]; ddy = m.y[u - 1] - m.y[d - 1]; ddz = m.z[u - 1] - m.z[d - 1]
        return (ddx ** 2.0 + ddy ** 2.0 + ddz ** 2.0) ** 0.5

    # vertical: do kk, ii, jj (owned rows only: a connection between two ghost cells is not kept)
    L = lid[:, op, :]
    up = L[:-1].transpose(0, 2, 1).ravel(); dn = L[1:].transpose(0, 2, 1).ravel()
    ups.append(up); dns.append(dn)
    dus.append(np.full(up.size, 0.5 * dz)); dds.append(np.full(up.size, 0.5 * dz)); areas.append(np.full(up.size, dx * dy))
    # horizontal x: do ii, kk, jj
    up = L[:, :, :-1].transpose(2, 0, 1).ravel(); dn = L[:, :, 1:].transpose(2, 0, 1).ravel()
    dist = dist_of(up, dn)
    ups.append(up); dns.append(dn); dus.append(0.5 * dist); dds.append(0.5 * dist); areas.append(np.full(up.size, dz * dy))
    # horizontal y: do jj, kk, ii -- pairs (jj, jj + 1) of present rows with at least one owned
    owned = set(int(j) for j in own_rows)
    pairs = [j for j in range(ny - 1) if j in pos and (j + 1) in pos and (j in owned or (j + 1) in owned)]
    if pairs:
        up = np.stack([lid[:, pos[j], :] for j in pairs]).ravel(); dn = np.stack([lid[:, pos[j + 1], :] for j in pairs]).ravel()
        dist = dist_of(up, dn)
        ups.append(up); dns.append(dn); dus.append(0.5 * dist); dds.append(0.5 * dist); areas.append(np.full(up.size, dz * dx))
    up, dn = np.concatenate(ups), np.concatenate(dns)
    cs = M.ConnectionSet(up, dn, np.concatenate(dus), np.concatenate(dds), np.concatenate(areas), None, None)
    cs.unit_vec = M.unit_vec_from_centroids(m, up, dn)
    m.intrn_conn_sets.append(cs)
    m.ncols, m.ordering = nxy, "x-fastest"
    return m, l2g, int(nown)


def _vchannel_soil(m, l2g, nx, ny, nz) -> Problem:
    n = m.ncells_local
    perm, alpha, lam = clm_soil_conversion(8.3913e-12, 3.4257e-4, 0.5, C.GRAV_CLM)
    jj = l2g // (nx * ny) + 1                                    # the level (the driver's loop variable is called jj)
    press_ic = (18.75 - 0.5 * (jj - 1) - 2.0) * 997.18 * 9.8 + 101325.0
    return Problem(
        name="vsfm_vchannel", mesh=m, perm=np.full((n, 3), perm), por=np.full(n, 0.368),
        satfunc_type=np.full(n, C.SAT_FUNC_VAN_GENUCHTEN, dtype=np.int32), alpha=np.full(n, alpha),
        lam=np.full(n, lam), sat_res=np.full(n, 0.2772), density_type=C.DENSITY_TGDPB01, press_ic=press_ic,
        dt=8640.0, nstep=3)


def vchannel(nx=20, ny=10, nz=30) -> Problem:
    """src/driver/standalone/vsfm/vsfm_vchannel_problem.F90 (20 x 10 x 30, 7-point, no boundary condition, dt = 8640 s x 3)"""
    m, l2g, _ = _vchannel_block(nx, ny, nz, np.arange(ny), ny)
    return _vchannel_soil(m, l2g, nx, ny, nz)


def _rank_rows(ny, world, rank):
    """grid rows of rank `rank` when the nx x ny columns are cut into `world` blocks of whole rows: (owned rows, ghost rows grouped by
    owner [lower neighbour's row first], peers)"""
    npl = ny // world
    j0 = rank * npl
    own = np.arange(j0, j0 + npl)
    ghosts, peers = [], []
    if rank > 0:
        ghosts.append(j0 - 1); peers.append(rank - 1)
    if rank < world - 1:
        ghosts.append(j0 + npl); peers.append(rank + 1)
    return own, np.asarray(ghosts, dtype=np.int64), peers


def _block_part(m, l2g, nown, nx, nz, own_rows, peers, rank):
    """MeshPart of a block of whole grid rows built directly: each neighbour receives my row next to it -- nz x nx cells, ascending
    global id -- and sends me as many"""
    from mpp_amd import parallel as PAR
    nrows = own_rows.size
    send_cells, send_ptr = [], [0]
    base = (np.arange(nz, dtype=np.int64)[:, None] * nrows * nx + np.arange(nx, dtype=np.int64)[None, :])
    for q in peers:
        p = 0 if q < rank else nrows - 1          # my row next to that neighbour
        send_cells.append((base + p * nx).ravel() + 1)
        send_ptr.append(send_ptr[-1] + nz * nx)
    m.nghost = int(l2g.size - nown)
    m.halo = (np.asarray(peers, dtype=np.int32), np.asarray(send_ptr, dtype=np.int32),
              np.concatenate(send_cells).astype(np.int32) if send_cells else np.zeros(0, np.int32),
              np.full(len(peers), nz * nx, dtype=np.int32))
    part = PAR.MeshPart(m, l2g, m.nghost, *m.halo)
    part.mode, part.ncols_owned = "columns", int(nrows * nx)
    return part


def vchannel_rank(nx, ny, nz, world, rank):
    """Rank `rank`'s share of vchannel(nx, ny, nz) split over `world` ranks by whole soil columns (mpp_amd.parallel.partition_mesh's
    default: x-y blocks of the column set) -- built WITHOUT the global problem when the blocks are whole grid rows (ny a multiple of
    world), which is what a rank process of `bench.py --workload vsfm3d --gpus N` calls.  Same cells, order, connections, halo plan
    as split_problem(vchannel(...), world)[rank] (tests/test_parallel_gloo.py compares the two array by array);
    returns (part, local problem)."""
    from mpp_amd import parallel as PAR
    if world == 1:
        prob = vchannel(nx, ny, nz)
        n = prob.mesh.ncells_local
        part = PAR.MeshPart(prob.mesh, np.arange(n), 0, np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
        part.mode, part.ncols_owned = "columns", nx * ny
        return part, prob
    if ny % world != 0:
        return split_problem(vchannel(nx, ny, nz), world)[rank]
    own, ghosts, peers = _rank_rows(ny, world, rank)
    m, l2g, nown = _vchannel_block(nx, ny, nz, np.concatenate([own, ghosts]), own.size)
    part = _block_part(m, l2g, nown, nx, nz, own, peers, rank)
    return part, _vchannel_soil(m, l2g, nx, ny, nz)


def vchannel_operator_split(nx=6, ny=5, nz=7) -> Problem:
    """the v-channel grid in ELM's operator-split form: only the vertical connections are implicit (mesh%intrn_conn_set_list),
    the horizontal ones form mesh%lateral_conn_set_list and enter through the 'Lateral_flux' mass-rate source
    (VSFMComputeLateralFlux, SystemOfEquationsVSFMType.F90:848-889)"""
    prob = vchannel(nx, ny, nz)
    cs = prob.mesh.intrn_conn_sets[0]
    nv = nx * ny * (nz - 1)
    pick = lambda a, sl: None if a is None else a[sl]
    vert = M.ConnectionSet(cs.id_up[:nv], cs.id_dn[:nv], cs.dist_up[:nv], cs.dist_dn[:nv], cs.area[:nv], cs.unit_vec[:nv], pick(cs.itype, slice(0, nv)))
    lat = M.ConnectionSet(cs.id_up[nv:], cs.id_dn[nv:], cs.dist_up[nv:], cs.dist_dn[nv:], cs.area[nv:], cs.unit_vec[nv:], pick(cs.itype, slice(nv, None)))
    prob.mesh.intrn_conn_sets = [vert]
    prob.lateral = lat
    n = prob.mesh.ncells_local
    z = np.zeros(n)
    allcells = M.ConnectionSet(np.full(n, -1), np.arange(1, n + 1), z, z, np.ones(n), np.zeros((n, 3)))
    prob.sss = [(C.COND_MASS_RATE, allcells, np.zeros(n))]
    prob.name = "vsfm_vchannel_operator_split"
    return prob


def vsfm_mms(nx=20, fully_saturated=False) -> Problem:
    """src/driver/standalone/vsfm/vsfm_mms_problem.F90 + vsfm_mms_steady_state_soil_only_1D.F90: the reference's analytic check of
    the VSFM that does not depend on PETSc -- a steady 1-D horizontal column x in [0, 10] m of nx cells (ny = nz = 1, so no gravity
    term along the connections), porosity 0 (no accumulation), manufactured pressure
    P(x) = 15000 sin(pi x / 10) - 20000 + 101325 (:52-60; `fully_saturated`: + 20000), permeability
    k(x) = 1e-11 (2 + cos(pi x / 10)) (:74-75), van Genuchten alpha = 1/4000, m = 1/2, residual saturation 1/2 (:89-121),
    DENSITY_TGDPB01, Dirichlet ends at x = 0 and 10 (ComputeBoundaryDomainConnection: dist_up 0, dist_dn dx/2, unit vectors +-x),
    mass source = -d/dx(rho k kr / mu dP/dx) dx per cell (DATA_MASS_SOURCE :263-302, the constitutive laws called on the host as
    the driver calls Density / SatFunc_* there), start = the mean of the manufactured pressures (:196-203), one StepDT(1.0)
    (vsfm_mms_problem.F90:97-102).  The per-cell-array setters VSFMMPPSetSoilPorosity / Permeability / SaturationFunction /
    DensityType (:405-418) are what to_gpu calls.  `exact` = P at the cell centres (DATA_PRESSURE)."""
    from mpp_amd import laws as L
    pi = 4.0 * np.arctan(1.0)
    x_min, x_max = 0.0, 10.0
    xlim = x_max - x_min
    m = M.mesh_create("Soil mesh", x_max, 1.0, 1.0, nx, 1, 1, C.CONN_IN_XYZ_DIR, x_min=x_min)
    n = m.ncells_local
    dx = (x_max - x_min) / nx
    xc = np.asarray(m.x)
    a0, a1 = 15000.0, -20000.0

    def pres(x):      # compute_pressure_or_deriv :43-63
        val = a0 * np.sin((x - x_min) / xlim * pi) + (-a1 if fully_saturated else a1) + C.PRESSURE_REF
        return val, a0 * pi / xlim * np.cos((x - x_min) / xlim * pi), -a0 * pi * pi / xlim / xlim * np.sin((x - x_min) / xlim * pi)

    def permf(x):     # compute_permeability_or_deriv :66-79
        p0 = 1e-11
        return p0 * (2.0 + np.cos((x - x_min) / xlim * pi)), p0 * pi / xlim * (-np.sin((x - x_min) / xlim * pi))

    alpha0, lam0, sr0 = 1.0 / 4000.0, 0.5, 0.5
    mu = 8.904156e-4                                                       # Viscosity (EOSWaterMod.F90:582)
    k, dk_dx = permf(xc)
    P, dP_dx, d2P_dx2 = pres(xc)
    den, dden_dP, _ = L.density(P, 298.15, C.DENSITY_TGDPB01)
    rho, drho_dP = den * C.FMWH2O, dden_dP * C.FMWH2O
    _se, _dse_dP, kr, dkr_dP = L.sat_relperm(P, C.SAT_FUNC_VAN_GENUCHTEN, sr0, alpha0, lam0)
    dkr_dx = dkr_dP * dP_dx                                                # (dse_dp0 = 0: alpha does not vary)
    drho_dx = drho_dP * dP_dx
    src = (-((k * kr / mu) * drho_dx + (rho * kr / mu) * dk_dx + (rho * k / mu) * dkr_dx) * dP_dx - (rho * k * kr / mu) * d2P_dx2) * dx
    P0 = 0.0
    for i in range(n):                                                     # DATA_INITIAL_PRESSURE: running mean, same summation order
        P0 = P0 + 1.0 / nx * P[i]
    pres_bc = np.array([pres(xc[0] - dx / 2.0)[0], pres(xc[-1] + dx / 2.0)[0]])
    ends = M.ConnectionSet(np.array([0, 0]), np.array([1, nx]), np.zeros(2), np.full(2, dx / 2.0), np.full(2, 1.0 * 1.0),
                           np.array([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]]))      # ComputeXBoundaryDomainConnection (mpp_mesh_utils.F90:783-812)
    z = np.zeros(n)
    allcells = M.ConnectionSet(np.full(n, -1), np.arange(1, n + 1), z, z, np.ones(n), np.zeros((n, 3)))
    prob = Problem(
        name="vsfm_mms", mesh=m, perm=np.stack([k] * 3, axis=1), por=np.zeros(n),
        satfunc_type=np.full(n, C.SAT_FUNC_VAN_GENUCHTEN, dtype=np.int32), alpha=np.full(n, alpha0), lam=np.full(n, lam0),
        sat_res=np.full(n, sr0), density_type=C.DENSITY_TGDPB01, press_ic=np.full(n, P0),
        bcs=[(C.COND_DIRICHLET, ends, pres_bc)], sss=[(C.COND_MASS_RATE, allcells, src)], dt=1.0, nstep=1)
    prob.exact = P
    return prob


def regression_stats(data, num_cells=5):
    """min / max / mean / every k-th cell, as regression_mod%WriteData (src/driver/standalone/util/regression.F90:76-124)."""
    data = np.asarray(data)
    inc = data.size // min(num_cells, data.size)
    cells = list(range(1, data.size + 1, inc))
    return dict(min=float(data.min()), max=float(data.max()), mean=float(data.sum() / data.size),
                cells={c: float(data[c - 1]) for c in cells})


def to_gpu(prob: Problem, options=None, device_id=0):
    """Build the problem on the GPU through the reference-shaped host API (mpp_amd.vsfm), following the
    call order of the reference drivers (vsfm_celia1990_problem.F90:106-137)."""
    from mpp_amd import vsfm as V
    mpp = V.MultiPhysicsProbVSFM(device_id)
    mpp.Init()
    mpp.SetName("Variably-Saturated-Flow-Model")
    mpp.SetID(C.MPP_VSFM_SNES_CLM)
    mpp.SetMPIRank(0)
    mpp.SetNumMeshes(1)
    mpp.AddMesh(1, prob.mesh)
    mpp.AddGovEqn(C.GE_RE, "Richards Equation ODE", C.MESH_CLM_SOIL_COL)
    mpp.SetMeshesOfGoveqns()
    for cond_type, cs, _vals in prob.bcs:
        mpp.soe.AddConditionInGovEqn(1, C.COND_BC, "bc", "Pa", cond_type, conn_set=cs)
    for cond_type, cs, _vals in prob.sss:
        mpp.soe.AddConditionInGovEqn(1, C.COND_SS, "ss", "kg/s", cond_type, conn_set=cs)
    for k, v in (options or {}).items():
        mpp.soe.SetOption(k, v)
    mpp.AllocateAuxVars()
    mpp.SetupProblem()
    V.VSFMMPPSetSoilPermeability(mpp, 1, prob.perm[:, 0], prob.perm[:, 1], prob.perm[:, 2])
    if prob.por_slope is not None:
        V.VSFMMPPSetSoilPorosityLinearModel(mpp, 1, prob.por, prob.por_pressure_reference, prob.por_slope)
    else:
        V.VSFMMPPSetSoilPorosity(mpp, 1, prob.por)
    V.VSFMMPPSetSaturationFunction(mpp, 1, prob.satfunc_type, prob.alpha, prob.lam, prob.sat_res)
    V.VSFMMPPSetDensityType(mpp, 1, prob.density_type)
    for i, (_t, _cs, vals) in enumerate(prob.bcs):
        mpp.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, i + 1, vals)
    for i, (_t, _cs, vals) in enumerate(prob.sss):
        mpp.soe.SetDataFromCLM(C.AUXVAR_SS, C.VAR_BC_SS_CONDITION, i + 1, vals)
    if prob.lateral is not None:
        mpp.soe.SetLateralConnections(prob.lateral)
    if prob.sink_pressure is not None:
        V.VSFMMPPSetSourceSinkAuxVarRealValue(mpp, 1, C.VAR_POT_MASS_SINK_PRESSURE, prob.sink_pressure)
        V.VSFMMPPSetSourceSinkAuxVarRealValue(mpp, 1, C.VAR_POT_MASS_SINK_EXPONENT, prob.sink_exponent)
    mpp.Restart(prob.press_ic)
    return mpp


# ------------------------------------------------------------------------------------------------------------
# thermal MMS problems (src/driver/standalone/thermal/thermal_mms_problem.F90 and
# thermal_mms_steady_state_problem_{1D,2D,3D}.F90): steady heat conduction on the unit cube, Dirichlet faces,
# manufactured source; cnfac = 0, csol = 0, one StepDT(1.0)
# ------------------------------------------------------------------------------------------------------------
@dataclass
class ThermalProblem:
    name: str
    mesh: M.Mesh
    tkdry: np.ndarray          # per cell, in the cell numbering the reference ends up using (see below)
    bc: M.ConnectionSet
    bc_temperature: np.ndarray
    heat_source: np.ndarray
    exact: np.ndarray
    t_init: float = 290.0
    cnfac: float = 0.0
    dt: float = 1.0


def _mms_funcs(ptype):
    pi = 4.0 * np.arctan(1.0)
    if ptype == 1:
        T = lambda x, y, z: 10 * np.sin(pi * x) + 270.0
        lam = lambda x, y, z: np.exp(x)
        src = lambda x, y, z: -np.exp(x) * (10.0 * pi * np.cos(pi * x)) - np.exp(x) * (-10.0 * pi * pi * np.sin(pi * x))
    elif ptype == 2:
        T = lambda x, y, z: 10.0 * np.sin(x * pi) * np.cos(2.0 * y * pi) + 270.0
        lam = lambda x, y, z: (x + 0.5) * np.exp(x + y - 1.0)

        def src(x, y, z):
            l = (x + 0.5) * np.exp(x + y - 1.0)
            dl_dx = ((x + 0.5) + 1.0) * np.exp(x + y - 1.0)
            dl_dy = (x + 0.5) * np.exp(x + y - 1.0)
            dT_dx = 10.0 * pi * np.cos(x * pi) * np.cos(2.0 * pi * y)
            dT_dy = -20.0 * pi * np.sin(x * pi) * np.sin(2.0 * pi * y)
            d2x = -10.0 * pi * pi * np.sin(x * pi) * np.cos(2.0 * pi * y)
            d2y = -40.0 * pi * pi * np.sin(x * pi) * np.cos(2.0 * pi * y)
            return -dl_dx * dT_dx - l * d2x - dl_dy * dT_dy - l * d2y
    else:
        T = lambda x, y, z: 10.0 * np.sin(x * pi) * np.cos(2.0 * pi * y) * np.sin(3.0 * pi * z) + 270.0
        lam = lambda x, y, z: np.exp(x + y + z - 1.0)

        def src(x, y, z):
            a = np.exp(x + y + z - 1.0)
            sx, cx = np.sin(x * pi), np.cos(x * pi)
            sy, cy = np.sin(2.0 * pi * y), np.cos(2.0 * pi * y)
            sz, cz = np.sin(3.0 * pi * z), np.cos(3.0 * pi * z)
            dT_dx, dT_dy, dT_dz = 10.0 * pi * cx * cy * sz, -20.0 * pi * sx * sy * sz, 30.0 * pi * sx * cy * cz
            d2x, d2y, d2z = -10.0 * pi * pi * sx * cy * sz, -40.0 * pi * pi * sx * cy * sz, -90.0 * pi * pi * sx * cy * sz
            return -a * dT_dx - a * d2x - a * dT_dy - a * d2y - a * dT_dz - a * d2z
    return T, lam, src


def thermal_mms(ptype=3, nx=3, ny=3, nz=3, replicate_reference_soil_indexing=True) -> ThermalProblem:
    """ptype 1/2/3 = STEADY_STATE_1D/2D/3D.  The reference fills the soil arrays as data_2D(c, kk) with
    c = ii + (jj-1)*nx and MPPThermalSetSoils reads them as cell (c-1)*nlevgrnd + kk (z-fastest) although the mesh is
    numbered x-fastest: for nz > 1 cell ids and conductivities are permuted against each other (invisible on the
    3x3x3 baseline, whose conductivity is symmetric in x, y, z; reproduced by default for fidelity)."""
    m = M.mesh_create("Soil mesh", 1.0, 1.0, 1.0, nx, ny, nz, C.CONN_IN_XYZ_DIR)
    dx, dy, dz = 1.0 / nx, 1.0 / ny, 1.0 / nz
    n = nx * ny * nz
    T, lam, src = _mms_funcs(ptype)
    ids = (np.arange(n) + 1).reshape(nz, ny, nx)           # [kk, jj, ii] -> x-fastest id
    X, Y, Z = m.x.reshape(nz, ny, nx), m.y.reshape(nz, ny, nx), m.z.reshape(nz, ny, nx)
    lam_pos = lam(X, Y, Z)                                 # conductivity at the position of cell [kk,jj,ii]
    if replicate_reference_soil_indexing:
        tk = np.zeros(n)
        c = (np.arange(ny)[:, None] * nx + np.arange(nx)[None, :])           # c-1 for (jj,ii)
        for kk in range(nz):
            tk[(c * nz + kk).ravel()] = lam_pos[kk].ravel()                  # cell (c-1)*nz + kk (0-based)
    else:
        tk = lam_pos.ravel().copy()
    # boundary connections in the driver's order (thermal_mms_problem.F90:365-452): x faces per (kk, jj) low then high, y faces
    # per (kk, ii), z faces per (jj, ii) -- built as arrays (a Python loop over the 6 n^2 faces took a minute at 512^3)
    parts = []

    def faces(lo, hi, axis, d, area):
        """cells of the low / high plane (same shape), the face centre offset by -+ d/2 along `axis`"""
        cell = np.stack([ids[lo], ids[hi]], axis=-1).ravel()
        xyz = []
        for a, A in enumerate((X, Y, Z)):
            off = (-d / 2, d / 2) if a == axis else (0.0, 0.0)
            xyz.append(np.stack([A[lo] + off[0], A[hi] + off[1]], axis=-1).ravel())
        parts.append((cell, np.full(cell.size, d / 2), np.full(cell.size, area), T(*xyz)))

    if nx > 1:
        faces((slice(None), slice(None), 0), (slice(None), slice(None), nx - 1), 0, dx, dy * dz)       # [kk, jj]
    if ny > 1:
        faces((slice(None), 0, slice(None)), (slice(None), ny - 1, slice(None)), 1, dy, dx * dz)       # [kk, ii]
    if nz > 1:
        faces((0, slice(None), slice(None)), (nz - 1, slice(None), slice(None)), 2, dz, dx * dy)       # [jj, ii]
    idd = np.concatenate([q[0] for q in parts]) if parts else np.zeros(0, dtype=np.int64)
    dd = np.concatenate([q[1] for q in parts]) if parts else np.zeros(0)
    ar = np.concatenate([q[2] for q in parts]) if parts else np.zeros(0)
    tb = np.concatenate([q[3] for q in parts]) if parts else np.zeros(0)
    du = np.zeros(idd.size)
    nb = len(idd)
    uv = np.zeros((nb, 3)); uv[:, 0] = 1.0
    bc = M.ConnectionSet(np.zeros(nb, dtype=np.int64), np.asarray(idd, dtype=np.int64), np.asarray(du), np.asarray(dd),
                         np.asarray(ar), uv)
    return ThermalProblem(name="thermal_mms_%dd" % ptype, mesh=m, tkdry=tk, bc=bc, bc_temperature=np.asarray(tb),
                          heat_source=(src(X, Y, Z) * dx * dy * dz).ravel(), exact=T(X, Y, Z).ravel())


def to_thermal_gpu(tp: ThermalProblem, options=None, device_id=0):
    """the call sequence of thermal_mms_problem.F90:236-262 / :603-705 through the reference-shaped host API"""
    from mpp_amd import thermal as T
    n = tp.mesh.ncells_local
    mpp = T.MultiPhysicsProbThermal(device_id)
    mpp.Init(); mpp.SetName("Thermal model for MMS"); mpp.SetID(T.MPP_THERMAL_TBASED_KSP_CLM); mpp.SetMPIRank(0)
    mpp.SetNumMeshes(1)
    mpp.AddMesh(1, tp.mesh)
    mpp.AddGovEqn(T.GE_THERM_SOIL_TBASED, "Thermal equation using temprature formulation in soil (KSP formulation)", 302)
    mpp.SetMeshesOfGoveqns()
    mpp.soe.AddConditionInGovEqn(1, C.COND_BC, "Temp BC", "T", C.COND_DIRICHLET, conn_set=tp.bc)
    mpp.soe.AddConditionInGovEqn(1, C.COND_SS, "Source term for MMS", "W/m^2", T.COND_HEAT_RATE, region_type=T.ALL_CELLS)
    mpp.soe.SetOption("cnfac", tp.cnfac)
    for k, v in (options or {}).items():
        mpp.soe.SetOption(k, v)
    mpp.AllocateAuxVars()
    mpp.SetupProblem()
    T.MPPThermalSetSoils(mpp, T.ISTSOIL, 1, 0.1, 0.0, 0.0, tp.tkdry)
    mpp.soe.SetSolnPrevCLM(np.full(n, tp.t_init))
    mpp.soe.SetRDataFromCLM(C.AUXVAR_INTERNAL, T.VAR_TUNING_FACTOR, 1, np.ones(n))
    mpp.soe.SetRDataFromCLM(C.AUXVAR_INTERNAL, T.VAR_LIQ_AREAL_DEN, 1, np.zeros(n))
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, 1, tp.bc_temperature)
    nb = tp.bc.num_connections          # the driver's is_active_1d / fraction_1d of the boundary auxvars (:664-668)
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, T.VAR_ACTIVE, 1, np.ones(nb))
    mpp.soe.SetRDataFromCLM(C.AUXVAR_BC, T.VAR_FRAC, 1, np.ones(nb))
    mpp.soe.SetRDataFromCLM(C.AUXVAR_SS, C.VAR_BC_SS_CONDITION, 1, tp.heat_source)
    mpp.soe.PreStepDT()
    return mpp


# ======================================================================================================
# coupled thermal-hydrology: Richards mass equation + enthalpy-based heat equation
# ======================================================================================================
@dataclass
class THProblem:
    name: str
    mesh: M.Mesh
    # mass equation (goveqn_richards_ode_pressure_type)
    perm: np.ndarray
    por: np.ndarray
    satfunc_type: np.ndarray
    alpha: np.ndarray
    lam: np.ndarray
    sat_res: np.ndarray
    density_type: int
    ie_type: int
    # heat equation (goveqn_thermal_enthalpy_soil_type)
    heat_cap: np.ndarray
    tcond_wet: np.ndarray
    tcond_dry: np.ndarray
    talpha: np.ndarray
    den_soil: np.ndarray
    energy_perm: object = None          # None: the auxvar default 8.3913e-12 (MPPTHSetSoils never sets it)
    por_pressure_reference: np.ndarray = None   # linear porosity model (PorosityFunctionMod.F90:75-96) in both equations:
    por_slope: np.ndarray = None                # por is then porosity_base
    press_ic: np.ndarray = None
    temp_ic: np.ndarray = None
    mass_bcs: list = field(default_factory=list)     # [(cond_type, ConnectionSet, values, bc_temperature or None)]
    mass_sss: list = field(default_factory=list)     # [(cond_type, ConnectionSet, values)]
    energy_bcs: list = field(default_factory=list)   # [(cond_type, ConnectionSet, values, bc_pressure)]
    energy_sss: list = field(default_factory=list)
    dt: float = 3600.0
    nstep: int = 1


def mass_and_heat(nx=100) -> THProblem:
    """src/driver/standalone/thermal-e/mass_and_heat_model_problem.F90: nx x 1 x 1 cells along x (no gravity term),
    VG + IFC67 density + IFC67 enthalpy, P = 91325 Pa, T = 283.15 K initially, heat equation held at 303.15 K / 293.15 K
    at the two ends (with boundary pressure 91325 Pa), no condition on the mass equation, one step of 3600 s."""
    m = M.mesh_create("Soil mesh", 1.0, 1.0, 1.0, nx, 1, 1, C.CONN_IN_X_DIR)
    n = m.ncells_local
    dx = 1.0 / nx
    perm_eff, alpha_eff, lam_eff = clm_soil_conversion(8.3913e-12, 3.4257e-4, 0.5, C.GRAVITY_CONSTANT)  # :361-366; MPPTH :505-513

    def end_face(cell, ux):
        return M.ConnectionSet(np.array([0]), np.array([cell]), np.array([0.0]), np.array([0.5 * dx]), np.array([1.0 * 1.0]),
                               np.array([[ux, 0.0, 0.0]]))
    return THProblem(
        name="mass_and_heat", mesh=m, perm=np.full((n, 3), perm_eff), por=np.full(n, 0.368),
        satfunc_type=np.full(n, C.SAT_FUNC_VAN_GENUCHTEN, dtype=np.int32), alpha=np.full(n, alpha_eff),
        lam=np.full(n, lam_eff), sat_res=np.full(n, 0.2772), density_type=C.DENSITY_IFC67, ie_type=2,
        heat_cap=np.full(n, 837.0), tcond_wet=np.full(n, 1.3), tcond_dry=np.full(n, 0.25), talpha=np.full(n, 0.45),
        den_soil=np.full(n, 2700.0), press_ic=np.full(n, 91325.0), temp_ic=np.full(n, 283.15),
        energy_bcs=[(C.COND_DIRICHLET, end_face(1, 1.0), np.array([303.15]), np.array([91325.0])),
                    (C.COND_DIRICHLET, end_face(nx, -1.0), np.array([293.15]), np.array([91325.0]))],
        dt=3600.0, nstep=1)


def th_mms(nx=20) -> THProblem:
    """src/driver/standalone/thermal-e/th_mms_problem.F90 (STEADY_STATE_SOIL_ONLY_1D): manufactured steady pressure and
    temperature on x in [0,10], zero porosity and heat capacity (no storage), constant density + IFC67 enthalpy, Dirichlet
    P and T at both ends of both equations, manufactured mass and heat sources in every cell
    (set_source_sink_conditions :901-934; DATA_MASS_SOURCE :1262-1316, DATA_HEAT_SOURCE :1365-1452, with the driver's 1e-6
    central differences of density, enthalpy and the Kersten number), one step.  The driver evaluates the constitutive laws
    on the host to form those sources; so does this builder (mpp_amd.laws: the library's own functions)."""
    from . import laws as L
    x_min, x_max = 0.0, 10.0
    xlim = x_max - x_min
    dx = (x_max - x_min) / nx
    m = M.mesh_create("Soil mesh", x_max, 1.0, 1.0, nx, 1, 1, C.CONN_IN_XYZ_DIR, x_min=x_min)
    n = m.ncells_local
    xc = np.asarray(m.x, dtype=np.float64)
    pi = 4 * np.arctan(1.0)
    FMW = C.FMWH2O
    density_type, ie_type = C.DENSITY_CONSTANT, 2     # set_problem :960-962 (the second assignment wins)

    def pres(x):      # compute_pressure_or_deriv :991-1021
        a0, a1 = 15000.0, -20000.0
        sn, cs = np.sin((x - x_min) / xlim * pi), np.cos((x - x_min) / xlim * pi)
        return a0 * sn + a1 + 101325.0, a0 * pi / xlim * cs, -a0 * pi * pi / xlim / xlim * sn

    def temp(x):      # compute_temperature_or_deriv :1023-1044 (the second set of assignments wins)
        a0, a1 = 5.0, 290.0
        sn, cs = np.sin((x - x_min) / xlim * pi), np.cos((x - x_min) / xlim * pi)
        return a0 * sn * 1.0 + a1, a0 * pi / xlim * cs * 1.0, -a0 * pi * pi / xlim / xlim * sn * 1.0

    def permf(x):     # compute_permeability_or_deriv :1046-1060
        p0 = 1e-11
        return p0 * (2.0 - np.cos((x - x_min) / xlim * pi)), p0 * pi / xlim * np.sin((x - x_min) / xlim * pi)

    alpha0, lam0, sr0, kdry0, kwet0, talpha0 = 1.0 / 4000.0, 0.5, 0.0, 0.25, 1.3, 0.45
    pert = 1e-6
    mu = 8.904156e-4

    def rho_of(P, T):   # Density in kmol/m^3 -> kg/m^3, with both derivatives
        den, ddp, ddt = L.density(P, T, density_type)
        return den * FMW, ddp * FMW, ddt * FMW

    def sat_of(P):
        return L.sat_relperm(P, C.SAT_FUNC_VAN_GENUCHTEN, sr0, alpha0, lam0)   # sat, dsat_dP, kr, dkr_dP

    def enth(P, T, rho, drho_dT, drho_dP):
        return L.internal_energy_enthalpy(P, T, ie_type, rho, drho_dT, drho_dP)[1]

    k, dk_dx = permf(xc)
    P, dP_dx, d2P_dx2 = pres(xc)
    T, dT_dx, d2T_dx2 = temp(xc)
    rho, drho_dP, drho_dT = rho_of(P, T)
    se, _dse_dP, kr, dkr_dP = sat_of(P)
    dkr_dx = dkr_dP * dP_dx
    Pp, Tp = pres(xc + pert)[0], temp(xc + pert)[0]
    Pn, Tn = pres(xc - pert)[0], temp(xc - pert)[0]
    rho_p, drho_p_dP, drho_p_dT = rho_of(Pp, Tp)
    rho_n, drho_n_dP, drho_n_dT = rho_of(Pn, Tn)
    drho_dx = (rho_p - rho_n) / pert / 2.0
    drhoq_dx = -((k * kr / mu) * drho_dx + (rho * kr / mu) * dk_dx + (rho * k / mu) * dkr_dx) * dP_dx - (rho * k * kr / mu) * d2P_dx2
    mass_src = drhoq_dx * dx                                               # DATA_MASS_SOURCE :1262-1316
    H = enth(P, T, rho, drho_dT, drho_dP)
    rhoq = -rho * (k * kr / mu * dP_dx)
    se_p, se_n = sat_of(Pp)[0], sat_of(Pn)[0]
    Ke = (se + 1e-6) ** talpha0
    dKe_dx = ((se_p + 1e-6) ** talpha0 - (se_n + 1e-6) ** talpha0) / pert / 2.0
    kappa = kwet0 * Ke + kdry0 * (1.0 - Ke)
    dkappa_dx = 0.0 * Ke + 0.0 * (1.0 - Ke) + (kwet0 - kdry0) * dKe_dx
    dH_dx = (enth(Pp, Tp, rho_p, drho_p_dT, drho_p_dP) - enth(Pn, Tn, rho_n, drho_n_dT, drho_n_dP)) / pert / 2.0
    heat_src = -(drhoq_dx * H / FMW + rhoq * dH_dx / FMW - dkappa_dx * dT_dx - kappa * d2T_dx2) * dx   # :1365-1452

    P0 = 0.0; T0 = 0.0
    for i in range(n):      # DATA_INITIAL_PRESSURE / _TEMPERATURE: running mean, same summation order
        P0 = P0 + 1.0 / nx * P[i]; T0 = T0 + 1.0 / nx * T[i]
    pres_bc = np.array([pres(xc[0] - dx / 2.0)[0], pres(xc[-1] + dx / 2.0)[0]])
    temp_bc = np.array([temp(xc[0] - dx / 2.0)[0], temp(xc[-1] + dx / 2.0)[0]])
    ends = M.ConnectionSet(np.array([0, 0]), np.array([1, nx]), np.zeros(2), np.full(2, dx / 2.0), np.full(2, 1.0 * 1.0),
                           np.array([[1.0, 0.0, 0.0], [-1.0, 0.0, 0.0]]))      # ComputeBoundaryDomainConnection, ny = nz = 1
    ids = np.arange(1, n + 1)
    z = np.zeros(n)
    allcells = M.ConnectionSet(np.zeros(n, dtype=np.int64), ids, z, z, np.ones(n), np.zeros((n, 3)))
    return THProblem(
        name="th_mms", mesh=m, perm=np.stack([k] * 3, axis=1), por=np.zeros(n),
        satfunc_type=np.full(n, C.SAT_FUNC_VAN_GENUCHTEN, dtype=np.int32), alpha=np.full(n, alpha0), lam=np.full(n, lam0),
        sat_res=np.full(n, sr0), density_type=density_type, ie_type=ie_type, heat_cap=np.zeros(n),
        tcond_wet=np.full(n, kwet0), tcond_dry=np.full(n, kdry0), talpha=np.full(n, talpha0), den_soil=np.zeros(n),
        energy_perm=(k, k, k), press_ic=np.full(n, P0), temp_ic=np.full(n, T0),
        mass_bcs=[(C.COND_DIRICHLET, ends, pres_bc, temp_bc)], mass_sss=[(C.COND_MASS_RATE, allcells, mass_src)],
        energy_bcs=[(C.COND_DIRICHLET, ends, temp_bc, pres_bc)], energy_sss=[(C.COND_HEAT_RATE, allcells, heat_src)],
        dt=1.0, nstep=1)


def to_th_gpu(tp: THProblem, options=None, device_id=0):
    """the reference drivers' call sequence (mass_and_heat_model_problem.F90:112-139) through the Python mirror"""
    from mpp_amd import th as TH
    mpp = TH.MultiPhysicsProbTH(device_id)
    mpp.Init(); mpp.SetName(tp.name); mpp.SetID(TH.MPP_TH_SNES_CLM); mpp.SetMPIRank(0)
    mpp.SetNumMeshes(1); mpp.AddMesh(1, tp.mesh)
    mpp.AddGovEqn(C.GE_RE, "Mass equation", 0)
    mpp.AddGovEqn(TH.GE_THERM_SOIL_EBASED, "Heat transport based on enthalpy", 0)
    mpp.SetMeshesOfGoveqns()
    ids = {"mass_bc": [], "mass_ss": [], "energy_bc": [], "energy_ss": []}
    for bc in tp.mass_bcs:
        ids["mass_bc"].append(mpp.soe.AddConditionInGovEqn(1, C.COND_BC, "bc", "Pa", bc[0], conn_set=bc[1]))
    for ss in tp.mass_sss:
        ids["mass_ss"].append(mpp.soe.AddConditionInGovEqn(1, C.COND_SS, "ss", "kg/s", ss[0], conn_set=ss[1]))
    for bc in tp.energy_bcs:
        ids["energy_bc"].append(mpp.soe.AddConditionInGovEqn(2, C.COND_BC, "bc", "K", bc[0], conn_set=bc[1]))
    for ss in tp.energy_sss:
        ids["energy_ss"].append(mpp.soe.AddConditionInGovEqn(2, C.COND_SS, "ss", "W", ss[0], conn_set=ss[1]))
    mpp.GovEqnSetInternalCouplingVars(1, 1, [TH.VAR_TEMPERATURE], [2])
    mpp.GovEqnSetInternalCouplingVars(2, 1, [C.VAR_PRESSURE], [1])
    mpp.AllocateAuxVars()
    for k, v in (options or {}).items():
        mpp.soe.SetOption(k, v)
    mpp.SetupProblem()
    TH.MPPTHSetSoilsSI(mpp, tp.perm, tp.por, tp.satfunc_type, tp.alpha, tp.lam, tp.sat_res, tp.density_type, tp.ie_type,
                       tp.heat_cap, tp.tcond_wet, tp.tcond_dry, tp.talpha, tp.den_soil, energy_perm=tp.energy_perm)
    if tp.por_slope is not None:   # POROSITY_LINEAR in both equations (porParams are per auxvar; no MPP-level routine of the reference)
        from mpp_amd import vsfm as V
        V.VSFMMPPSetSoilPorosityLinearModel(mpp, 1, tp.por, tp.por_pressure_reference, tp.por_slope)
    mpp.Restart(tp.press_ic, tp.temp_ic)
    mpp.cond_ids = ids
    th_gpu_set_conditions(tp, mpp)
    return mpp


def th_gpu_set_conditions(tp: THProblem, mpp):
    ids = mpp.cond_ids
    for i, bc in enumerate(tp.mass_bcs):
        mpp.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, ids["mass_bc"][i], bc[2])
        if len(bc) > 3 and bc[3] is not None:
            mpp.soe.SetBoundaryAuxVarTemperature(ids["mass_bc"][i], bc[3])
    for i, ss in enumerate(tp.mass_sss):
        mpp.soe.SetDataFromCLM(C.AUXVAR_SS, C.VAR_BC_SS_CONDITION, ids["mass_ss"][i], ss[2])
    for i, bc in enumerate(tp.energy_bcs):
        mpp.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, ids["energy_bc"][i], bc[2])
        mpp.soe.SetBoundaryAuxVarPressure(ids["energy_bc"][i], bc[3])
    for i, ss in enumerate(tp.energy_sss):
        mpp.soe.SetDataFromCLM(C.AUXVAR_SS, C.VAR_BC_SS_CONDITION, ids["energy_ss"][i], ss[2])


def th_columns(ncols=1, nz=100, perm_mult=None, density_type=None, ie_type=2) -> THProblem:
    """'TH-extruded' grid (BASELINE.json configs[4] shape): `ncols` celia1990 soil columns (vertical connections,
    gravity) carrying the coupled mass + heat equations: the celia soil and pressure boundary values, 283.15 K initially,
    303.15 K at the wet top boundary and 283.15 K at the bottom."""
    vs = celia1990(ncols=ncols, nz=nz, perm_mult=perm_mult)
    m = vs.mesh
    n = m.ncells_local
    (ttype, tcs, tvals), (btype, bcs_, bvals) = vs.bcs
    one = lambda cs, v: np.full(cs.num_connections, v)
    return THProblem(
        name="th_columns", mesh=m, perm=vs.perm, por=vs.por, satfunc_type=vs.satfunc_type, alpha=vs.alpha, lam=vs.lam,
        sat_res=vs.sat_res, density_type=C.DENSITY_IFC67 if density_type is None else density_type, ie_type=ie_type,
        heat_cap=np.full(n, 837.0), tcond_wet=np.full(n, 1.3), tcond_dry=np.full(n, 0.25), talpha=np.full(n, 0.45),
        den_soil=np.full(n, 2700.0), press_ic=vs.press_ic.copy(), temp_ic=np.full(n, 283.15),
        mass_bcs=[(ttype, tcs, tvals, one(tcs, 303.15)), (btype, bcs_, bvals, one(bcs_, 283.15))],
        energy_bcs=[(C.COND_DIRICHLET, tcs, one(tcs, 303.15), tvals.copy()), (C.COND_DIRICHLET, bcs_, one(bcs_, 283.15), bvals.copy())],
        dt=vs.dt, nstep=1)


# ------------------------------------------------------------------------------------------------------------
# a problem split across ranks: owned cells + ghost cells per rank (mpp_amd.parallel.partition_mesh)
# ------------------------------------------------------------------------------------------------------------
def _restrict_conn_set(cs, vals, part, g2l):
    """the entries of a boundary / source connection set whose cell this rank owns (cells renumbered locally)"""
    cells = np.asarray(cs.id_dn) - 1
    loc = g2l[cells]
    keep = (loc >= 0) & (loc < part.nowned)
    sub = M.ConnectionSet(np.asarray(cs.id_up)[keep], loc[keep] + 1, np.asarray(cs.dist_up)[keep], np.asarray(cs.dist_dn)[keep],
                          np.asarray(cs.area)[keep], np.asarray(cs.unit_vec)[keep], None if cs.itype is None else np.asarray(cs.itype)[keep])
    return sub, np.asarray(vals)[keep]


def split_problem(prob: Problem, nranks, mode="auto"):
    """per-rank VSFM problems of a 3-D problem: ranks own whole soil columns (x-y blocks of the column set, ghosts = the neighbour
    columns' cells; mpp_amd.parallel.partition_mesh, SURVEY.md 8e, MeshType.F90:293-460); returns [(part, local problem)]"""
    from mpp_amd import parallel as PAR
    out = []
    n = prob.mesh.ncells_local
    for part in PAR.partition_mesh(prob.mesh, nranks, mode):
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[part.l2g] = np.arange(part.l2g.size)
        l = part.l2g
        lp = Problem(name=prob.name, mesh=part.mesh, perm=prob.perm[l], por=prob.por[l], satfunc_type=prob.satfunc_type[l],
                     alpha=prob.alpha[l], lam=prob.lam[l], sat_res=prob.sat_res[l], density_type=prob.density_type,
                     press_ic=prob.press_ic[l], dt=prob.dt, nstep=prob.nstep)
        lp.bcs = [(t,) + _restrict_conn_set(cs, v, part, g2l) for (t, cs, v) in prob.bcs]
        lp.sss = [(t,) + _restrict_conn_set(cs, v, part, g2l) for (t, cs, v) in prob.sss]
        if prob.por_slope is not None:
            lp.por_pressure_reference = np.asarray(prob.por_pressure_reference)[l]
            lp.por_slope = np.asarray(prob.por_slope)[l]
        out.append((part, lp))
    return out


def split_thermal_problem(tp: ThermalProblem, nranks, mode="ranges"):
    """per-rank thermal problems; the isotropic cube is indifferent to the axis of the cut, so the default stays contiguous id ranges
    (z-slabs), which thermal_mms_rank builds without the global mesh; mode="columns" cuts x-y blocks of whole columns instead"""
    from mpp_amd import parallel as PAR
    out = []
    n = tp.mesh.ncells_local
    for part in PAR.partition_mesh(tp.mesh, nranks, mode):
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[part.l2g] = np.arange(part.l2g.size)
        l = part.l2g
        bc, bct = _restrict_conn_set(tp.bc, tp.bc_temperature, part, g2l)
        out.append((part, ThermalProblem(name=tp.name, mesh=part.mesh, tkdry=np.asarray(tp.tkdry)[l], bc=bc, bc_temperature=bct,
                                         heat_source=np.asarray(tp.heat_source)[l], exact=np.asarray(tp.exact)[l], t_init=tp.t_init,
                                         cnfac=tp.cnfac, dt=tp.dt)))
    return out


def thermal_mms_rank(nx, ny, nz, world, rank):
    """Rank `rank`'s share of thermal_mms(3, nx, ny, nz, replicate_reference_soil_indexing=False) split over `world` ranks into
    contiguous id ranges (mpp_amd.parallel.shard_columns over the cells: z-slabs for the x-fastest numbering) -- built WITHOUT the
    global problem when the ranges are whole planes (nz a multiple of world), which is what a rank process of
    `bench.py --workload thermal --gpus N` calls: a 512^3 mesh per rank would cost every rank the memory and the minute of the
    whole problem.  Same cells, order, connections, halo plan and boundary entries as split_thermal_problem(...)[rank]
    (tests/test_parallel_gloo.py compares the two array by array); returns (part, local problem)."""
    from mpp_amd import parallel as PAR
    if world == 1:
        tp = thermal_mms(3, nx, ny, nz, replicate_reference_soil_indexing=False)
        n = tp.mesh.ncells_local
        return PAR.MeshPart(tp.mesh, np.arange(n), 0, np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)), tp
    if nz % world != 0:
        return split_thermal_problem(thermal_mms(3, nx, ny, nz, replicate_reference_soil_indexing=False), world)[rank]
    nxy, npl = nx * ny, nz // world
    k0 = rank * npl                                   # owned planes k0 .. k0 + npl - 1 (0-based)
    lo_ghost, hi_ghost = rank > 0, rank < world - 1
    planes = list(range(k0, k0 + npl)) + ([k0 - 1] if lo_ghost else []) + ([k0 + npl] if hi_ghost else [])
    planes = np.asarray(planes, dtype=np.int64)
    nown, nloc = npl * nxy, len(planes) * nxy
    dx, dy, dz = 1.0 / nx, 1.0 / ny, 1.0 / nz
    # local cell [p, jj, ii] (p: position in `planes`): centroids by MeshCreate's expression in the GLOBAL indices (mesh.mesh_create)
    kk = (planes + 1)[:, None, None]
    jj = np.arange(1, ny + 1)[None, :, None]
    ii = np.arange(1, nx + 1)[None, None, :]
    shape = (len(planes), ny, nx)
    X = np.broadcast_to(dx / 2.0 + dx * (ii - 1) + 0.0, shape).astype(np.float64)
    Y = np.broadcast_to(dy / 2.0 + dy * (jj - 1) + 0.0, shape).astype(np.float64)
    Z = np.broadcast_to(dz / 2.0 + dz * (kk - 1) + 0.0, shape).astype(np.float64)
    l2g = (planes[:, None, None] * nxy + (jj - 1) * nx + (ii - 1)).reshape(-1)
    lm = M.Mesh(name="Soil mesh", ncells_local=nloc, nlev=nz, orientation=C.MESH_AGAINST_GRAVITY)
    lm.x, lm.y, lm.z = X.ravel().copy(), Y.ravel().copy(), Z.ravel().copy()
    lm.dx, lm.dy, lm.dz = np.full(nloc, dx), np.full(nloc, dy), np.full(nloc, dz)
    lm.area_xy = np.full(nloc, dx * dy)
    lm.is_active = np.ones(nloc, dtype=np.int32)
    lm.compute_volume()
    # local id (1-based) of the cell (plane position p, jj, ii): owned planes first, then the ghost planes -- the order of `planes`
    lid = (np.arange(nloc, dtype=np.int64) + 1).reshape(shape)
    pos = {int(k): p for p, k in enumerate(planes)}
    # internal connections with at least one owned end, in the global order: x set, y set, z set; within a set ii outer, jj, kk inner
    own = lid[:npl]                                    # [kk_owned, jj, ii]
    sets = []
    if nx > 1:
        sets.append((own[:, :, :-1], own[:, :, 1:], dx, dy * dz, 0))
    if ny > 1:
        sets.append((own[:, :-1, :], own[:, 1:, :], dy, dx * dz, 1))
    zlo = [k for k in range(k0 - 1, k0 + npl) if 0 <= k < nz - 1 and (k in pos) and (k + 1 in pos)]   # pairs (k, k + 1)
    if zlo:
        up = np.stack([lid[pos[k]] for k in zlo]); dn = np.stack([lid[pos[k + 1]] for k in zlo])
        sets.append((up, dn, dz, dx * dy, 2))
    ups, dns, dus, ars, uvs = [], [], [], [], []
    for up, dn, dist, area, axis in sets:
        u = np.ascontiguousarray(up.transpose(2, 1, 0)).ravel(); d = np.ascontiguousarray(dn.transpose(2, 1, 0)).ravel()
        ups.append(u); dns.append(d); dus.append(np.full(u.size, 0.5 * dist)); ars.append(np.full(u.size, area))
        uv = np.zeros((u.size, 3)); uv[:, axis] = 1.0
        uvs.append(uv)
    up, dn, du, ar, uv = np.concatenate(ups), np.concatenate(dns), np.concatenate(dus), np.concatenate(ars), np.concatenate(uvs)
    lm.intrn_conn_sets.append(M.ConnectionSet(up, dn, du, du.copy(), ar, uv, np.full(up.size, C.CONN_HORIZONTAL, dtype=np.int32)))
    lm.ncols, lm.ordering = nxy, "x-fastest"
    nghost = nloc - nown
    peers = [q for q, there in ((rank - 1, lo_ghost), (rank + 1, hi_ghost)) if there]
    send_cells, send_ptr = [], [0]
    for q in peers:    # my plane next to q, ascending id
        plane = 0 if q < rank else npl - 1
        send_cells.append(np.arange(plane * nxy, (plane + 1) * nxy, dtype=np.int64) + 1)
        send_ptr.append(send_ptr[-1] + nxy)
    lm.nghost = nghost
    lm.halo = (np.asarray(peers, dtype=np.int32), np.asarray(send_ptr, dtype=np.int32),
               np.concatenate(send_cells).astype(np.int32) if send_cells else np.zeros(0, np.int32), np.full(len(peers), nxy, dtype=np.int32))
    part = PAR.MeshPart(lm, l2g, nghost, *lm.halo)
    # the manufactured problem on these cells; boundary faces of owned cells in the driver's order (x faces per (kk, jj) low then
    # high, y faces per (kk, ii), z faces per (jj, ii): thermal_mms above)
    T, lam, src = _mms_funcs(3)
    out_cells, out_dd, out_ar, out_tb = [], [], [], []

    def emit(cell, xyz, d, area):
        out_cells.append(cell.ravel()); out_dd.append(np.full(cell.size, d / 2)); out_ar.append(np.full(cell.size, area))
        out_tb.append(T(xyz[0].ravel(), xyz[1].ravel(), xyz[2].ravel()))

    Xo, Yo, Zo = X[:npl], Y[:npl], Z[:npl]

    def pair(sel_lo, sel_hi, axis, d, area):
        cell = np.stack([own[sel_lo], own[sel_hi]], axis=-1)
        xyz = [np.stack([A[sel_lo] + ((-d / 2) if a == axis else 0.0), A[sel_hi] + ((d / 2) if a == axis else 0.0)], axis=-1)
               for a, A in enumerate((Xo, Yo, Zo))]
        emit(cell, xyz, d, area)

    if nx > 1:
        pair((slice(None), slice(None), 0), (slice(None), slice(None), nx - 1), 0, dx, dy * dz)
    if ny > 1:
        pair((slice(None), 0, slice(None)), (slice(None), ny - 1, slice(None)), 1, dy, dx * dz)
    if nz > 1:   # the bottom plane belongs to rank 0, the top plane to the last rank
        if rank == 0:
            emit(own[0], [Xo[0], Yo[0], Zo[0] - dz / 2], dz, dx * dy)
        if rank == world - 1:
            emit(own[npl - 1], [Xo[npl - 1], Yo[npl - 1], Zo[npl - 1] + dz / 2], dz, dx * dy)
    idd = np.concatenate(out_cells); dd = np.concatenate(out_dd); arb = np.concatenate(out_ar); tb = np.concatenate(out_tb)
    nb = idd.size
    uvb = np.zeros((nb, 3)); uvb[:, 0] = 1.0
    bc = M.ConnectionSet(np.zeros(nb, dtype=np.int64), idd.astype(np.int64), np.zeros(nb), dd, arb, uvb)
    tp = ThermalProblem(name="thermal_mms_3d", mesh=lm, tkdry=lam(X, Y, Z).ravel().copy(), bc=bc, bc_temperature=tb,
                        heat_source=(src(X, Y, Z) * dx * dy * dz).ravel(), exact=T(X, Y, Z).ravel())
    return part, tp


def _th_on_vchannel(vs: Problem, nx, ny, nz) -> THProblem:
    """the coupled problem's extra inputs on a (piece of a) v-channel mesh; the extents that enter are those of the WHOLE grid, from
    the surface, so that a rank's piece carries the values of the global problem"""
    m = vs.mesh
    n = m.ncells_local
    z = np.asarray(m.z); x = np.asarray(m.x)
    xc2, _yc2, zc2 = _vchannel_surface(nx, ny)
    dz = 0.5
    zmin, zmax = float(zc2.min()) - (dz / 2.0 + (nz - 1) * dz), float(zc2.max()) - dz / 2.0
    xmin, xmax = float(xc2.min()), float(xc2.max())
    temp_ic = 283.15 + 4.0 * (z - zmin) / max(zmax - zmin, 1e-30)
    ids = np.arange(1, n + 1)
    zero = np.zeros(n)
    allcells = M.ConnectionSet(np.zeros(n, dtype=np.int64), ids, zero, zero, np.ones(n), np.zeros((n, 3)))
    heat = 40.0 * np.asarray(m.vol) * np.exp(-((x - 0.5 * (xmin + xmax)) / (0.25 * (xmax - xmin + 1e-30))) ** 2)
    return THProblem(
        name="th_vchannel", mesh=m, perm=vs.perm, por=vs.por, satfunc_type=vs.satfunc_type, alpha=vs.alpha, lam=vs.lam,
        sat_res=vs.sat_res, density_type=C.DENSITY_TGDPB01, ie_type=2,
        heat_cap=np.full(n, 837.0), tcond_wet=np.full(n, 1.3), tcond_dry=np.full(n, 0.25), talpha=np.full(n, 0.45),
        den_soil=np.full(n, 2700.0), press_ic=vs.press_ic.copy(), temp_ic=temp_ic,
        energy_sss=[(C.COND_HEAT_RATE, allcells, heat)], dt=vs.dt, nstep=2)


def th_vchannel(nx=6, ny=5, nz=7) -> THProblem:
    """the coupled mass + heat equations on the 3-D v-channel mesh (7-point stencil, sloping surface, gravity along every
    connection): vchannel's soil and hydrostatic start, a temperature that falls with depth, no boundary conditions (as the
    v-channel problem), and a heat source in every cell that warms the channel's axis"""
    return _th_on_vchannel(vchannel(nx, ny, nz), nx, ny, nz)


def th_vchannel_rank(nx, ny, nz, world, rank):
    """rank `rank`'s share of th_vchannel(nx, ny, nz) over `world` ranks that own whole soil columns, built without the global
    problem when ny is a multiple of world (`bench.py --workload th --mesh 3d --gpus N`); equals split_th_problem(...)[rank]
    array by array (tests/test_parallel_gloo.py); returns (part, local problem)"""
    if world > 1 and ny % world != 0:
        return split_th_problem(th_vchannel(nx, ny, nz), world)[rank]
    part, vs = vchannel_rank(nx, ny, nz, world, rank)
    tp = _th_on_vchannel(vs, nx, ny, nz)
    if part.nghost > 0:          # sources act on owned cells only (workloads._restrict_conn_set)
        t, cs, vals = tp.energy_sss[0]
        k = part.nowned
        sub = M.ConnectionSet(np.asarray(cs.id_up)[:k], np.asarray(cs.id_dn)[:k], np.asarray(cs.dist_up)[:k], np.asarray(cs.dist_dn)[:k],
                              np.asarray(cs.area)[:k], np.asarray(cs.unit_vec)[:k], None)
        tp.energy_sss = [(t, sub, np.asarray(vals)[:k])]
    return part, tp


def split_th_problem(tp: THProblem, nranks, mode="auto"):
    """per-rank coupled TH problems of a mesh split by whole soil columns (owned cells + ghost cells, halo plan; mode="ranges":
    contiguous id ranges, the z-slabs rounds 3-4 used); returns [(part, local problem)]"""
    from mpp_amd import parallel as PAR
    out = []
    n = tp.mesh.ncells_local
    pick = lambda a, l: None if a is None else np.asarray(a)[l]
    for part in PAR.partition_mesh(tp.mesh, nranks, mode):
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[part.l2g] = np.arange(part.l2g.size)
        l = part.l2g

        def bc4(entry):
            t, cs, vals, other = entry[0], entry[1], entry[2], (entry[3] if len(entry) > 3 else None)
            sub, v = _restrict_conn_set(cs, vals, part, g2l)
            o = None if other is None else _restrict_conn_set(cs, other, part, g2l)[1]
            return (t, sub, v, o)

        lp = THProblem(name=tp.name, mesh=part.mesh, perm=np.asarray(tp.perm)[l], por=pick(tp.por, l), satfunc_type=pick(tp.satfunc_type, l),
                       alpha=pick(tp.alpha, l), lam=pick(tp.lam, l), sat_res=pick(tp.sat_res, l), density_type=tp.density_type,
                       ie_type=tp.ie_type, heat_cap=pick(tp.heat_cap, l), tcond_wet=pick(tp.tcond_wet, l), tcond_dry=pick(tp.tcond_dry, l),
                       talpha=pick(tp.talpha, l), den_soil=pick(tp.den_soil, l),
                       energy_perm=None if tp.energy_perm is None else tuple(np.asarray(a)[l] for a in tp.energy_perm),
                       por_pressure_reference=pick(tp.por_pressure_reference, l), por_slope=pick(tp.por_slope, l),
                       press_ic=pick(tp.press_ic, l), temp_ic=pick(tp.temp_ic, l), dt=tp.dt, nstep=tp.nstep)
        lp.mass_bcs = [bc4(e) for e in tp.mass_bcs]
        lp.energy_bcs = [bc4(e) for e in tp.energy_bcs]
        lp.mass_sss = [(t,) + _restrict_conn_set(cs, v, part, g2l) for (t, cs, v) in tp.mass_sss]
        lp.energy_sss = [(t,) + _restrict_conn_set(cs, v, part, g2l) for (t, cs, v) in tp.energy_sss]
        out.append((part, lp))
    return out
