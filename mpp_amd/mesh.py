"""Host-side mirror of the reference's mesh / connection-set builders (numpy, no GPU).

What it restates (paths relative to the reference tree):
  MeshCreate            src/mpp/dtypes/MeshType.F90:173-290   (uniform structured mesh, x-fastest ids)
  ComputeCellID         src/mpp/util/mpp_mesh_utils.F90:967-990
  ComputeCentroids1     src/mpp/util/mpp_mesh_utils.F90 (dc = dd/2 + dd*(index-1) + dmin)
  ComputeInternalConnections / ComputeIntConnAlongADirection
                        src/mpp/util/mpp_mesh_utils.F90:251-470 (loop order ii outer, jj, kk inner)
  MeshCreateConnectionSet (region based)  src/mpp/dtypes/MeshType.F90:648-851
  MeshCreateConnectionSet (explicit)      src/mpp/dtypes/MeshType.F90:854-950 (unit vec from centroids dn-up)
  CLM soil-column mesh (z-fastest ids)    src/mpp/dtypes/MeshType.F90:293-645 (layout only)

All ids are 1-based, as the Fortran callers hold them.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import constants as C


@dataclass
class ConnectionSet:
    """connection_set_type (src/mpp/dtypes/ConnectionSetType.F90:15-62) as struct-of-arrays."""

    id_up: np.ndarray
    id_dn: np.ndarray
    dist_up: np.ndarray
    dist_dn: np.ndarray
    area: np.ndarray
    unit_vec: np.ndarray  # (n, 3)
    itype: np.ndarray | None = None

    @property
    def num_connections(self) -> int:
        return int(self.id_dn.shape[0])


@dataclass
class Mesh:
    """mesh_type (src/mpp/dtypes/MeshType.F90) -- only the members the VSFM hot path reads."""

    name: str
    ncells_local: int
    nlev: int
    orientation: int = C.MESH_AGAINST_GRAVITY
    x: np.ndarray | None = None
    y: np.ndarray | None = None
    z: np.ndarray | None = None
    dx: np.ndarray | None = None
    dy: np.ndarray | None = None
    dz: np.ndarray | None = None
    area_xy: np.ndarray | None = None
    vol: np.ndarray | None = None
    is_active: np.ndarray | None = None
    intrn_conn_sets: list = field(default_factory=list)
    # structure hint carried to the device library (never needed for correctness)
    ncols: int = 0
    ordering: str = "x-fastest"  # or "z-fastest"

    def compute_volume(self):
        """MeshComputeVolume, MeshType.F90:1244-1264."""
        self.vol = self.area_xy * self.dz

    def internal_connections(self) -> ConnectionSet:
        if len(self.intrn_conn_sets) != 1:
            # the reference reads ids with conn(sum_conn) but geometry with conn(iconn)
            # (GoveqnRichardsODEPressureType.F90:1759 vs 1775): only one internal set is meaningful.
            raise ValueError("exactly one internal connection set is supported (reference quirk 1)")
        return self.intrn_conn_sets[0]


def compute_internal_connections(nx, ny, nz, dx, dy, dz, idir) -> tuple:
    """ComputeInternalConnections: returns (id_up, id_dn, dist_up, dist_dn, area, itype)."""
    ids = (np.arange(nx * ny * nz, dtype=np.int64) + 1).reshape(nz, ny, nx)  # id(ii,jj,kk) = [kk,jj,ii]
    outs = []
    dirs = {C.CONN_IN_X_DIR: [0], C.CONN_IN_Y_DIR: [1], C.CONN_IN_Z_DIR: [2], C.CONN_IN_XYZ_DIR: [0, 1, 2]}[idir]
    for d in dirs:
        if d == 0:
            up, dn, dist, area = ids[:, :, :-1], ids[:, :, 1:], dx, dy * dz
        elif d == 1:
            up, dn, dist, area = ids[:, :-1, :], ids[:, 1:, :], dy, dx * dz
        else:
            up, dn, dist, area = ids[:-1, :, :], ids[1:, :, :], dz, dx * dy
        # reference loop order: ii outer, jj middle, kk inner  ->  transpose [kk,jj,ii] -> [ii,jj,kk]
        up = np.ascontiguousarray(up.transpose(2, 1, 0)).ravel()
        dn = np.ascontiguousarray(dn.transpose(2, 1, 0)).ravel()
        n = up.size
        outs.append((up, dn, np.full(n, 0.5 * dist), np.full(n, 0.5 * dist), np.full(n, area),
                     np.full(n, C.CONN_HORIZONTAL, dtype=np.int32)))  # itype is always HORIZONTAL (:462)
    return tuple(np.concatenate([o[k] for o in outs]) for k in range(6))


def unit_vec_from_centroids(mesh: Mesh, id_up, id_dn) -> np.ndarray:
    """MeshCreateConnectionSet2 without unit_vec: (x_dn - x_up)/|.| (MeshType.F90:932-938)."""
    d = np.stack([mesh.x[id_dn - 1] - mesh.x[id_up - 1], mesh.y[id_dn - 1] - mesh.y[id_up - 1],
                  mesh.z[id_dn - 1] - mesh.z[id_up - 1]], axis=1)
    dist = (d[:, 0] ** 2.0 + d[:, 1] ** 2.0 + d[:, 2] ** 2.0) ** 0.5
    return d / dist[:, None]


def mesh_create(name, x_max, y_max, z_max, nx, ny, nz, internal_conn_dir,
                x_min=0.0, y_min=0.0, z_min=0.0) -> Mesh:
    """MeshCreate(mesh, name, [x_min..,] x_max, y_max, z_max, nx, ny, nz, conn_dir)."""
    dx, dy, dz = (x_max - x_min) / nx, (y_max - y_min) / ny, (z_max - z_min) / nz
    n = nx * ny * nz
    kk, jj, ii = np.meshgrid(np.arange(1, nz + 1), np.arange(1, ny + 1), np.arange(1, nx + 1), indexing="ij")
    m = Mesh(name=name, ncells_local=n, nlev=nz, orientation=C.MESH_AGAINST_GRAVITY)
    m.x = (dx / 2.0 + dx * (ii - 1) + x_min).ravel().astype(np.float64)
    m.y = (dy / 2.0 + dy * (jj - 1) + y_min).ravel().astype(np.float64)
    m.z = (dz / 2.0 + dz * (kk - 1) + z_min).ravel().astype(np.float64)
    m.dx, m.dy, m.dz = np.full(n, dx), np.full(n, dy), np.full(n, dz)
    m.area_xy = np.full(n, dx * dy)
    m.is_active = np.ones(n, dtype=np.int32)
    m.compute_volume()
    up, dn, du, dd, area, itype = compute_internal_connections(nx, ny, nz, dx, dy, dz, internal_conn_dir)
    cs = ConnectionSet(up, dn, du, dd, area, None, itype)
    # unit vectors (x_dn - x_up) / |.| (MeshType.F90:932-938): on this grid the two cells of a connection differ in one coordinate
    # only, so the quotient is exactly +1 along that axis and 0 elsewhere -- written down per direction block instead of gathered
    # from the centroids (identical values; the gathers took longer than everything else in here)
    dirs = {C.CONN_IN_X_DIR: [0], C.CONN_IN_Y_DIR: [1], C.CONN_IN_Z_DIR: [2], C.CONN_IN_XYZ_DIR: [0, 1, 2]}[internal_conn_dir]
    counts = {0: (nx - 1) * ny * nz, 1: nx * (ny - 1) * nz, 2: nx * ny * (nz - 1)}
    uv = np.zeros((up.size, 3))
    at = 0
    for d in dirs:
        uv[at:at + counts[d], d] = 1.0
        at += counts[d]
    cs.unit_vec = uv
    m.intrn_conn_sets.append(cs)
    m.ncols, m.ordering = nx * ny, "x-fastest"
    return m


def mesh_create_soil_columns(name, ncols, nz, dz, area=1.0, dx=1.0, dy=1.0) -> Mesh:
    """CLM-style mesh: `ncols` independent soil columns, ids z-fastest ((c-1)*nz + k), cell 1 of a
    column at the bottom (MESH_AGAINST_GRAVITY), vertical connections only.  This is the layout
    MeshCreateConnectionSet's SOIL_TOP_CELLS/SOIL_BOTTOM_CELLS regions (id_dn = nlev*c, nlev*(c-1)+1;
    MeshType.F90:738-761) and VSFMMPPSetSoilsCLM's icell=(c-begc)*nlevgrnd+j (MultiPhysicsProbVSFM.F90:364) assume.
    `dz` may be a scalar or an array of nz layer thicknesses (bottom -> top)."""
    dzv = np.broadcast_to(np.asarray(dz, dtype=np.float64), (nz,)).copy()
    n = ncols * nz
    m = Mesh(name=name, ncells_local=n, nlev=nz, orientation=C.MESH_AGAINST_GRAVITY)
    zi = np.concatenate([[0.0], np.cumsum(dzv)])
    zc = 0.5 * (zi[:-1] + zi[1:])
    cols = np.arange(ncols)
    m.x = np.repeat(dx / 2.0 + dx * cols, nz).astype(np.float64)
    m.y = np.full(n, dy / 2.0)
    m.z = np.tile(zc, ncols)
    m.dx, m.dy = np.full(n, dx), np.full(n, dy)
    m.dz = np.tile(dzv, ncols)
    m.area_xy = np.full(n, float(area))
    m.is_active = np.ones(n, dtype=np.int32)
    m.compute_volume()
    base = (np.arange(ncols, dtype=np.int64) * nz)[:, None]
    k = np.arange(1, nz, dtype=np.int64)[None, :]
    up = (base + k).ravel()
    dn = up + 1
    du = np.tile(0.5 * dzv[:-1], ncols)
    dd = np.tile(0.5 * dzv[1:], ncols)
    uv = np.zeros((up.size, 3))
    uv[:, 2] = 1.0
    m.intrn_conn_sets.append(ConnectionSet(up, dn, du, dd, np.full(up.size, float(area)), uv,
                                           np.full(up.size, C.CONN_VERTICAL, dtype=np.int32)))
    m.ncols, m.ordering = ncols, "z-fastest"
    return m


def create_region_connection_set(mesh: Mesh, region_itype: int) -> ConnectionSet:
    """MeshCreateConnectionSet1 for SOIL_TOP_CELLS / SOIL_BOTTOM_CELLS (MeshType.F90:723-806).
    Faithful to the reference: id_dn = nlev*c (top) or nlev*(c-1)+1 (bottom) for MESH_AGAINST_GRAVITY,
    i.e. it addresses column tops/bottoms only on z-fastest meshes or when nx*ny == 1."""
    ncols = mesh.ncells_local // mesh.nlev
    c = np.arange(1, ncols + 1, dtype=np.int64)
    against = mesh.orientation == C.MESH_AGAINST_GRAVITY
    if region_itype == C.SOIL_TOP_CELLS:
        id_dn = mesh.nlev * c if against else mesh.nlev * (c - 1) + 1
        uz = -1.0
    elif region_itype == C.SOIL_BOTTOM_CELLS:
        id_dn = mesh.nlev * (c - 1) + 1 if against else mesh.nlev * c
        uz = 1.0
    else:
        raise ValueError("region not supported: %r" % region_itype)
    uv = np.zeros((ncols, 3))
    uv[:, 2] = uz
    return ConnectionSet(np.full(ncols, -1, dtype=np.int64), id_dn, np.zeros(ncols), 0.5 * mesh.dz[id_dn - 1],
                         mesh.area_xy[id_dn - 1].copy(), uv, None)


def column_blocks(mesh: Mesh) -> tuple:
    """Cells of every soil column, bottom -> top, as (block_ptr, block_cells[1-based]).  Derived from
    the vertical chains of the internal connection set (any numbering)."""
    cs = mesh.internal_connections()
    n = mesh.ncells_local
    vert = np.abs(cs.unit_vec[:, 2]) > 0.5 if cs.unit_vec is not None else np.ones(cs.num_connections, bool)
    nxt = np.zeros(n + 1, dtype=np.int64)
    has_below = np.zeros(n + 1, dtype=bool)
    nxt[cs.id_up[vert]] = cs.id_dn[vert]
    has_below[cs.id_dn[vert]] = True
    starts = np.nonzero(~has_below[1:])[0] + 1
    ptr, cells = [0], []
    for s in starts:
        cur = s
        while cur:
            cells.append(cur)
            cur = nxt[cur]
        ptr.append(len(cells))
    return np.asarray(ptr, dtype=np.int32), np.asarray(cells, dtype=np.int32)
