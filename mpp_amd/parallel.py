"""Host-side helpers for the multi-GPU run: one process per GPU, soil columns sharded across ranks.

The reference has no counterpart (all its solver objects live on PETSC_COMM_SELF,
src/mpp/mpp/MultiPhysicsProbBaseType.F90:1132,1154,1193; an ELM run gives every MPI rank its own columns and its
own SNES).  Here the ranks form ONE global Newton-Krylov solve: residual, Jacobian, SpMV and the per-column
preconditioner need no neighbour data (columns are independent), and the only coupling is the sum of a few fp64
scalars per reduction, done inside libmppgpu.so with an RCCL all-reduce on the solver's stream.

These helpers only move the 128-byte RCCL unique id and a few timings between hosts; they work with any
torch.distributed backend (gloo on CPU for tests, nccl = RCCL on GPUs).
"""
from __future__ import annotations

import os
import socket
import struct
import time


def shard_columns(ncols_global: int, world: int, rank: int):
    """Contiguous block partition of the column set: returns (first_column, ncols_local).  Ranks < remainder get
    one extra column, so shard sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(ncols_global), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def broadcast_unique_id(dist, rank: int, make_id):
    """Rank 0 creates the RCCL unique id (make_id() -> 128 bytes); every rank returns the same bytes."""
    ids = [make_id() if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(ids, src=0)
    if not isinstance(ids[0], (bytes, bytearray)) or len(ids[0]) != 128:
        raise RuntimeError("unique id broadcast failed")
    return bytes(ids[0])


def max_over_ranks(dist, value: float) -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value: float) -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class HostGroup:
    """Minimal host-side process group over TCP (rank 0 is the hub): barrier, byte broadcast and small fp64
    all-reduces.  bench.py uses it instead of torch.distributed so that a rank process loads exactly one ROCm
    runtime (importing torch after libmppgpu.so mixes torch's bundled ROCm libraries with the system ones and
    crashes at exit).  The rendezvous reads the variables torch.distributed.run exports: RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT (the hub listens on MASTER_PORT + 1 .. + 16; the agent's own store owns MASTER_PORT).
    Data-path collectives never go through here: those are RCCL all-reduces inside the library."""

    MAGIC = b"MPPGPU-HOSTGROUP-1"

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", "29500")) + 1 if port is None else int(port)
        self.peers = []
        self.sock = None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = None
            for p in range(base, base + 16):
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("HostGroup: no free rendezvous port in [%d, %d)" % (base, base + 16))
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                hello = self._recv_exact(c, len(self.MAGIC) + 4)
                if hello[:len(self.MAGIC)] != self.MAGIC:
                    c.close()
                    continue
                r = struct.unpack("<i", hello[len(self.MAGIC):])[0]
                c.sendall(self.MAGIC + struct.pack("<i", self.world))
                got[r] = c
            srv.close()
            self.peers = [got[r] for r in sorted(got)]
        else:
            deadline = time.time() + timeout
            while self.sock is None:
                for p in range(base, base + 16):
                    try:
                        c = socket.create_connection((addr, p), timeout=5.0)
                        c.settimeout(10.0)     # a foreign listener on this port must not stall the scan
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.sendall(self.MAGIC + struct.pack("<i", self.rank))
                        reply = self._recv_exact(c, len(self.MAGIC) + 4)
                        if reply[:len(self.MAGIC)] == self.MAGIC and \
                                struct.unpack("<i", reply[len(self.MAGIC):])[0] == self.world:
                            c.settimeout(timeout)
                            self.sock = c
                            break
                        c.close()
                    except (OSError, RuntimeError):
                        pass
                if self.sock is None:
                    if time.time() > deadline:
                        raise RuntimeError("HostGroup: rendezvous with rank 0 timed out")
                    time.sleep(0.2)

    @staticmethod
    def _recv_exact(c, n):
        buf = b""
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise RuntimeError("HostGroup: peer closed the connection")
            buf += chunk
        return buf

    def broadcast_bytes(self, payload):
        """payload from rank 0 is returned on every rank"""
        if self.world == 1:
            return bytes(payload)
        if self.rank == 0:
            msg = struct.pack("<q", len(payload)) + bytes(payload)
            for c in self.peers:
                c.sendall(msg)
            return bytes(payload)
        n = struct.unpack("<q", self._recv_exact(self.sock, 8))[0]
        return self._recv_exact(self.sock, n)

    def allreduce(self, values, op="sum"):
        vals = [float(v) for v in values]
        if self.world == 1:
            return vals
        k = len(vals)
        if self.rank == 0:
            rows = [vals] + [list(struct.unpack("<%dd" % k, self._recv_exact(c, 8 * k))) for c in self.peers]
            red = [max(col) if op == "max" else (min(col) if op == "min" else sum(col)) for col in zip(*rows)]
            out = struct.pack("<%dd" % k, *red)
            for c in self.peers:
                c.sendall(out)
            return red
        self.sock.sendall(struct.pack("<%dd" % k, *vals))
        return list(struct.unpack("<%dd" % k, self._recv_exact(self.sock, 8 * k)))

    def barrier(self):
        self.allreduce([0.0])

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None


# ------------------------------------------------------------------------------------------------------------
# 3-D meshes split across ranks: owned cells + ghost cells + halo plan (mppgpu_set_ghost_cells / mppgpu_set_halo)
# ------------------------------------------------------------------------------------------------------------
class MeshPart:
    """one rank's piece of a global mesh: `mesh` (owned cells in global order first, ghost cells last, grouped by owner),
    `l2g` (0-based global id of every local cell), `nghost`, and the halo plan arrays of mppgpu_set_halo"""

    def __init__(self, mesh, l2g, nghost, peer, send_ptr, send_cells, recv_count):
        self.mesh, self.l2g, self.nghost = mesh, l2g, nghost
        self.peer, self.send_ptr, self.send_cells, self.recv_count = peer, send_ptr, send_cells, recv_count
        self.nowned = mesh.ncells_local - nghost


def column_of_cells(mesh):
    """0-based soil-column number of every cell, from the vertical chains of the internal connection set (any cell numbering; a
    connection is vertical when its unit vector is mostly along z -- the library's own rule, solver.cpp `is_vertical`).  Columns are
    numbered in the order of their lowest cells' ids (for MeshCreate's x-fastest numbering: jj outer, ii inner).  A cell without a
    vertical connection is a column of its own."""
    import numpy as np
    n = mesh.ncells_local
    cs = mesh.internal_connections()
    up, dn = np.asarray(cs.id_up, dtype=np.int64) - 1, np.asarray(cs.id_dn, dtype=np.int64) - 1
    if cs.unit_vec is None:
        vert = np.ones(up.size, dtype=bool)
    else:
        uv = np.asarray(cs.unit_vec, dtype=np.float64)
        vert = np.abs(uv[:, 2]) > 0.5 * np.sqrt((uv * uv).sum(axis=1))
    below = np.arange(n, dtype=np.int64)           # the cell under this one in its chain (itself at the chain's lower end)
    below[dn[vert]] = up[vert]
    root = below
    while True:                                    # pointer jumping: log2(longest chain) rounds
        nxt = root[root]
        if np.array_equal(nxt, root):
            break
        root = nxt
    _, col = np.unique(root, return_inverse=True)  # roots ascending = columns in the order of their lowest cells
    return col.astype(np.int64)


def partition_mesh(mesh, nranks, mode="auto"):
    """Split `mesh` over `nranks` ranks.

    mode "columns" (the default whenever the mesh has a vertical connection): ranks own WHOLE soil columns -- the column set,
    in the order of the columns' lowest cells, is cut into `nranks` contiguous blocks (x-y blocks: for MeshCreate's x-fastest
    numbering a block is a range of jj) -- which is how the reference distributes a 3-D mesh (MeshCreateFromCLMCols,
    src/mpp/dtypes/MeshType.F90:293-460: a rank holds ncols_local columns of nlevgrnd cells and its ghosts are whole ghost columns,
    :362 ncells_ghost = ncols_ghost*nlevgrnd) and what SURVEY.md section 8(e) asks for.  No soil column is cut: the strong vertical
    couplings stay inside a rank's block of the rank-local preconditioner, and the halo of a 7-point grid is nz x edge cells per
    neighbour.
    mode "ranges": contiguous id ranges (z-slabs for the x-fastest numbering) -- kept for the isotropic thermal cube
    (workloads.thermal_mms_rank builds exactly these slabs without the global mesh).

    A cell is a ghost on rank r when an internal connection joins it to a cell r owns; connections with at least one owned end are
    kept (in the global connection order); owned cells come first in ascending global id, ghosts after them grouped by owner in
    ascending id; both neighbours list shared cells in ascending global id, which makes rank q's send list match the order of its
    cells among rank r's ghosts."""
    import numpy as np
    from . import mesh as M
    n = mesh.ncells_local
    cs = mesh.internal_connections()
    up, dn = np.asarray(cs.id_up) - 1, np.asarray(cs.id_dn) - 1
    if mode not in ("auto", "columns", "ranges"):
        raise ValueError("partition_mesh: mode must be auto, columns or ranges")
    owner = np.empty(n, dtype=np.int64)
    col = None
    if mode != "ranges":
        col = column_of_cells(mesh)
        if mode == "auto" and int(col.max()) + 1 == n:
            col = None                                 # no vertical connection anywhere: columns are single cells = id ranges
    if col is not None:
        ncol = int(col.max()) + 1
        if ncol < nranks:
            raise ValueError("partition_mesh: %d columns cannot be split over %d ranks" % (ncol, nranks))
        col_owner = np.empty(ncol, dtype=np.int64)
        for r in range(nranks):
            c0, cnt = shard_columns(ncol, nranks, r)
            col_owner[c0:c0 + cnt] = r
        owner[:] = col_owner[col]
    else:
        for r in range(nranks):
            c0, cnt = shard_columns(n, nranks, r)
            owner[c0:c0 + cnt] = r
    bounds = [np.nonzero(owner == r)[0] for r in range(nranks)]
    parts = []
    for r, own_g in enumerate(bounds):
        mine_up, mine_dn = owner[up] == r, owner[dn] == r
        keep = mine_up | mine_dn
        ghost_g = np.unique(np.concatenate([dn[mine_up & ~mine_dn], up[mine_dn & ~mine_up]]))
        ghost_g = ghost_g[np.lexsort((ghost_g, owner[ghost_g]))]          # grouped by owner, ascending id inside a group
        l2g = np.concatenate([own_g, ghost_g])
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[l2g] = np.arange(l2g.size)
        peers = np.unique(owner[ghost_g])
        recv_count = [int((owner[ghost_g] == q).sum()) for q in peers]
        send_ptr, send_cells = [0], []
        for q in peers:   # my cells that rank q sees as ghosts = my cells joined to one of q's cells
            mineq = np.unique(np.concatenate([up[mine_up & (owner[dn] == q)], dn[mine_dn & (owner[up] == q)]]))
            send_cells.extend((g2l[mineq] + 1).tolist())
            send_ptr.append(len(send_cells))
        pick = lambda a: None if a is None else np.asarray(a)[l2g].copy()
        lm = M.Mesh(name=mesh.name, ncells_local=int(l2g.size), nlev=mesh.nlev, orientation=mesh.orientation)
        lm.x, lm.y, lm.z, lm.dx, lm.dy, lm.dz = (pick(a) for a in (mesh.x, mesh.y, mesh.z, mesh.dx, mesh.dy, mesh.dz))
        lm.area_xy, lm.vol, lm.is_active = pick(mesh.area_xy), pick(mesh.vol), pick(mesh.is_active)
        sel = np.nonzero(keep)[0]
        lcs = M.ConnectionSet(g2l[up[sel]] + 1, g2l[dn[sel]] + 1, np.asarray(cs.dist_up)[sel], np.asarray(cs.dist_dn)[sel],
                              np.asarray(cs.area)[sel], np.asarray(cs.unit_vec)[sel], None if cs.itype is None else np.asarray(cs.itype)[sel])
        lm.intrn_conn_sets.append(lcs)
        lm.ncols, lm.ordering = mesh.ncols, mesh.ordering
        lm.nghost = int(ghost_g.size)
        lm.halo = (np.asarray(peers, dtype=np.int32), np.asarray(send_ptr, dtype=np.int32), np.asarray(send_cells, dtype=np.int32),
                   np.asarray(recv_count, dtype=np.int32))
        part = MeshPart(lm, l2g, int(ghost_g.size), *lm.halo)
        part.mode = "columns" if col is not None else "ranges"
        part.ncols_owned = int(np.unique(col[own_g]).size) if col is not None else 0
        parts.append(part)
    return parts
