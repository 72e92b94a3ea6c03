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
                        c.settimeout(timeout)
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.sendall(self.MAGIC + struct.pack("<i", self.rank))
                        reply = self._recv_exact(c, len(self.MAGIC) + 4)
                        if reply[:len(self.MAGIC)] == self.MAGIC and \
                                struct.unpack("<i", reply[len(self.MAGIC):])[0] == self.world:
                            self.sock = c
                            break
                        c.close()
                    except OSError:
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
