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
