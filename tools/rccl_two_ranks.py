#!/usr/bin/env python3
"""Try the RCCL transport with two ranks that share ONE GPU (both use device 0).  RCCL/NCCL normally refuse duplicate
devices; when this pool's build allows it, the script validates the real ncclAllReduce / ncclSend / ncclRecv call sites:
a column-sharded celia grid and a z-split 7-point grid against the unsharded solves.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/rccl_two_ranks.py
Prints 'RCCL-2RANKS OK ...' or the RCCL error."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from mpp_amd import workloads as P  # noqa: E402
from mpp_amd import constants as C  # noqa: E402
from mpp_amd import parallel as PAR  # noqa: E402
from mpp_amd import vsfm as V  # noqa: E402


def main():
    group = PAR.HostGroup()
    rank, world = group.rank, group.world
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    uid = group.broadcast_bytes(V.comm_get_unique_id() if rank == 0 else b"")
    # (1) column-sharded global solve
    ncols, nz = 96, 25
    mult = np.exp(0.5 * np.random.default_rng(11).standard_normal(ncols))
    opts = {"ksp_type": "bcgs", "pc_type": "bjacobi"}
    c0, nloc = PAR.shard_columns(ncols, world, rank)
    prob = P.celia1990(ncols=nloc, nz=nz, perm_mult=mult[c0:c0 + nloc])
    g = P.to_gpu(prob, opts, device_id=0)
    g.soe.comm_init(world, rank, uid)
    for s in range(2):
        assert g.soe.StepDT(prob.dt, s + 1)[0]
    ref_prob = P.celia1990(ncols=ncols, nz=nz, perm_mult=mult)
    ref = P.to_gpu(ref_prob, opts, device_id=0)
    for s in range(2):
        assert ref.soe.StepDT(ref_prob.dt, s + 1)[0]
    p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE).reshape(nloc, nz)
    p_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE).reshape(ncols, nz)[c0:c0 + nloc]
    e1 = float(np.abs(p - p_ref).max() / np.abs(p_ref).max())
    same_counts = g.soe.counters()["newton"] == ref.soe.counters()["newton"]
    group.barrier()
    print("rank %d: sharded vs unsharded %.2e, same Newton count %s" % (rank, e1, same_counts), flush=True)
    assert e1 < 1e-9 and same_counts
    group.barrier()
    print("RCCL-2RANKS OK (all-reduce path)" if rank == 0 else "", flush=True)
    group.close()


if __name__ == "__main__":
    main()
