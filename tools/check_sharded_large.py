#!/usr/bin/env python3
"""One-off check of the multi-rank code paths at bench scale on ONE GPU: the 1024 x NY column grid as one handle and as R
in-process ranks (threads, mppgpu_comm_init_local) must take the same Newton / Krylov decisions and give the same pressures.
usage: python tools/check_sharded_large.py [NY=256] [R=2] [steps=2]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from mpp_amd import workloads as P  # noqa: E402
from mpp_amd import constants as C  # noqa: E402
from mpp_amd import parallel as PAR  # noqa: E402
from mpp_amd import vsfm as V  # noqa: E402

ny = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ncols, nz = 1024 * ny, 100
opts = {"ksp_type": "bcgs", "pc_type": "bjacobi"}


def run(prob, g, out):
    for s in range(steps):
        for b, (_t, _cs, vals) in enumerate(prob.bcs):
            g.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, b + 1, vals)
        out.append(g.soe.StepDT(prob.dt, s + 1))


ref_prob = bench.build_problem(ncols, nz, 0, ncols)
ref = P.to_gpu(ref_prob, opts)
ref_out = []
run(ref_prob, ref, ref_out)
p_ref = ref.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1).reshape(ncols, nz)
c_ref = ref.soe.counters()
del ref
shards = []
for r in range(nranks):
    c0, nloc = PAR.shard_columns(ncols, nranks, r)
    prob = bench.build_problem(ncols, nz, c0, nloc)
    shards.append((c0, nloc, prob, P.to_gpu(prob, opts)))
V.comm_init_local([s[3].soe for s in shards])
outs = [[] for _ in range(nranks)]
errs = []


def worker(r):
    try:
        run(shards[r][2], shards[r][3], outs[r])
    except Exception as e:  # noqa: BLE001
        errs.append(e)


th = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
for t in th:
    t.start()
for t in th:
    t.join()
assert not errs, errs
ok = True
for r in range(nranks):
    c0, nloc, _p, g = shards[r]
    c = g.soe.counters()
    same = outs[r] == ref_out and c["newton"] == c_ref["newton"] and c["linear"] == c_ref["linear"]
    p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1).reshape(nloc, nz)
    err = float(np.abs(p - p_ref[c0:c0 + nloc]).max() / np.abs(p_ref).max())
    print("rank %d: steps %s newton %d/%d linear %d/%d max rel dp %.2e" % (r, outs[r] == ref_out, c["newton"], c_ref["newton"],
                                                                          c["linear"], c_ref["linear"], err), flush=True)
    ok &= same and err <= 1e-9
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
