#!/usr/bin/env python3
"""Diagnostic: where does the 1024x1024x100 heterogeneous answer differ from per-column oracle solves, and why?
usage: diag_answer.py [KEY=VALUE ...]   (mppgpu options), env STEPS (default 2)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from mpp_amd import constants as C, workloads as P
from oracle import adapters as A

nx = int(os.environ.get("NX", 1024)); ny = int(os.environ.get("NY", 1024)); nz = 100
steps = int(os.environ.get("STEPS", 2))
ncols = nx * ny
mult = bench.perm_multipliers(ncols)
prob = bench.build_problem(ncols, nz, 0, ncols)
opts = {"ksp_type": "bcgs", "pc_type": "bjacobi"}
opts.update(kv.split("=", 1) for kv in sys.argv[1:])
g = P.to_gpu(prob, opts)
hist = []
for s in range(steps):
    c0 = g.soe.counters()
    conv, reason = g.soe.StepDT(prob.dt, s + 1)
    c1 = g.soe.counters()
    hist.append(g.soe.step_history())
    print("accepted sub-steps", hist[-1])
    print("step", s + 1, conv, reason, "newton", c1["newton"] - c0["newton"], "linear", c1["linear"] - c0["linear"],
          "cuts", c1["time_cuts"] - c0["time_cuts"], flush=True)
p = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1).reshape(ncols, nz)
order = np.argsort(mult)
pick = np.unique(np.concatenate([order[:24], order[-24:], np.random.default_rng(7).choice(ncols, 64, replace=False)]))
worst = []
for c in pick:
    one = P.celia1990(ncols=1, nz=nz, perm_mult=mult[c:c + 1])
    o = A.to_oracle(one, with_blocks=False)
    o.set_solver(C.KSP_GMRES, C.PC_ILU0, ksp_rtol=1e-12, snes_rtol=1e-13, snes_stol=1e-14)
    k = 0
    for subs in hist:
        for dts in subs:
            k += 1
            o.step_dt(dts, k)
    po = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    d = np.abs(p[c] - po)
    worst.append((d.max(), int(c), float(mult[c]), int(d.argmax()), o.counters()["newton"], o.counters()["time_cuts"]))
worst.sort(reverse=True)
print("options", opts)
for w in worst[:12]:
    print("err %.3e Pa  col %d  mult %.3f  level %d  oracle newton %d cuts %d" % w)
print("median err %.3e" % np.median([w[0] for w in worst]), " #cols err>1Pa:", sum(w[0] > 1 for w in worst), "of", len(worst), flush=True)
