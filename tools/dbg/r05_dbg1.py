import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import problems as P
from mpp_amd import constants as C
import test_gpu_fuzz as TF

for seed in (25, 34):
    for c3 in (0, 1):
        prob, x, opts, (oksp, opc), what = TF.draw3d(seed)
        opts = dict(opts, cols3d=c3)
        g = P.to_gpu(prob, opts)
        o = P.to_oracle(prob); o.set_solver(oksp, opc)
        res_g = [g.soe.StepDT(prob.dt, s + 1) for s in range(2)]
        res_o = [tuple(o.step_dt(prob.dt, s + 1)) for s in range(2)]
        print("seed", seed, "cols3d", c3, what, res_g, res_o, g.soe.counters(), o.counters(), g.soe.assembly_info()["columns_plus_lateral_assembly"], flush=True)

for sat in (False, True):
    for nx in (20, 40, 80, 160):
        for tol in ({}, {"snes_rtol": 1e-10, "snes_stol": 1e-12, "ksp_rtol": 1e-10}):
            pr = P.vsfm_mms(nx, sat)
            h = P.to_gpu(pr, tol)
            h.soe.PreStepDT()
            r = h.soe.StepDT(pr.dt, 1)
            err = float(np.abs(h.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1) - pr.exact).max())
            print("mms sat", sat, "nx", nx, tol, r, h.soe.counters(), err, flush=True)
