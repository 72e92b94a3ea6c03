import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import problems as P
from mpp_amd import constants as C
for opts in ({}, {"pc_type": "ilu"}, {"pc_type": "ilu", "ilu_box": 0}, {"pc_type": "ilu", "ilu_box": 0, "ilu_graph": 0}, {"setup_device": 0}, {"pc_type": "bjacobi"}, {"pc_type": "jacobi"}):
    pr = P.vsfm_mms(20)
    h = P.to_gpu(pr, opts)
    h.soe.PreStepDT()
    r = h.soe.StepDT(pr.dt, 1)
    print(opts, r, h.soe.counters(), h.soe.assembly_info(), h.soe.pc_info(), flush=True)
    # PCApply exactness: z = M^-1 (A v) should be v for an exact factorisation
    v = np.random.default_rng(1).standard_normal(20)
    h.soe.SetDtime(pr.dt); h.soe.PreSolve()
    h.soe.ComputeJacobian(pr.exact)
    z = h.soe.PCApply(h.soe.MatMult(v))
    print("   |M^-1 A v - v| =", float(np.abs(z - v).max()), flush=True)
