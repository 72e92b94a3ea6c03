import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import problems as P
from mpp_amd import constants as C
import test_gpu_fuzz as TF
prob, x, opts, (oksp, opc), what = TF.draw3d(34)
print(what)
gs = {c3: P.to_gpu(prob, dict(opts, cols3d=c3)) for c3 in (0, 1)}
for c3, g in gs.items():
    print(c3, g.soe.StepDT(prob.dt, 1), g.soe.counters())
p = {c3: g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1) for c3, g in gs.items()}
print("pressure diff after step 1", np.abs(p[0] - p[1]).max(), np.isnan(p[1]).any())
out = {}
for c3, g in gs.items():
    g.soe.SetDtime(prob.dt); g.soe.PreSolve()
    f = g.soe.ComputeResidual(p[0])
    rp, cj, val = g.soe.ComputeJacobian(p[0])
    out[c3] = (f, val)
    print(c3, "F nan", np.isnan(f).any(), "J nan", np.isnan(val).any(), "max|F|", np.abs(f).max(), "max|J|", np.abs(val).max())
print("F diff", np.abs(out[0][0] - out[1][0]).max(), "J diff", np.abs(out[0][1] - out[1][1]).max())
bad = np.nonzero(np.abs(out[0][1] - out[1][1]) > 1e-9 * np.abs(out[0][1]).max())[0]
print("bad J entries", bad[:20], out[0][1][bad[:10]], out[1][1][bad[:10]])
badf = np.nonzero(np.abs(out[0][0] - out[1][0]) > 1e-9 * (np.abs(out[0][0]).max() + 1e-30))[0]
print("bad F rows", badf[:20], out[0][0][badf[:10]], out[1][0][badf[:10]])
for c3, g in gs.items():
    print(c3, "step 2:", g.soe.StepDT(prob.dt, 2), g.soe.counters())
