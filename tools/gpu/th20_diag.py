import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, problems as P
from mpp_amd import constants as C
from test_gpu_th import _state
seed=20
rng = np.random.default_rng(9000 + seed)
ncols, nz = int(rng.choice([1, 3, 70, 131])), int(rng.choice([2, 5, 9, 20]))
asm=str(rng.choice(["general", "columns"])); mixed=bool(rng.random() < 0.5); lp=bool(rng.random() < 0.4)
den=int(rng.choice([C.DENSITY_CONSTANT, C.DENSITY_TGDPB01, C.DENSITY_IFC67])); ie=int(rng.choice([1, 2]))
print(ncols,nz,asm,mixed,lp,den,ie)
r2 = np.random.default_rng(100+seed)
tp = P.th_columns(ncols=ncols, nz=nz, perm_mult=np.exp(0.3 * r2.standard_normal(ncols)), density_type=den, ie_type=ie)
n = tp.mesh.ncells_local
if lp:
    tp.por_pressure_reference = np.full(n, 101325.0) + r2.uniform(-2e4, 2e4, n); tp.por_slope = r2.uniform(1e-7, 6e-7, n)
o = P.to_th_oracle(tp)
for asm in ("general","columns"):
    g = P.to_th_gpu(tp, {"ksp_type": "gmres", "pc_type": "ilu", "assembly": asm})
    x = _state(tp, seed=seed+102, dp=(0.0, 9e4), dt=(0.0, 15.0))
    o.set_dtime(tp.dt); o.pre_solve(); g.soe.SetDtime(tp.dt); g.soe.PreSolve()
    rp, cj, val = o.jacobian(x); rp_g, cj_g, val_g = g.soe.ComputeJacobian(x)
    scale = np.zeros(val.size)
    for i in range(2 * n):
        for (a, b) in ((0, n), (n, 2 * n)):
            sel = np.arange(rp[i], rp[i + 1])[(cj[rp[i]:rp[i + 1]] >= a) & (cj[rp[i]:rp[i + 1]] < b)]
            scale[sel] = np.abs(val[sel]).max()
    bad = np.nonzero(np.abs(val_g - val) > 1e-10 * scale + 1e-300)[0]
    rows = np.repeat(np.arange(2*n), np.diff(rp))
    print(asm, "bad", bad.size)
    for k in bad[:12]:
        i=rows[k]; j=cj[k]
        print(" row",i,"col",j,"block","PT"[i>=n]+"PT"[j>=n],"cell",i%n,"lev",(i%n)%nz,"val_o",val[k],"val_g",val_g[k],"scale",scale[k], "P",x[i%n],"T",x[n+i%n])
