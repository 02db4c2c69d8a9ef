import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, torch, tds_amd, oraclelib
from tds_amd import hip_backend
m=tds_amd.load_model("humanoid"); nq,nd=m.dof_q,m.dof_qd
xs=np.load("/root/repo/tmp_x13.npy")
sim=hip_backend.HipSim(m, xs.shape[1])
for t in range(xs.shape[0]):
    x=xs[t]
    y=sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    yo=oraclelib.step(m,x)
    err=np.abs(y-yo)/np.maximum(np.abs(yo),1e-3)
    e,c=np.unravel_index(np.argmax(err),err.shape)
    if err.max()>1e-9:
        print("t",t,"max err %.2e"%err.max(),"env",e,"col",c,"(nq+nd=%d)"%(nq+nd), "gpu",y[e,c],"oracle",yo[e,c])
        bad=np.where(err[e]>1e-9)[0]; print("  bad cols",bad[:40])
        d=oraclelib.step_debug(m,x[e]); print("  contacts dist", np.round(d["contacts"][:,9],6))
        print("  quat", x[e,3:7], "norm", np.linalg.norm(x[e,3:7]))
print("done")
