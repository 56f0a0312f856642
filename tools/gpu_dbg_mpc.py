"""Debug helper: perturbed-batch MPC solve, GPU vs the numpy twin, per aircraft."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config2_states
from oracle import mpc_oracle as mo

B, N = 192, 30
x0, u0 = config2_states(B, seed=4)
env = F16Batch(x0, u0, xcg=0.35)
Ad, Bd, Cd = env.build_ssr()
dem = np.array([0.05, -0.02, 0.01])
u, info = env._calc_MPC_action(*dem, N, return_info=True)
st = info["status"].cpu().numpy(); it = info["iters"].cpu().numpy(); rho = info["rho"].cpu().numpy()
print("status counts", {int(k): int((st == k).sum()) for k in np.unique(st)})
print("iters", np.percentile(it, [0, 50, 90, 100]))
Adh, Bdh, Cdh = (t.t().cpu().numpy() for t in (Ad, Bd, Cd))
bad = np.nonzero(st == 64)[0][:4].tolist()
for b in [0, 17, 101, 150, 26, 29] + bad:
    P, q, A, l, uu = mo.mpc_qp(x0[b], Adh[b].reshape(9, 9), Bdh[b].reshape(9, 3), Cdh[b].reshape(9, 9), N, 0.001, *dem)
    ref = mo.admm_osqp(P, q, A, l, uu, drop_unbounded_rows=True)
    print(b, "gpu st", st[b], "it", it[b], "rho", rho[b], "rp", float(info["r_prim"][b]), "rd", float(info["r_dual"][b]), "| ref it", ref["iters"], "rho", ref["rho"], "conv", ref["converged"],
          "inf", ref["infeasible"], "rp", ref["r_prim"], "rd", ref["r_dual"], "dx", np.abs(u[b].cpu().numpy() - ref["x"][:3]).max())
