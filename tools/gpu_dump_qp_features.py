"""Dump what a first-call dispatch order could be predicted from: the config-4 batch (4096 aircraft, N = 30), its frozen models, the
iteration counts of its solves and the solver's per-solve figures -> gpurun_out/qp_features.npz (studied offline: DESIGN.md 7)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B, N = 4096, 30
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
Ad, Bd, Cd = env.build_ssr()
u, info = env._calc_MPC_action(0.0, 0.0, 0.0, N, return_info=True)
m = lambda t, r, c: t.t().reshape(B, r, c).cpu().numpy()
np.savez_compressed("gpurun_out/qp_features.npz", x0=x0, iters=info["iters"].cpu().numpy(), rho=info["rho"].cpu().numpy(), r_prim=info["r_prim"].cpu().numpy(),
                    Ad=m(Ad, 9, 9), Bd=m(Bd, 9, 3), Cd=m(Cd, 9, 9), u_seq=info["u_seq"].cpu().numpy())
print("saved", float(info["iters"].mean()))
