"""Dump the iteration counts (and final rho) of the config-4 batch solve to gpurun_out/mpc_iters.npz (dispatch-order studies)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
u, info = env._calc_MPC_action(0, 0, 0, 30, return_info=True)
torch.cuda.synchronize()
np.savez("gpurun_out/mpc_iters.npz", iters=info["iters"].cpu().numpy(), rho=info["rho"].cpu().numpy(), u=u.cpu().numpy(),
         useq=info["u_seq"].cpu().numpy())
print("saved", float(info["iters"].mean()))
