"""Run only the batched MPC solve (for rocprofv3).  usage: gpu_mpc_only.py [B] [N] [max_iter] [check_every]"""
import sys
sys.path.insert(0, ".")
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
ce = int(sys.argv[4]) if len(sys.argv) > 4 else 25
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
for _ in range(3):
    u, info = env._calc_MPC_action(0, 0, 0, N, settings=dict(max_iter=mi, check_every=ce), return_info=True)
torch.cuda.synchronize()
print("iters", float(info["iters"].min()), float(info["iters"].median()), float(info["iters"].max()))
