"""Run only the config-5 closed loop as ONE launch (f16_rollout_mpc; for rocprofv3).  usage: gpu_config5_only.py [B] [T] [N]"""
import sys
sys.path.insert(0, ".")
import time
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
env.prepare_MPC(N)
env.rollout_MPC(2, 0.0, 0.0, 0.0, N)
env.reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
traj, info = env.rollout_MPC(T, 0.0, 0.0, 0.0, N, traj_every=1, return_info=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("aircraft-steps/s %.0f  (%.2f ms per step)  iters mean %.1f  flagged %d" % (B * T / dt, dt / T * 1e3, float(info["iters"].float().mean()),
      int(((env.status & (32 | 128)) != 0).sum())))
