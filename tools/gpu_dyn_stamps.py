"""Diagnostic: per-wave cycles of the quad rollout from a -DF16_EXP_STAMPQ build (run on the GPU box).
usage: F16HIP_SO=build/libf16hip_stampq.so python tools/gpu_dyn_stamps.py [B]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config2_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, u0 = config2_states(B)
env = F16Batch(x0, u0)
T = 1000
traj = env.rollout(T, traj_every=1)
torch.cuda.synchronize()
d = traj[2].reshape(-1)[:16].cpu().numpy().reshape(4, 4) / T
np.set_printoptions(linewidth=200, precision=0, suppress=True)
print("cycles per step; rows = waves (long, lat, trig/forces, atmos/act); cols = second half+publish, barrier A, first half, barrier B")
print(d, d.sum(1))
