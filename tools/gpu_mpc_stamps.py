"""Diagnostic: per-phase cycles of k_mpc_fast from a -DF16_EXP_STAMPM build (run on the GPU box).
usage: F16HIP_SO=build/libf16hip_stamp.so python tools/gpu_mpc_stamps.py [B]"""
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
mode = dict(scaling=0, rho=0.0) if (len(sys.argv) > 2 and sys.argv[2] == "builder") else {}
u, info = env._calc_MPC_action(0, 0, 0, 30, settings=dict(max_iter=100, check_every=1000, **mode), return_info=True)
torch.cuda.synchronize()
import os
os.environ["F16_MPC_DISPATCH_ORDER"] = "0"                    # workgroup 0 = aircraft 0
u, info = env._calc_MPC_action(0, 0, 0, 30, settings=dict(max_iter=100, check_every=1000, **mode), return_info=True)
torch.cuda.synchronize()
s = info["u_seq"][0, :48].cpu().numpy().reshape(8, 6)      # aircraft 0 column holds the stamps
np.set_printoptions(linewidth=200, precision=0, suppress=True)
print("cycles per iteration, rows = waves, cols = A, bar1, B, bar2, C, bar3 (s_memtime ticks @100MHz => x24 for 2.4GHz)")
print(s, s.sum(1))
print("factorisation, per wave (work, barrier wait) cycles:", info["u_seq"][0, 50:66].cpu().numpy().reshape(8, 2))
print("last factorisation, wave 0: P loads + gram / assembly (+ trace) / sweep / re-layout cycles:", info["u_seq"][0, 66:70].cpu().numpy())
print("equilibration cycles:", float(info["u_seq"][0, 70]), " iterations stamped:", float(info["u_seq"][0, 71]), " gram product alone (F16_EXP_GRAMSTAMP builds):", float(info["u_seq"][0, 72]))
