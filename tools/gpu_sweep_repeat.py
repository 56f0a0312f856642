"""Diagnostic: the horizon sweep four times in one process (first call, then calls with the queue ordered by history).
usage: python tools/gpu_sweep_repeat.py bench|script|info"""
import sys, time
sys.path.insert(0, ".")
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
mode = sys.argv[1]
dev = torch.device("cuda", 0)
x0, u0 = config4_states(64)
env = F16Batch(x0, u0, xcg=0.35, device=dev) if mode == "bench" else F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
env._calc_MPC_action(0, 0, 0, 33)
torch.cuda.synchronize()
for k in range(4):
    t0 = time.perf_counter()
    if mode == "info" and k == 0:
        sw, keep = env._calc_constr_checking_hzn(max_hzn=150, return_info=True)
    else:
        sw = env._calc_constr_checking_hzn(max_hzn=150)
    torch.cuda.synchronize()
    print(mode, k, round(time.perf_counter() - t0, 3), flush=True)
