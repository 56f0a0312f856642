"""The horizon sweep alone (for rocprofv3 --kernel-trace): python tools/gpu_sweep_only.py [B] [max_hzn]"""
import sys, time
sys.path.insert(0, ".")
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 150
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
env._calc_MPC_action(0, 0, 0, 33); torch.cuda.synchronize()
t0 = time.perf_counter()
sw, inf = env._calc_constr_checking_hzn(max_hzn=H, return_info=True)
torch.cuda.synchronize()
print(f"_calc_constr_checking_hzn({H}) for B = {B}: {time.perf_counter() - t0:.2f} s", flush=True)
t0 = time.perf_counter()
sw2 = env._calc_constr_checking_hzn(max_hzn=H)
torch.cuda.synchronize()
print(f"again (queue ordered by the first sweep's iteration counts): {time.perf_counter() - t0:.2f} s; identical {bool(torch.equal(torch.nan_to_num(sw, nan=1e300), torch.nan_to_num(sw2, nan=1e300)))}", flush=True)
it = inf["iters"].cpu().numpy()
for N in range(1, H + 1):
    print(N, int(it[N - 1].max()), float(it[N - 1].mean()))
