"""Timing of F16._calc_constr_checking_hzn (env.py:426-436: calc_MPC_action for every horizon N = 1..150), B aircraft.
usage: python tools/gpu_time_hzn_sweep.py [B] [max_hzn]"""
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
for N in (30, 33, 40, 41, 57, 100, 150):
    if N > H:
        continue
    env._calc_MPC_action(0, 0, 0, N); torch.cuda.synchronize()
    t0 = time.perf_counter(); u, info = env._calc_MPC_action(0, 0, 0, N, return_info=True); torch.cuda.synchronize()
    print(f"N={N}: {1e3 * (time.perf_counter() - t0):.1f} ms for {B} aircraft, iterations mean {float(info['iters'].mean()):.0f} max {float(info['iters'].max()):.0f}, status {sorted(set(info['status'].cpu().numpy().tolist()))}", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
sw, inf = env._calc_constr_checking_hzn(max_hzn=H, return_info=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
it = inf["iters"]
print(f"_calc_constr_checking_hzn({H}) for B = {B}: {dt:.2f} s; aircraft-iterations in total {float(it.sum()):.3e}, "
      f"largest {float(it.max()):.0f}; infeasible (NaN command) {int((inf['status'] & 128).ne(0).sum())} of {it.numel()}", flush=True)
for N in (1, 30, 33, 41, 57, 100, 150):
    if N > H:
        continue
    u, i1 = env._calc_MPC_action(0, 0, 0, N, return_info=True)
    same = torch.equal(torch.nan_to_num(sw[:, :, N - 1], nan=1e300), torch.nan_to_num(u, nan=1e300))
    print(f"N={N}: slice == direct call {same}; iterations equal {bool(torch.equal(it[N - 1], i1['iters']))}; status equal "
          f"{bool(torch.equal(inf['status'][N - 1], i1['status']))}", flush=True)
