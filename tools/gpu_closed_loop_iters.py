"""Diagnostic: per step of the config-5 closed loop (B aircraft, reference settings, cold start), the distribution of ADMM iteration counts
and the wall time of the step.  usage: python tools/gpu_closed_loop_iters.py [B] [steps]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
dem = torch.zeros((3, B), dtype=torch.float64, device=env.device)
rows = []
for k in range(T):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cmd = env._calc_MPC_action(dem, None, None, 30)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    env._u[1:4] = cmd.t(); env.rollout(1)
    it = env.last_iters.cpu().numpy(); st = env.last_status.cpu().numpy()
    rows.append((k, 1e3 * (t1 - t0), it.mean(), it.max(), int((it > 1000).sum()), int((it > 2000).sum()), int((it > 5000).sum()), int((it >= 40000).sum()),
                 int(((st & 128) != 0).sum()), int(((st & 16) != 0).sum())))
print("step  ms     mean   max   >1000 >2000 >5000 =40000 infeasible left-envelope")
for r in rows:
    if r[0] < 5 or r[0] % 10 == 9: print("%3d %6.2f %7.1f %6d %5d %5d %5d %5d %5d %5d" % r)
a = np.array(rows)
print("mean ms %.2f; steps whose longest solve exceeds the throughput bound (max iters x 4.1 us > ms): %d" % (a[:, 1].mean(), int((a[:, 3] * 4.1e-3 > 0.9 * a[:, 1]).sum())))
