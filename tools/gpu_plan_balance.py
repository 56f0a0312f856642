"""Diagnostic: how far is a cold plan solve (B = 8192) from the perfectly balanced schedule?  Compares the measured time
of one plan solve with sum_i t_i / 256 CUs, t_i = t0 + iters_i * t_it + tests_i * t_test from the per-aircraft iteration counts."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
env.prepare_MPC(30)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
full = t(lambda: env._calc_MPC_action(0, 0, 0, 30, use_plan=True))
u, info = env._calc_MPC_action(0, 0, 0, 30, use_plan=True, return_info=True)
it = info["iters"].cpu().numpy()
vals, cnt = np.unique(it, return_counts=True)
print("plan solve ms", full, " iteration histogram", dict(zip(vals.tolist(), cnt.tolist())))
# per-aircraft model (us): prologue + iterations + tests (every 25)
for t0 in (5.0, 10.0, 15.0):
    ti = t0 + it * 1.17 + np.ceil(it / 25) * 4.0
    print("t0 = %4.1f us: balanced bound %.3f ms (mean %.1f us, max %.1f us)" % (t0, ti.sum() / 256 / 1e3, ti.mean(), ti.max()))
# the same solve with the aircraft sorted by descending iteration count (longest first)
order = np.argsort(-it, kind="stable")
env2 = F16Batch(x0[order], u0[order], xcg=0.35)
env2.build_ssr(); env2.prepare_MPC(30)
print("sorted longest-first: plan solve ms", t(lambda: env2._calc_MPC_action(0, 0, 0, 30, use_plan=True)))
order = np.argsort(it, kind="stable")
env3 = F16Batch(x0[order], u0[order], xcg=0.35)
env3.build_ssr(); env3.prepare_MPC(30)
print("sorted shortest-first: plan solve ms", t(lambda: env3._calc_MPC_action(0, 0, 0, 30, use_plan=True)))
