"""Diagnostic: time of the MPC build kernel + factorisation only (max_iter=0).  usage: [B]"""
import sys, time
sys.path.insert(0, ".")
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("N=30 setup+factor ms", t(lambda: env._calc_MPC_action(0, 0, 0, 30, settings=dict(max_iter=0))))
