"""Diagnostic: per-phase cycles of k_mpc_wave from a -DF16_EXP_STAMPW build (run on the GPU box).
usage: F16HIP_SO=build/libf16hip_stampw.so python tools/gpu_wave_stamps.py [B]"""
import os, sys
sys.path.insert(0, ".")
os.environ["F16_MPC_DISPATCH_ORDER"] = "0"                    # workgroup 0 = aircraft 0
import numpy as np
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
for _ in range(2):
    u, info = env._calc_MPC_action(0, 0, 0, 30, return_info=True)
    torch.cuda.synchronize()
s = info["u_seq"][0, 60:72].cpu().numpy()
e = info["u_seq"][0, 72:79].cpu().numpy()      # 9..15: test cycles, factorise cycles, #tests, #factorisations, run_iterations cycles, other, -
it = s[8]
names = ["A stage1+rhs", "sync", "B matvec", "sync+xt+sync", "C stage3", "projection", "w+sync", "loop+primal test"]
print(f"aircraft 0: {it:.0f} iterations (info: {float(info['iters'][0]):.0f}); cycles per iteration:")
for n, v in zip(names, s[:8]):
    print(f"   {n:14s} {v / it:8.0f}")
print(f"   total          {s[:8].sum() / it:8.0f}")
print(f"tests: {e[2]:.0f} x {e[0] / max(e[2], 1):.0f} cycles; factorisations: {e[3]:.0f} x {e[1] / max(e[3], 1):.0f} cycles; run_iterations calls total {e[4]:.0f} cycles; other {e[5]:.0f}; whole kernel (workgroup 0) {s[11]:.0f} cycles")
t = info["u_seq"][0, 80:85].cpu().numpy()
print("termination test, cycles per test: P x %.0f | W y + G load %.0f | A x, A'y %.0f | norms + reductions %.0f | decision + w %.0f" % tuple(t / max(e[2], 1)))
f = info["u_seq"][0, 85:88].cpu().numpy()
print("factorisation, cycles per call: K from P and A'WA %.0f | sweep (24 pivot blocks) %.0f | scatter + A-block image %.0f" % tuple(f / max(e[3], 1)))
