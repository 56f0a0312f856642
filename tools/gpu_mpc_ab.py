"""A / B timing of the batched MPC solve (config 4: B = 4096, N = 30, osqp defaults) for two builds of the library in ONE process-per-
build run on the same box:  python tools/gpu_mpc_ab.py libA.so libB.so   (first-call order and repeated calls, 12 calls each)"""
import os, subprocess, sys
CODE = """
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
x0, u0 = config4_states(4096)
env = F16Batch(x0, u0, xcg=0.35); env.build_ssr()
for _ in range(3): env._calc_MPC_action(0, 0, 0, 30)
ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); u, info = env._calc_MPC_action(0, 0, 0, 30, return_info=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("repeated ms: min %%.3f median %%.3f | iters mean %%.2f | u checksum %%.12e" %% (min(ts) * 1e3, float(np.median(ts)) * 1e3, float(info["iters"].mean()), float(u.double().sum())))
"""
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
    for so in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", CODE % REPO], env=dict(os.environ, F16HIP_SO=os.path.abspath(so)), capture_output=True, text=True, timeout=300)
        print(os.path.basename(so), r.stdout.strip()[-200:], r.stderr.strip()[-200:] if r.returncode else "", flush=True)
