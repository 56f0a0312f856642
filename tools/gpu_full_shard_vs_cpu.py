"""A FULL config-5 shard (8,192 aircraft, N = 30) for T closed-loop steps through the one-launch loop (f16_rollout_mpc) against the same
loop on the host cores (oracle.mpc_closed_loop: the C twin with the device's own frozen models) -- every iteration count, every status
word, every command.  Test-side tool (it loads the checker); ~1 minute of CPU time for T = 10.   python tools/gpu_full_shard_vs_cpu.py [T]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
from oracle import mpc_oracle as mo
B, N = 8192, 30
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35); env.build_ssr(); env.prepare_MPC(N)
_, info = env.rollout_MPC(T, 0.0, 0.0, 0.0, N, traj_every=1, return_info=True)
torch.cuda.synchronize()
m = lambda t, r, c: t.t().reshape(B, r, c).cpu().numpy()
Ad, Bd, Cd = (m(a, r, c) for a, (r, c) in zip(env.ssr, ((9, 9), (9, 3), (9, 9))))
ora = mo.COracle()
t0 = time.perf_counter()
r = ora.mpc_closed_loop(x0, u0, Ad, Bd, Cd, N, T, (0.0, 0.0, 0.0), nthreads=len(os.sched_getaffinity(0)))
tc = time.perf_counter() - t0
ig, cg = info["iters"].cpu().numpy(), info["cmd"].permute(0, 2, 1).cpu().numpy()
same = ig == r["iters"]
agree = np.logical_and.accumulate(same, axis=0)
fin = agree[:, :, None] & ~np.isnan(r["cmd"]) & ~np.isnan(cg)
sg = env.status.cpu().numpy()
xg = env.x_values.cpu().numpy()
good = agree[-1] & np.isfinite(r["x"]).all(1)
print("B %d, N %d, T %d: CPU loop %.1f s on %d threads (%.0f aircraft-steps/s)" % (B, N, T, tc, len(os.sched_getaffinity(0)), B * T / tc))
print("iteration counts equal: %d of %d solves; aircraft equal over all steps: %d of %d" % (same.sum(), same.size, agree[-1].sum(), B))
print("commands, max abs difference where the counts agree: %.3e" % np.abs(cg - r["cmd"])[fin].max())
print("status words (bits 16 | 32 | 64 | 128) equal: %s; states, max rel difference: %.3e"
      % (bool(np.array_equal(sg[agree[-1]] & 240, r["status"][agree[-1]] & 240)), np.max(np.abs(xg[good] - r["x"][good]) / np.maximum(1.0, np.abs(r["x"][good])))))
