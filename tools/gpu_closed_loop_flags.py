"""Config 5 on one GPU (8192 aircraft, N = 30, T = 100, reference settings): per step, how many aircraft's solves raise
F16_ST_QP_INFEASIBLE / QP_MAXITER / NONFINITE in the HOST loop -- with the reference's rule (NaN command -> NaN surface states, np.clip)
and with F16_FLAG_HOLD_COMMAND -- and the sticky totals of the one-launch loop beside them.   python tools/gpu_closed_loop_flags.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f16_mpc_oop_py_amd import F16Batch, dist as fdist
from f16_mpc_oop_py_amd.workload import config4_states
B, N, T = 8192, 30, 100
x0, u0 = config4_states(B)
for hold in (False, True):
    env = F16Batch(x0, u0, xcg=0.35); env.build_ssr(); env.prepare_MPC(N)
    stats = {}
    fdist.closed_loop_mpc_rollout(env, T, N, gather=False, stats=stats, hold_command=hold, fused=False)
    f = stats["flagged_per_step"]
    print("host loop, %s: per step [infeasible, max_iter, not finite] (every 10th step)" % ("hold the previous command" if hold else "reference rule (NaN command)"))
    for k in list(range(0, T, 10)) + [T - 1]:
        print("  step %3d  %s" % (k, f[k].tolist()))
    st = env.status.cpu().numpy()
    print("  sticky after %d steps: infeasible at some step %d, not finite %d, max_iter %d; iters mean %.1f, longest solve per step (mean) %.0f"
          % (T, ((st & 128) != 0).sum(), ((st & 32) != 0).sum(), ((st & 64) != 0).sum(), stats["iters_mean"], stats["iters_max_mean"]))
    envf = F16Batch(x0, u0, xcg=0.35); envf.build_ssr(); envf.prepare_MPC(N)
    envf.rollout_MPC(T, 0.0, 0.0, 0.0, N, hold_command=hold)
    sf = envf.status.cpu().numpy()
    print("  one launch: infeasible at some step %d, not finite %d, max_iter %d; same flagged set as the host loop: %s"
          % (((sf & 128) != 0).sum(), ((sf & 32) != 0).sum(), ((sf & 64) != 0).sum(), bool(np.array_equal(sf & (32 | 64 | 128), st & (32 | 64 | 128)))))
