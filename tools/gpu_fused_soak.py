"""Soak of f16_rollout_mpc: 8,192 aircraft x T steps (default 300, N = 30), one launch against the host loop stepped with F16_FLAG_ONE_LANE
-- the two trajectories must be identical in every sample, with the reference rule (NaN commands) and with F16_FLAG_HOLD_COMMAND."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f16_mpc_oop_py_amd import F16Batch, dist as fdist
from f16_mpc_oop_py_amd.workload import config4_states
B, N = 8192, 30
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
x0, u0 = config4_states(B)
for hold in (False, True):
    out = []
    for fused in (True, False):
        env = F16Batch(x0, u0, xcg=0.35); env.build_ssr(); env.prepare_MPC(N)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr = fdist.closed_loop_mpc_rollout(env, T, N, traj_every=10, gather=False, fused=fused, hold_command=hold, one_lane=not fused)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out.append((tr, env.status.clone(), env._u.clone(), dt))
    same = all(bool(torch.equal(torch.nan_to_num(a, nan=1e300), torch.nan_to_num(b, nan=1e300))) for a, b in zip(out[0][:3], out[1][:3]))
    st = out[0][1]
    print("hold=%s T=%d: one launch %.2f s (%.0f aircraft-steps/s), host loop %.2f s; identical: %s; infeasible at some step %d, not finite %d, stalled %d"
          % (hold, T, out[0][3], B * T / out[0][3], out[1][3], same, int(((st & 128) != 0).sum()), int(((st & 32) != 0).sum()), int(((st & (1 << 26)) != 0).sum())), flush=True)
    assert same
