"""Ad-hoc GPU timing of the dynamics kernels (run on the GPU box).  usage: gpu_time_dyn.py [B] [T]"""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config2_states
from f16_mpc_oop_py_amd.env import _vp

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
x0, u0 = config2_states(B)


def timeit(env, traj, n=5):
    def run():
        env._x.copy_(env._x_init)
        rc = env.lib.f16_rollout(env.ctx.handle, _vp(env._x), _vp(env._u), _vp(traj), _vp(env.status), B, B, T,
                                 1, env.dt, env.xcg, env.fi_flag, env.flags, env._stream)
        assert rc == 0
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        env._x.copy_(env._x_init)
        s.record(); 
        rc = env.lib.f16_rollout(env.ctx.handle, _vp(env._x), _vp(env._u), _vp(traj), _vp(env.status), B, B, T,
                                 1, env.dt, env.xcg, env.fi_flag, env.flags, env._stream)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts))


traj = torch.empty((T, 18, B), dtype=torch.float64, device="cuda") if B * T * 144 < 8e9 else None
for fi in (1, 0):
    env = F16Batch(x0, u0, fi_flag=fi)
    ms = timeit(env, traj)
    print(f"B={B} T={T} fi={fi} traj=yes  {ms:.3f} ms  {B*T/ms/1e3:.1f} M steps/s  ({ms/T*1e3:.2f} us/step)")
    if traj is not None:
        w = torch.arange(1, 19, dtype=torch.float64, device="cuda")[None, :, None]
        print("   traj checksum %.17g  last-sample checksum %.17g" % (float((traj * w).sum()), float((traj[-1] * w[0]).sum())))
    ms = timeit(env, None)
    print(f"B={B} T={T} fi={fi} traj=no   {ms:.3f} ms  {B*T/ms/1e3:.1f} M steps/s")
if os.environ.get("PARITY"):
    from oracle import mpc_oracle as mo
    o = mo.COracle()
    env = F16Batch(x0, u0)
    xd = env._calc_xdot().cpu().numpy()
    ref = o.xdot_batch(x0, u0, nthreads=8)
    print("xdot max rel err", np.max(np.abs(xd - ref) / np.maximum(1, np.abs(ref))), "bit-exact frac", (xd == ref).mean())
