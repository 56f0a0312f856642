"""Cycles per phase of a (step, aircraft) pair of f16_rollout_mpc, summed over a launch (library built with -DF16_DBG_PAIRSTAMP):
   F16HIP_SO=$PWD/f16_mpc_oop_py_amd/libdbg_STAMP.so python tools/gpu_pair_stamps.py [B] [T]
phases: ticket draw | wait for the previous step | acquire | pair_prepare | solve | pair_finish | release + progress store"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.env import _vp
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35); env.build_ssr(); env.prepare_MPC(30)
env.rollout_MPC(2, 0.0, 0.0, 0.0, 30); env.reset()
cmd = torch.zeros(T * 3 * B + 8, dtype=torch.float64, device="cuda")      # the command record + eight words for the stamps behind it
dem = env._demands(0.0, 0.0, 0.0)
rc = env.lib.f16_rollout_mpc(env._plan, _vp(env._x), _vp(env._u), _vp(dem), None, _vp(cmd), None, _vp(env.status), T, 1, env.xcg, 1, 0, env._stream)
assert rc == 0
torch.cuda.synchronize()
raw = cmd[T * 3 * B:].cpu().numpy().view(np.uint64)[:7].astype(np.float64)
names = ["ticket draw", "wait for step t - 1", "acquire", "pair_prepare", "solve", "pair_finish", "release + progress"]
pairs = B * T
print("B %d, T %d: shader-clock cycles (s_memtime) per pair, mean over %d pairs" % (B, T, pairs))
for n, v in zip(names, raw):
    print("  %-22s %10.0f cycles  (%.2f %%)" % (n, v / pairs, 100 * v / raw.sum()))
