"""Diagnostic: per-phase cycles of the MPC build kernel k_mpc<true> from a -DF16_EXP_STAMPB build (run on the GPU box).
usage: F16HIP_SO=build/libf16hip_stampb.so python tools/gpu_build_stamps.py [B]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
for _ in range(3):      # the first call pays first-touch / code-load costs
    u, info = env._calc_MPC_action(0, 0, 0, 30, return_info=True)
    torch.cuda.synchronize()
s = info["u_seq"][:, :7].cpu().numpy()      # [B, 7] shader-clock cycles
names = ["load+Q", "DARE", "G_k", "pred", "q", "P,A'A", "bounds+ext"]
for q, lab in ((None, "mean"), (50, "median"), (5, "p5"), (95, "p95")):
    v = s.mean(0) if q is None else np.percentile(s, q, axis=0)
    print("%-6s cycles: " % lab + " | ".join("%s %d" % (n, x) for n, x in zip(names, v)) + " | total %d" % v.sum())
print("aircraft 0:", s[0].astype(int), " aircraft B-1:", s[-1].astype(int))
