"""Diagnostic: per-phase s_memtime ticks of k_mpc_big from a -DF16_EXP_STAMPG build (run on the GPU box).
usage: F16HIP_SO=build/libf16hip_stampg.so python tools/gpu_big_stamps.py [B] [N ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Ns = [int(v) for v in sys.argv[2:]] or [41, 100, 150]
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
names = ["w + sync", "A'w (stage 1)", "rhs", "x~ = K^-1 rhs", "projection + dual update", "A x~ (stage 3)", "-", "test + rest"]
for N in Ns:
    env._calc_MPC_action(0, 0, 0, N); torch.cuda.synchronize()
    t0 = time.perf_counter(); u, info = env._calc_MPC_action(0, 0, 0, N, return_info=True); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    it = info["iters"].cpu().numpy(); sa = info["u_seq"][:, :12].cpu().numpy(); s = sa[:, :8]
    b = int(it.argmax())
    print(f"N={N}: {ms:.1f} ms, slowest aircraft {b}: {it[b]:.0f} iterations = {1e3 * ms / it[b]:.1f} us per iteration; ticks per iteration:")
    for k, nm in enumerate(names):
        if nm != "-":
            print(f"   {nm:30s} {s[b, k] / it[b]:9.1f}  ({100 * s[b, k] / s[b].sum():.0f} %)")
    if B > 256:                                  # all CUs busy: the phases of an aircraft of median length (it never runs alone)
        bm = int(np.argsort(it)[len(it) // 2])
        print(f"   aircraft {bm} ({it[bm]:.0f} iterations, all CUs busy): " + " | ".join(f"{nm} {s[bm, k] / it[bm]:.0f}" for k, nm in enumerate(names) if nm != "-"))
    print(f"   set-up (equilibration, Gram product): {sa[b, 10]:.3e} ticks; KKT factorisations: {sa[b, 9]:.0f} x {sa[b, 8] / max(sa[b, 9], 1):.3e} ticks;  batch medians: set-up {np.median(sa[:, 10]):.3e}, factorisations {np.median(sa[:, 9]):.0f} x {np.median(sa[:, 8] / np.maximum(sa[:, 9], 1)):.3e}, iterations {np.median(it):.0f} x {np.median(s[:, :7].sum(1) / it):.0f}")
