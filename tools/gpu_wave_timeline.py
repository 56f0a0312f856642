"""Diagnostic (-DF16_EXP_STAMPW build): start / end of every solve of the wavefront solver's launch on the 100 MHz clock.
usage: F16HIP_SO=build/libf16hip_stampw.so python tools/gpu_wave_timeline.py [B]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
for rep in range(3):
    u, inf = env._calc_MPC_action(0, 0, 0, 30, return_info=True)
    torch.cuda.synchronize()
    it = inf["iters"].cpu().numpy(); st = inf["r_prim"].cpu().numpy() / 1e8; en = inf["r_dual"].cpu().numpy() / 1e8
    t0 = st.min(); st -= t0; en -= t0
    span = en.max(); busy = float((en - st).sum())
    print(f"call {rep} ({'caller order' if rep == 0 else 'longest first'}): span {1e3 * span:.2f} ms, busy {1e3 * busy:.0f} slot-ms = {busy / span:.0f} of 1024 slots on average; "
          f"per iteration incl. everything {1e6 * np.median((en - st) / it):.2f} us (median)")
    for q in (0.25, 0.5, 0.75, 0.9, 0.95):
        t = q * span
        print(f"   t = {1e3 * t:5.2f} ms: running {int(((st <= t) & (en > t)).sum()):5d}, not started {int((st > t).sum()):5d}")
