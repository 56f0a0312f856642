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
dump = {}
for rep in range(4):
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
    hw = inf["rho"].cpu().numpy().astype(np.int64)          # (diagnostic build: XCC_ID << 16 | HW_ID)
    slot = (hw >> 16) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 16 + ((hw >> 4) & 3)
    gaps, firsts = [], []
    for sl in np.unique(slot):
        m = slot == sl
        o = np.argsort(st[m]); s_, e_ = st[m][o], en[m][o]
        firsts.append(s_[0]); gaps += list(s_[1:] - e_[:-1])
    gaps = np.array(gaps)
    print(f"   {len(np.unique(slot))} SIMDs used; first start {1e6 * np.median(firsts):.1f} us (median) after the earliest; gap between a solve's end and the next "
          f"solve's start on the same SIMD: median {1e6 * np.median(gaps):.1f} us, mean {1e6 * gaps.mean():.1f} us, 95 % {1e6 * np.percentile(gaps, 95):.1f} us, sum {1e3 * gaps.sum():.0f} slot-ms")
    xcc, se = slot // 4096, (slot % 4096) // 512
    per_part = np.bincount(xcc * 8 + se)
    per_part = per_part[per_part > 0]
    per_simd = np.bincount(np.unique(slot, return_inverse=True)[1])
    print(f"   solves per XCD: {np.bincount(xcc).min()}..{np.bincount(xcc).max()}; per (XCD, shader engine): {per_part.min()}..{per_part.max()} over {len(per_part)} partitions; "
          f"per SIMD: " + ", ".join(f"{k} x {v}" for k, v in enumerate(np.bincount(per_simd)) if v))
    last = np.argsort(-en)[:12]
    rank = np.empty(B, dtype=np.int64); rank[np.argsort(st, kind="stable")] = np.arange(B)
    print("   last finishers (aircraft: iterations, start rank, start ms, duration ms, us per iteration): " +
          "; ".join(f"{b}: {int(it[b])}, {int(rank[b])}, {1e3 * st[b]:.2f}, {1e3 * (en[b] - st[b]):.2f}, {1e6 * (en[b] - st[b]) / it[b]:.2f}" for b in last))
    per = (en - st) / it
    print(f"   us per iteration over all solves: 5 % {1e6 * np.percentile(per, 5):.2f}, median {1e6 * np.median(per):.2f}, 95 % {1e6 * np.percentile(per, 95):.2f}, max {1e6 * per.max():.2f}")
    dump[f"it{rep}"] = it; dump[f"st{rep}"] = st; dump[f"en{rep}"] = en; dump[f"slot{rep}"] = slot
np.savez("gpurun_out/wave_timeline.npz", **dump)
