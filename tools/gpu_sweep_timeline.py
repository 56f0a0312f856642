"""Diagnostic (-DF16_EXP_STAMPG build): start / end of every (horizon, aircraft) solve of the sweep launch on the 100 MHz clock.
usage: F16HIP_SO=build/libf16hip_stampg.so python tools/gpu_sweep_timeline.py [B] [max_hzn]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 150
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
env._calc_MPC_action(0, 0, 0, 33); torch.cuda.synchronize()
t0 = time.perf_counter()
sw, inf = env._calc_constr_checking_hzn(max_hzn=H, return_info=True)
torch.cuda.synchronize()
print(f"sweep {time.perf_counter() - t0:.2f} s")
if len(sys.argv) > 3 and sys.argv[3] == "repeat":          # analyse a repeated sweep (queue ordered by the first one's iteration counts)
    t0 = time.perf_counter()
    sw, inf = env._calc_constr_checking_hzn(max_hzn=H, return_info=True)
    torch.cuda.synchronize()
    print(f"repeated sweep {time.perf_counter() - t0:.2f} s")
it = inf["iters"].cpu().numpy()[32:]; st = inf["r_prim"].cpu().numpy()[32:] / 1e8; en = inf["r_dual"].cpu().numpy()[32:] / 1e8
t00 = st.min(); st -= t00; en -= t00
print(f"launch span {en.max():.2f} s; busy CU-seconds {float((en - st).sum()):.0f} = {float((en - st).sum()) / en.max():.0f} CUs on average")
for q in (0.25, 0.5, 0.75, 1.0, 1.5, 2, 3, 4, 5, 6, 7, 8):
    if q < en.max():
        print(f"  t = {q:5.2f} s: running {int(((st <= q) & (en > q)).sum()):4d}, not started {int((st > q).sum()):5d}")
o = np.dstack(np.unravel_index(np.argsort(-en, axis=None), en.shape))[0][:12]
for k, b in o:
    print(f"  N = {k + 33:3d} aircraft {b:2d}: start {st[k, b]:.2f} end {en[k, b]:.2f} iterations {it[k, b]:.0f} -> {1e6 * (en[k, b] - st[k, b]) / it[k, b]:.1f} us per iteration")
d = (en - st) / np.maximum(it, 1) * 1e6
for N in (33, 41, 57, 80, 100, 125, 150):
    if N <= H:
        print(f"  N = {N}: us per iteration (incl. set-up, factorisations) min {d[N - 33].min():.1f} median {np.median(d[N - 33]):.1f} max {d[N - 33].max():.1f}; start {st[N - 33].min():.2f}..{st[N - 33].max():.2f}")
print("started within 10 ms of the launch:", int((st < 0.01).sum()), "jobs; within 100 ms:", int((st < 0.1).sum()))
for N in range(H, H - 6, -1):
    print(f"  N = {N}: start {np.sort(st[N - 33])[[0, 15, 31, 47, 63]].round(3)}")
xcd = np.arange(B) % 8
for x in range(8):
    sel = xcd == x
    print(f"  aircraft = {x} mod 8: busy seconds {float((en - st)[:, sel].sum()):.0f}, last end {en[:, sel].max():.2f}, iterations {it[:, sel].sum():.3e}")
