"""Development check of the wavefront MPC solver (k_mpc_wave) against the 512-lane solver (k_mpc_fast) on the same inputs:
iteration counts, residuals, rho and the whole input sequence.  usage: python tools/gpu_wave_dev.py [B] [N ...]"""
import os, subprocess, sys, json
sys.path.insert(0, ".")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Ns = [int(v) for v in sys.argv[2:]] or [30]
if os.environ.get("_WAVE_CHILD") is None:
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, F16_MPC_WAVE=mode, _WAVE_CHILD="1")
        r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
        print(f"---- F16_MPC_WAVE={mode} rc={r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
    sys.exit(0)
import numpy as np, torch, time
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
mode = os.environ["F16_MPC_WAVE"]
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
for N in Ns:
    u, info = env._calc_MPC_action(0.0, 0.0, 0.0, N, return_info=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        u, info = env._calc_MPC_action(0.0, 0.0, 0.0, N, return_info=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    it = info["iters"].cpu().numpy(); us = info["u_seq"].cpu().numpy(); st = info["status"].cpu().numpy()
    np.savez(f"gpurun_out/wave_dev_{mode}_{N}.npz", it=it, us=us, st=st, rho=info["rho"].cpu().numpy(), rp=info["r_prim"].cpu().numpy(),
             rd=info["r_dual"].cpu().numpy(), u=u.cpu().numpy())
    print(f"N={N} B={B} wave={mode}: {dt*1e3:.3f} ms per batch, iters min/mean/max {it.min():.0f}/{it.mean():.1f}/{it.max():.0f}, status {np.unique(st)}, finite {np.isfinite(us).all()}")
    if mode == "1" and os.path.exists(f"gpurun_out/wave_dev_0_{N}.npz"):
        o = np.load(f"gpurun_out/wave_dev_0_{N}.npz")
        print(f"   vs 512-lane solver: iters equal {np.mean(o['it'] == it):.4f}, max |du_seq| {np.nanmax(np.abs(o['us'] - us)):.3e}, "
              f"max |drho|/rho {np.max(np.abs(o['rho'] - info['rho'].cpu().numpy()) / o['rho']):.3e}, status equal {np.array_equal(o['st'], st)}")
        bad = np.nonzero(o['it'] != it)[0][:10]
        print("   first differing aircraft:", bad, o['it'][bad], it[bad])
