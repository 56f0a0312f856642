"""Diagnosis of the capture mismatch of round 2 (one-shot calc_MPC_action captured into a HIP graph returned wrong results on
some replays).  Replays the ONE-SHOT call -- its per-call workspace is a stream-ordered allocation, i.e. the graph holds a
mem-alloc and a mem-free node -- with the refusal switched off (F16_MPC_ALLOW_CAPTURE=1) and no host state baked into the
capture (F16_MPC_DISPATCH_ORDER=0), and compares every replay with the eager result per aircraft.
usage (GPU box): python tools/gpu_capture_probe.py [B]; runs both solvers (F16_MPC_WAVE = 1, 0) in child processes."""
import os, subprocess, sys
sys.path.insert(0, ".")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if os.environ.get("_PROBE_CHILD") is None:
    for wave in ("1", "0"):
        env = dict(os.environ, F16_MPC_WAVE=wave, F16_MPC_ALLOW_CAPTURE="1", F16_MPC_DISPATCH_ORDER="0", _PROBE_CHILD="1")
        r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
        print(f"---- F16_MPC_WAVE={wave} rc={r.returncode}\n{r.stdout[-3000:]}\n{r.stderr[-1500:]}")
    sys.exit(0)
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
x0, u0 = config4_states(B, seed=5)
env = F16Batch(x0, u0, xcg=0.35)
env.build_ssr()
dem = torch.zeros((3, B), dtype=torch.float64, device="cuda")
u_eager = env._calc_MPC_action(dem, None, None, 30).clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    env._calc_MPC_action(dem, None, None, 30)            # warm the pool on the capture stream's neighbour
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    u_cap = env._calc_MPC_action(dem, None, None, 30)
bad_total = 0
for rep in range(10):
    u_cap.zero_()
    g.replay()
    if rep % 2:                                           # eager calls between replays use the same pool
        env._calc_MPC_action(dem, None, None, 30)
    torch.cuda.synchronize()
    d = (u_cap - u_eager).abs().amax(dim=1)
    nb = int((d > 0).sum())
    bad_total += nb
    print(f"replay {rep}: aircraft that differ from the eager call {nb} / {B}, max |du| {float(d.max()):.3e}")
print("RESULT:", "replays differ from the eager call" if bad_total else "every replay bit-identical to the eager call")
