import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
rng = np.random.default_rng(7)
B = 4096
h = rng.uniform(5e3, 3e4, B); v = rng.uniform(450.0, 800.0, B)
x, info = F16Batch.trim(h, v)
torch.cuda.synchronize()
nf = info["nfev"].cpu().numpy(); c = info["cost"].cpu().numpy(); it = info["iters"].cpu().numpy()
print("nfev pct 50/90/99/max", np.percentile(nf, [50, 90, 99]), nf.max(), "iters max", it.max())
print("cost pct 50/90/99/max", np.percentile(c, [50, 90, 99]), c.max(), " n(cost>1e-3)", (c > 1e-3).sum())
w = nf.reshape(-1, 64).max(1)
print("per-wave max nfev: mean", w.mean(), "max", w.max(), " sum/mean-based ideal ratio", w.mean() / nf.mean())
bad = c > 1e-3
print("bad h,v sample", np.c_[h[bad][:8], v[bad][:8], c[bad][:8], nf[bad][:8]])
