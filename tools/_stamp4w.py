import sys; sys.path.insert(0,'.')
import torch, numpy as np
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config2_states
x0,u0=config2_states(4096); env=F16Batch(x0,u0)
T=1000
traj=env.rollout(T, traj_every=1)
torch.cuda.synchronize()
d=traj[2].reshape(-1)[:16].cpu().numpy().reshape(4,4)/T
print("cycles/step per wave: [second-half work, barrier1 wait, first-half work, barrier2 wait]")
for w in range(4): print(w, np.round(d[w]))
