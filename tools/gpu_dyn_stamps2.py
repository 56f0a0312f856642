"""Diagnostic: cycles per phase of the longitudinal table role of the quad rollout (-DF16_EXP_STAMPQ2 build, workgroup 0).
usage: F16HIP_SO=build/libf16hip_stampq2.so python tools/gpu_dyn_stamps2.py"""
import ctypes, sys
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config2_states
x0, u0 = config2_states(4096)
env = F16Batch(x0, u0)
T = 1000
env.rollout(T)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
env.lib.f16_debug_qstamps.argtypes = [ctypes.c_void_p]
assert env.lib.f16_debug_qstamps(out) == 0
v = np.array(list(out), dtype=float) / T
print("longitudinal role, cycles per step: inputs %.0f | breakpoint reads + wait %.0f | cells + broadcasts + corner reads + wait %.0f | arithmetic %.0f | sum %.0f"
      % (v[0], v[1], v[2], v[3], v[:4].sum()))
