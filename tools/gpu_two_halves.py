"""Experiment: does running the one-shot MPC call as two half batches on two streams (build of half B overlapping the
ADMM tail of half A) beat one full-batch call?  Two contexts, so the halves have separate workspaces."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states
B = 4096
x0, u0 = config4_states(B)
full = F16Batch(x0, u0, xcg=0.35); full.build_ssr()
h = B // 2
ea = F16Batch(x0[:h], u0[:h], xcg=0.35); ea.build_ssr()
eb = F16Batch(x0[h:], u0[h:], xcg=0.35); eb.build_ssr()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    full._calc_MPC_action(0.0, 0.0, 0.0, 30)
def two():
    with torch.cuda.stream(sa):
        ea._calc_MPC_action(0.0, 0.0, 0.0, 30)
    with torch.cuda.stream(sb):
        eb._calc_MPC_action(0.0, 0.0, 0.0, 30)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("one call of 4096: %.3f ms" % t(one))
print("two halves on two streams: %.3f ms" % t(two))
print("one call of 4096: %.3f ms" % t(one))
print("two halves on two streams: %.3f ms" % t(two))
