"""Ad-hoc GPU timing of the control-chain kernels (run on the GPU box)."""
import sys, time
sys.path.insert(0, ".")
import torch
from f16_mpc_oop_py_amd import F16Batch
from f16_mpc_oop_py_amd.workload import config4_states

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, u0 = config4_states(B)
env = F16Batch(x0, u0, xcg=0.35)


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("linearise ms", t(lambda: env._linearise_na()))
print("c2d ms", t(lambda: env.discretise()))
print("lqr gain chain ms", t(lambda: env._calc_LQR_gain()))
env.build_ssr()
for N in (10, 30):
    print(N, "mpc setup only ms", t(lambda: env._calc_MPC_action(0, 0, 0, N, settings=dict(max_iter=0))))
    print(N, "mpc 25 it ms", t(lambda: env._calc_MPC_action(0, 0, 0, N, settings=dict(max_iter=25))))
    print(N, "mpc 100 it no-check ms", t(lambda: env._calc_MPC_action(0, 0, 0, N, settings=dict(max_iter=100, check_every=1000))))
    print(N, "mpc full ms", t(lambda: env._calc_MPC_action(0, 0, 0, N)))
# cost of one termination test: 100 iterations that never converge, tested every 5 (20 tests + the final one) against never
tn = t(lambda: env._calc_MPC_action(0, 0, 0, 30, settings=dict(max_iter=100, check_every=1000, eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=0)))
t5 = t(lambda: env._calc_MPC_action(0, 0, 0, 30, settings=dict(max_iter=100, check_every=5, eps_abs=1e-30, eps_rel=1e-30, adaptive_rho=0)))
print("30 mpc termination test: %.1f us per 4096 (= %.1f iterations)" % ((t5 - tn) / 19 * 1e3, (t5 - tn) / 19 / ((tn - 1.3) / 100)))
