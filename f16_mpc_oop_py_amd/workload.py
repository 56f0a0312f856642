"""Synthetic workloads named by BASELINE.json `configs` / SURVEY.md 8(d) (pure NumPy; no reference data needed
except the published trim point G5, quoted below)."""
import numpy as np

# trim(10000 ft, 700 ft/s), xcg 0.25 -- values the reference's F16.trim returns
TRIM_XCG25 = np.array([0., 0., 10000., 0., 0.0205901002, 0., 700., 0.0205901002, 0., 0., 0., 0.,
                       2886.64684, -2.03851753, -0.0875768294, -0.0387697724, 0.398604403, -1.17972584])
TRIM_XCG35 = np.array([0., 0., 10000., 0., 0.01682719, 0., 700., 0.01682719, 0., 0., 0., 0.,
                       2626.586, -0.5043574, -0.0875768294, -0.0387697724, 0.0, 0.0])


def _atmos_ratio(h, V):
    tfac = 1 - 0.703e-5 * h
    temp = np.where(h >= 35000.0, 390.0, 519.0 * tfac)
    rho = 2.377e-3 * tfac ** 4.14
    return (0.5 * rho * V * V) / (1715.0 * rho * temp)


def _draw(rng, B):
    """One block of B candidate flight conditions from the seed stream (field by field, in this order)."""
    x = np.zeros((B, 18))
    x[:, 2] = rng.uniform(5e3, 3e4, B)
    x[:, 6] = rng.uniform(400, 850, B)
    alpha_deg = rng.uniform(-5, 20, B)
    x[:, 7] = np.deg2rad(alpha_deg)
    x[:, 8] = np.deg2rad(rng.uniform(-5, 5, B))
    x[:, 3] = rng.uniform(-0.2, 0.2, B)
    x[:, 4] = x[:, 7] + rng.uniform(-0.2, 0.2, B)
    x[:, 5] = rng.uniform(-0.2, 0.2, B)
    x[:, 9:12] = rng.uniform(-0.2, 0.2, (B, 3))
    x[:, 12] = rng.uniform(2000, 8000, B)
    x[:, 13] = rng.uniform(-5, 2, B)
    x[:, 14:16] = rng.uniform(-1, 1, (B, 2))
    x[:, 16] = np.clip(1.38 * alpha_deg - 9.05 * _atmos_ratio(x[:, 2], x[:, 6]) + 1.45, 0, 25)
    x[:, 17] = -alpha_deg
    return x


def _resampled(B, seed, prepare, accept, max_blocks=64):
    """SURVEY.md 8(d)'s rejection-resample rule: candidates come block by block (B at a time) from ONE seed stream; `accept(x, u)`
    -> bool [n] says which candidates stay; the holes are filled, in order, from the next blocks.  With accept=None (or when every
    candidate of the first block is accepted) the result is the first block itself."""
    rng = np.random.default_rng(seed)
    xs, n = [], 0
    for _ in range(max_blocks):
        x = prepare(_draw(rng, B))
        if accept is not None:
            x = x[np.asarray(accept(x, np.copy(x[:, 12:16])), dtype=bool)]
        xs.append(x)
        n += len(x)
        if n >= B:
            break
    else:
        raise RuntimeError("rejection-resample: fewer than B acceptable candidates in %d blocks" % max_blocks)
    x = np.concatenate(xs)[:B]
    return x, np.copy(x[:, 12:16])


def config2_states(B, seed=20261003, accept=None):
    """Config 2 of SURVEY.md 8(d): B perturbed in-grid flight conditions, inputs held at each aircraft's initial actuator
    positions.  Returns (x0 [B,18], u0 [B,4]).  accept: the rejection rule of 8(d) ("resample, same seed stream, any aircraft that
    leaves the grid within 1,000 steps"), e.g. `in_grid_on_gpu(xcg)`; None = the raw stream."""
    return _resampled(B, seed, lambda x: x, accept)


def config4_states(B, seed=20261003, accept=None):
    """Config 4 of SURVEY.md 8(d): the config-2 flight conditions for the batched MPC solve (xcg 0.35, N = 30).
    The leading-edge-flap state is kept 1 degree inside its [0, 25] box: an aircraft sitting ON that bound whose
    linear model predicts crossing it gives an INFEASIBLE QP (the reference's OSQP call would return NaN), which is
    a property of the problem, not a workload one wants to time.  accept: e.g. `qp_feasible_on_gpu(hzn)`."""
    def prepare(x):
        x[:, 16] = np.clip(x[:, 16], 1.0, 24.0)
        return x
    return _resampled(B, seed, prepare, accept)


def in_grid_on_gpu(xcg=0.25, steps=1000, device="cuda:0", **kw):
    """Acceptance rule of config 2 evaluated by the product's own rollout: the aircraft raises no status bit (stays inside the tables
    and the envelope, stays finite) over `steps` open-loop Euler steps.  (8(d) names the CPU restatement for this check; the product
    may not touch the checker, and the two agree on which aircraft leave: tests.)"""
    def accept(x, u):
        from .env import F16Batch
        env = F16Batch(x, u, xcg=xcg, device=device, **kw)
        env.rollout(steps)
        return (env.status == 0).cpu().numpy()
    return accept


def qp_feasible_on_gpu(hzn=30, xcg=0.35, device="cuda:0", **kw):
    """Acceptance rule of config 4: the QP of calc_MPC_action(0, 0, 0, hzn) at the initial state is solved (no status bit: not
    certified infeasible, not at max_iter) with each aircraft's own linearisation."""
    def accept(x, u):
        from .env import F16Batch
        env = F16Batch(x, u, xcg=xcg, device=device, **kw)
        env.build_ssr()
        env._calc_MPC_action(0.0, 0.0, 0.0, hzn)
        return (env.last_status == 0).cpu().numpy()
    return accept
