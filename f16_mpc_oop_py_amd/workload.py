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


def config2_states(B, seed=20261003):
    """Config 2 of SURVEY.md 8(d): B perturbed in-grid flight conditions, inputs held at each aircraft's initial
    actuator positions.  Returns (x0 [B,18], u0 [B,4])."""
    rng = np.random.default_rng(seed)
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
    return x, np.copy(x[:, 12:16])


def config4_states(B, seed=20261003):
    """Config 4 of SURVEY.md 8(d): the config-2 flight conditions for the batched MPC solve (xcg 0.35, N = 30).
    The leading-edge-flap state is kept 1 degree inside its [0, 25] box: an aircraft sitting ON that bound whose
    linear model predicts crossing it gives an INFEASIBLE QP (the reference's OSQP call would return NaN), which is
    a property of the problem, not a workload one wants to time."""
    x, u = config2_states(B, seed)
    x[:, 16] = np.clip(x[:, 16], 1.0, 24.0)
    return x, u
