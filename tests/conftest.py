import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    """The C restatement (checker).  Built on demand with plain gcc."""
    so = os.path.join(REPO, "oracle", "libf16_oracle.so")
    # always through make (incremental): a stale checker binary would silently compare the kernels with old code
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "libf16_oracle.so"])
    from oracle import mpc_oracle
    return mpc_oracle.COracle(so)


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly (no fallback) when it is missing."""
    from f16_mpc_oop_py_amd import lib
    return lib.load()


# ---- G10: the time histories the reference holds as data files (tools/make_golden.py: g10_time_histories)
G10_TOL = np.array([0.03, 0.03, 0.03, 3e-3, 2e-3, 2e-3, 5e-3, 1e-3, 5e-4, 3e-3, 1e-3, 1e-3])      # npos epos alt [ft] | phi theta psi [deg] |
#                                                         vel [ft/s] | alpha beta [deg] | p q r [deg/s]: 5 x the largest
#                                                         difference seen over 10 s (Simulink's integrator vs explicit Euler at 1 ms)
G10_OUT_TOL = np.array([2e-5, 2e-5, 3e-5, 1e-5, 2e-3, 2e-3])                                      # nx ny nz mach qbar ps


def g10_case(k):
    """(rows [101,23], xcg, x0 [18], trim command [4], doublet): the initial state is the file's first row, the leading-edge flap
    at its steady state (utils.py:332-350: lf1 = -alpha, lf2 = 1.38 alpha - 9.05 qbar / ps + 1.45, degrees)."""
    g = golden("g10_time_histories.npz")
    a = g[f"rows_{k}"]
    d2r = np.pi / 180
    t, npos, epos, alt, phi, th, psi, vel, al, be, p, q, r, nx, ny, nz, mach, qbar, ps, T, el, ail, rud = a[0]
    lf2 = min(max(1.38 * al - 9.05 * qbar / ps + 1.45, 0.0), 25.0)
    x0 = np.array([npos, epos, alt, phi * d2r, th * d2r, psi * d2r, vel, al * d2r, be * d2r, p * d2r, q * d2r, r * d2r, T, el, ail, rud,
                   lf2, -al])
    return a, float(g[f"xcg_{k}"]), x0, np.array([T, el, ail, rud]), float(g[f"doublet_{k}"])


def g10_command(trim_u, doublet, k):
    """command during sample interval k (0.1 s each): runF16Sim.m's doublet +d on [1, 3) s, -d on [3, 5) s, on all three surfaces"""
    d = doublet if 10 <= k < 30 else (-doublet if 30 <= k < 50 else 0.0)
    return trim_u + np.array([0.0, d, d, d])


def g10_rows_of_states(s):
    """[T,18] states -> the file's 12 state columns (angles and rates in degrees)"""
    r2d = 180 / np.pi
    return np.column_stack([s[:, 0:3], s[:, 3:6] * r2d, s[:, 6], s[:, 7:9] * r2d, s[:, 9:12] * r2d])


def g11_case(tag):
    """(A [18,18], B [18,4], x [18], u [4], fi_flag) of fixture G11 (tools/make_golden.py: g11_state_space); tag "hi" or "lo"."""
    g = golden("g11_state_space.npz")
    r = g["trim_row_" + tag]
    d2r = np.pi / 180
    al, qbar, ps = r[8], r[17], r[18]
    if tag == "hi":
        T, el, ail, rud = r[19:23]
        lf2 = min(max(1.38 * al - 9.05 * qbar / ps + 1.45, 0.0), 25.0)
    else:                                       # (the lofi file repeats thrust and elevator in its output columns; no flap)
        T, el, ail, rud = r[13], r[14], 0.0, 0.0
        lf2 = 0.0
    x = np.array([r[1], r[2], r[3], r[4] * d2r, r[5] * d2r, r[6] * d2r, r[7], al * d2r, r[9] * d2r, r[10] * d2r, r[11] * d2r, r[12] * d2r,
                  T, el, ail, rud, lf2, -al])
    return g["A_" + tag], g["B_" + tag], x, np.array([T, el, ail, rud]), 1 if tag == "hi" else 0


def g11_perturbations(x, u, h=1e-5):
    """the 44 points of a central-difference Jacobian: (X [44,18], U [44,4]); rows 2j, 2j+1 = +h, -h on state j, then the inputs"""
    X, U = np.tile(x, (44, 1)), np.tile(u, (44, 1))
    for j in range(18):
        X[2 * j, j] += h
        X[2 * j + 1, j] -= h
    for j in range(4):
        U[36 + 2 * j, j] += h
        U[36 + 2 * j + 1, j] -= h
    return X, U


def g11_jacobians(xdot, h=1e-5):
    """xdot [44,18] at those points -> (A [18,18], B [18,4])"""
    A = np.array([(xdot[2 * j] - xdot[2 * j + 1]) / (2 * h) for j in range(18)]).T
    B = np.array([(xdot[36 + 2 * j] - xdot[36 + 2 * j + 1]) / (2 * h) for j in range(4)]).T
    return A, B
