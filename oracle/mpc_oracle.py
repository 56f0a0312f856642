"""oracle/mpc_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

numpy/scipy restatement of the reference's control chain (utils.py:21-285,
env.py:344-424): ZOH discretisation, dlqr, calc_MC, dmom, setup_OSQP, plus the two
things the reference delegates to the absent third-party `osqp` package
(PyPI `osqp`, version unpinned by the reference -- README.md:11):

  * `admm_osqp_style`  -- the published OSQP ADMM iteration (Stellato et al. 2020,
    Algorithm 1) in the reduced dense form, with the fixed deterministic settings of
    SURVEY.md 8(d) config 4.  This is what the HIP QP kernel is compared with
    iterate-for-iterate.
  * `qp_exact`         -- the unique minimiser of the strictly convex QP by an
    active-set KKT solve seeded from a tight ADMM run and verified through the
    KKT conditions.  MPC outputs are judged against this ("parity unpinned" at the
    OSQP boundary: no reference test asserts anything about OSQP's output).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
scipy's cont2discrete / solve_discrete_are / solve_discrete_lyapunov ARE the
reference's arithmetic for those steps (env.py:46,50,351; utils.py:100,242) and
are called directly.
"""
import ctypes
import os

import numpy as np
import scipy.linalg
from scipy.signal import cont2discrete

HERE = os.path.dirname(os.path.abspath(__file__))

# parameters.py:134-137,158-183,198-210 (index maps + MPC bound vectors)
OBS_X_IDX = [2, 3, 4, 7, 8, 9, 10, 11, 16, 17]
MPC_X_IDX = [3, 4, 7, 8, 9, 10, 11, 17, 16]
MPC_U_IN_X_IDX = [13, 14, 15]
MPC_U_IDX = [1, 2, 3]
INF = np.inf
MPC_X_LB = np.array([-INF, -INF, -20., -30., -300., -100., -50., -INF, 0.])
MPC_X_UB = np.array([INF, INF, 90., 30., 300., 100., 50., INF, 25.])
MPC_U_LB = np.array([-25., -21.5, -30.])
MPC_U_UB = np.array([25., 21.5, 30.])
MPC_UDOT_LB = np.array([-60., -80., -120.])
MPC_UDOT_UB = np.array([60., 80., 120.])


# --------------------------------------------------------------- C oracle
class CQPSettings(ctypes.Structure):
    _fields_ = [("rho", ctypes.c_double), ("sigma", ctypes.c_double), ("alpha", ctypes.c_double),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double), ("eps_prim_inf", ctypes.c_double),
                ("max_iter", ctypes.c_int), ("check_every", ctypes.c_int), ("rho_every", ctypes.c_int),
                ("adaptive_rho", ctypes.c_int), ("scaling", ctypes.c_int)]


class COracle:
    """ctypes view of oracle/libf16_oracle.so (build: `make -C oracle`)."""

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "libf16_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run `make -C oracle` (or __graft_entry__.build())")
        L = self.lib = ctypes.CDLL(path)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        d, i, l = ctypes.c_double, ctypes.c_int, ctypes.c_long
        L.f16o_init.restype = None
        L.f16o_table.restype = d
        L.f16o_table.argtypes = [i, d, d, d]
        L.f16o_lofi.argtypes = [i, d, d, d, dp]
        L.f16o_nlplant.argtypes = [dp, dp, i, d]
        L.f16o_calc_xdot.argtypes = [dp, dp, dp, i, d]
        L.f16o_calc_xdot_na.argtypes = [dp, dp, dp, dp, i, d]
        L.f16o_step.argtypes = [dp, dp, d, i, d]
        L.f16o_step.restype = i
        L.f16o_linearise_na.argtypes = [dp, dp, dp, d, dp, dp, dp, dp, i, d]
        L.f16o_linearise_full.argtypes = [dp, dp, d, dp, dp, dp, dp, i, d]
        L.f16o_xdot_batch.argtypes = [dp, dp, dp, l, i, d, i]
        L.f16o_rollout.argtypes = [dp, dp, l, i, d, i, d, dp, ip, i]
        L.f16o_rollout_lqr.argtypes = [dp, dp, dp, dp, l, i, d, i, d, dp, dp, ip, i]
        L.f16o_set_xcg.argtypes = [d]
        L.f16o_last_status.restype = i
        L.atmos.argtypes = [d, d, dp]
        L.f16o_set_fix_clr.argtypes = [i]
        # control chain in C (oracle/f16_mpc_oracle.c)
        sp = ctypes.POINTER(CQPSettings)
        L.f16o_c2d.argtypes = [dp, dp, i, i, d, dp, dp]
        L.f16o_dare.argtypes = [dp, dp, dp, i, i, dp]
        L.f16o_mpc_qp.argtypes = [dp, dp, dp, dp, i, d, dp, dp, dp, dp, dp, dp]
        L.f16o_admm.argtypes = [i, i, dp, dp, dp, dp, dp, sp, i, dp, dp]
        L.f16o_qp_default_settings.argtypes = [sp, i]
        L.f16o_mpc_batch.argtypes = [dp, l, i, d, d, i, dp, sp, i, dp, ip, ip, i]
        L.f16o_mpc_closed_loop.argtypes = [dp, dp, dp, dp, dp, l, i, i, d, d, i, dp, sp, i, i, dp, ip, ip, dp, i]
        L.f16o_init()

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

    def table(self, tid, alpha, beta=0.0, el=0.0):
        return self.lib.f16o_table(int(tid), float(alpha), float(beta), float(el))

    def lofi(self, which, alpha, beta, el):
        out = np.zeros(9)
        self.lib.f16o_lofi(which, alpha, beta, el, self._p(out))
        return out

    def atmos(self, alt, vt):
        out = np.zeros(3)
        self.lib.atmos(alt, vt, self._p(out))
        return out

    def nlplant(self, xu, fi_flag=1, xcg=0.25):
        xu = np.ascontiguousarray(xu, dtype=np.float64)
        xdot = np.zeros(18)
        self.lib.f16o_nlplant(self._p(xu), self._p(xdot), fi_flag, xcg)
        return xdot

    def calc_xdot(self, x, u, fi_flag=1, xcg=0.25):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        xdot = np.zeros(18)
        self.lib.f16o_calc_xdot(self._p(x), self._p(u), self._p(xdot), fi_flag, xcg)
        return xdot

    def calc_xdot_na(self, x_full, x9, u3, fi_flag=1, xcg=0.25):
        x_full = np.ascontiguousarray(x_full, dtype=np.float64)
        x9 = np.ascontiguousarray(x9, dtype=np.float64)
        u3 = np.ascontiguousarray(u3, dtype=np.float64)
        out = np.zeros(9)
        self.lib.f16o_calc_xdot_na(self._p(x_full), self._p(x9), self._p(u3), self._p(out), fi_flag, xcg)
        return out

    def xdot_batch(self, x, u, fi_flag=1, xcg=0.25, nthreads=1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.zeros_like(x)
        self.lib.f16o_xdot_batch(self._p(x), self._p(u), self._p(out), x.shape[0], fi_flag, xcg, nthreads)
        return out

    def rollout(self, x0, u, T, dt=0.001, fi_flag=1, xcg=0.25, store=True, nthreads=1):
        """x0 [B,18], u [B,4] -> (x_final [B,18], traj [T,B,18] or None, status [B])."""
        x = np.array(x0, dtype=np.float64, order="C")
        u = np.ascontiguousarray(u, dtype=np.float64)
        B = x.shape[0]
        traj = np.zeros((T, B, 18)) if store else None
        status = np.zeros(B, dtype=np.int32)
        self.lib.f16o_rollout(self._p(x), self._p(u), B, T, dt, fi_flag, xcg,
                              self._p(traj) if store else None,
                              status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), nthreads)
        return x, traj, status

    def rollout_lqr(self, x0, u0, K, dem, T, dt=0.001, fi_flag=1, xcg=0.25, store=True, nthreads=1):
        """test_env_mk2.py:70-85 for B aircraft: x0 [B,18], u0 [B,4], K [B,3,9] (= -dlqr), dem [B,3]
        -> (x_final, traj [T,B,18] or None, u_last [B,4], status)."""
        x = np.array(x0, dtype=np.float64, order="C")
        B = x.shape[0]
        u0 = np.ascontiguousarray(np.broadcast_to(u0, (B, 4)), dtype=np.float64)
        K = np.ascontiguousarray(np.broadcast_to(K, (B, 3, 9)), dtype=np.float64)
        dem = np.ascontiguousarray(np.broadcast_to(dem, (B, 3)), dtype=np.float64)
        traj = np.zeros((T, B, 18)) if store else None
        u_out = np.zeros((B, 4))
        status = np.zeros(B, dtype=np.int32)
        self.lib.f16o_rollout_lqr(self._p(x), self._p(u0), self._p(K), self._p(dem), B, T, dt, fi_flag, xcg,
                                  self._p(traj) if store else None, self._p(u_out),
                                  status.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), nthreads)
        return x, traj, u_out, status

    def linearise_na(self, x_full, x9=None, u3=None, eps=1e-5, fi_flag=1, xcg=0.25):
        x_full = np.ascontiguousarray(x_full, dtype=np.float64)
        x9 = np.ascontiguousarray(x_full[MPC_X_IDX] if x9 is None else x9, dtype=np.float64)
        u3 = np.ascontiguousarray(x_full[MPC_U_IN_X_IDX] if u3 is None else u3, dtype=np.float64)
        A, B, C, D = np.zeros((9, 9)), np.zeros((9, 3)), np.zeros((9, 9)), np.zeros((9, 3))
        self.lib.f16o_linearise_na(self._p(x_full), self._p(x9), self._p(u3), eps,
                                   self._p(A), self._p(B), self._p(C), self._p(D), fi_flag, xcg)
        return A, B, C, D

    # ---- control chain in C (second checker + CPU baseline); mode 0 = admm_osqp_style, 1 = admm_osqp, 2 = admm_osqp with
    # drop_unbounded_rows
    def qp_settings(self, mode, **kw):
        s = CQPSettings()
        self.lib.f16o_qp_default_settings(ctypes.byref(s), mode)
        for k, v in kw.items():
            setattr(s, k, v)
        return s

    def c2d(self, A, B, dt):
        A, B = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(B, dtype=np.float64)
        ns, ni = B.shape
        Ad, Bd = np.zeros((ns, ns)), np.zeros((ns, ni))
        self.lib.f16o_c2d(self._p(A), self._p(B), ns, ni, dt, self._p(Ad), self._p(Bd))
        return Ad, Bd

    def dare(self, A, B, Q):
        A, B, Q = (np.ascontiguousarray(a, dtype=np.float64) for a in (A, B, Q))
        X = np.zeros_like(A)
        rc = self.lib.f16o_dare(self._p(A), self._p(B), self._p(Q), A.shape[0], B.shape[1], self._p(X))
        assert rc == 0, rc
        return X

    def mpc_qp(self, x_full, Ad, Bd, Cd, hzn, dt, dem=(0.0, 0.0, 0.0)):
        x_full, Ad, Bd, Cd = (np.ascontiguousarray(a, dtype=np.float64) for a in (x_full, Ad, Bd, Cd))
        dem = np.ascontiguousarray(dem, dtype=np.float64)
        n, m = 3 * hzn, 15 * hzn
        P, q, A, l, u = np.zeros((n, n)), np.zeros(n), np.zeros((m, n)), np.zeros(m), np.zeros(m)
        self.lib.f16o_mpc_qp(self._p(x_full), self._p(Ad), self._p(Bd), self._p(Cd), hzn, dt, self._p(dem), self._p(P), self._p(q),
                             self._p(A), self._p(l), self._p(u))
        return P, q, A, l, u

    def admm(self, P, q, A, l, u, mode=1, **kw):
        P, q, A, l, u = (np.ascontiguousarray(a, dtype=np.float64) for a in (P, q, A, l, u))
        s = self.qp_settings(mode, **kw)
        x, info = np.zeros(P.shape[0]), np.zeros(4)
        st = self.lib.f16o_admm(P.shape[0], A.shape[0], self._p(P), self._p(q), self._p(A), self._p(l), self._p(u), ctypes.byref(s),
                                mode, self._p(x), self._p(info))
        return dict(x=x, iters=int(info[0]), r_prim=info[1], r_dual=info[2], rho=info[3], status=st, infeasible=st == 2)

    def mpc_batch(self, x, hzn, dt=0.001, xcg=0.35, fi_flag=1, dem=None, mode=2, nthreads=1, **kw):
        """calc_MPC_action for every row of x [B,18] on the CPU: linearise + ZOH + setup_OSQP + solve."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = x.shape[0]
        s = self.qp_settings(mode, **kw)
        u, it, st = np.zeros((B, 3)), np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        ipt = ctypes.POINTER(ctypes.c_int)
        dm = np.ascontiguousarray(dem, dtype=np.float64) if dem is not None else None
        self.lib.f16o_mpc_batch(self._p(x), B, hzn, dt, xcg, fi_flag, self._p(dm) if dm is not None else None, ctypes.byref(s), mode,
                                self._p(u), it.ctypes.data_as(ipt), st.ctypes.data_as(ipt), nthreads)
        return dict(u=u, iters=it, status=st)

    def mpc_closed_loop(self, x0, u0, Ad, Bd, Cd, hzn, T, dem, dt=0.001, xcg=0.35, fi_flag=1, mode=2, hold=False, store=False,
                        nthreads=1, **kw):
        """test_env.py:480-495 for B aircraft on the CPU with the reduced model frozen per aircraft (Ad [B,9,9], Bd [B,9,3],
        Cd [B,9,9]): T steps of calc_MPC_action(N = hzn) + step.  dem [B,3] or [3].  The product's rules for frozen / non-finite /
        infeasible aircraft (oracle/f16_mpc_oracle.c: f16o_mpc_closed_loop).
        -> dict(x [B,18], u [B,4], cmd [T,B,3], iters [T,B], status [B], traj [T,B,18] or None)."""
        x = np.array(x0, dtype=np.float64, order="C")
        B = x.shape[0]
        u = np.array(np.broadcast_to(u0, (B, 4)), dtype=np.float64, order="C")
        c = lambda a, shp: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), shp))
        Ad, Bd, Cd, dm = c(Ad, (B, 9, 9)), c(Bd, (B, 9, 3)), c(Cd, (B, 9, 9)), c(dem, (B, 3))
        s = self.qp_settings(mode, **kw)
        cmds, its, st = np.zeros((T, B, 3)), np.zeros((T, B), dtype=np.int32), np.zeros(B, dtype=np.int32)
        traj = np.zeros((T, B, 18)) if store else None
        ipt = ctypes.POINTER(ctypes.c_int)
        self.lib.f16o_mpc_closed_loop(self._p(x), self._p(u), self._p(Ad), self._p(Bd), self._p(Cd), B, hzn, T, dt, xcg, fi_flag,
                                      self._p(dm), ctypes.byref(s), mode, 1 if hold else 0, self._p(cmds), its.ctypes.data_as(ipt),
                                      st.ctypes.data_as(ipt), self._p(traj) if store else None, nthreads)
        return dict(x=x, u=u, cmd=cmds, iters=its, status=st, traj=traj)

    def linearise_full(self, x, u, eps=1e-5, fi_flag=1, xcg=0.25):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        A, B, C, D = np.zeros((18, 18)), np.zeros((18, 4)), np.zeros((10, 18)), np.zeros((10, 4))
        self.lib.f16o_linearise_full(self._p(x), self._p(u), eps, self._p(A), self._p(B), self._p(C), self._p(D),
                                     fi_flag, xcg)
        return A, B, C, D


def trim(oracle, h_t, v_t, fi_flag=1, xcg=0.25):
    """env.py:198-292 restated: scipy's Nelder-Mead (the reference's own optimiser call, env.py:273) on the C
    restatement of _calc_xdot.  Returns (x_trim[18], scipy result)."""
    from scipy.optimize import minimize
    pi = np.pi

    def obj(UX0):
        P3, dh, da, dr, alpha = UX0
        rho0 = 2.377e-3
        tfac = 1 - 0.703e-5 * h_t
        temp = 519 * tfac
        if h_t >= 35000:
            temp = 390
        rho = rho0 * tfac ** 4.14
        qbar = 0.5 * rho * v_t ** 2
        ps = 1715 * rho * temp
        dlef = 1.38 * alpha * 180 / pi - 9.05 * qbar / ps + 1.45
        x = np.array([0, 0, h_t, 0, alpha, 0, v_t, alpha, 0, 0, 0, 0, P3, dh, da, dr, dlef, -alpha * 180 / pi])
        x[12] = np.clip(x[12], 1000, 19000)
        x[13] = np.clip(x[13], -25, 25)
        x[14] = np.clip(x[14], -21.5, 21.5)
        x[15] = np.clip(x[15], -30., 30)
        x[7] = np.clip(x[7], -20. * pi / 180, 90 * pi / 180)
        xd = oracle.calc_xdot(x, x[12:16], fi_flag, xcg)
        w = np.array([0, 0, 5, 10, 10, 10, 2, 10, 10, 10, 10, 10.])
        return np.matmul(w, xd[0:12] ** 2)

    opt = minimize(obj, [5000, -0.09, 8.49, -0.01, 0.01], method="Nelder-Mead", tol=1e-10, options={"maxiter": 5e+04})
    P3, dh, da, dr, al = opt.x
    tfac = 1 - 0.703e-5 * h_t
    temp = 390 if h_t >= 35000 else 519 * tfac
    rho = 2.377e-3 * tfac ** 4.14
    dlef = 1.38 * al * 180 / pi - 9.05 * (0.5 * rho * v_t ** 2) / (1715 * rho * temp) + 1.45
    return np.array([0, 0, h_t, 0, al, 0, v_t, al, 0, 0, 0, 0, P3, dh, da, dr, dlef, -al * 180 / pi]), opt


# ------------------------------------------------------- control chain
def c2d(A, B, C, D, dt):
    """env.py:46,50,351 -- scipy.signal.cont2discrete, default method zoh."""
    return cont2discrete((A, B, C, D), dt)[0:4]


def dlqr(A, B, Q, R):
    """utils.py:219-245."""
    P = np.array(scipy.linalg.solve_discrete_are(A, B, Q, R))
    return np.array(scipy.linalg.inv(B.T @ P @ B + R) @ (B.T @ P @ A))


def calc_MC(A, B, dt, hzn):
    """utils.py:171-197: MM[i] = A^(i+1), CC[i,j] = A^(i-j) B (i>=j)."""
    ns, ni = A.shape[0], B.shape[1]
    CC = np.zeros((ns * hzn, ni * hzn))
    MM = np.zeros((ns * hzn, ns))
    for i in range(hzn):
        MM[ns * i:ns * (i + 1), :] = np.linalg.matrix_power(A, i + 1)
        for j in range(i + 1):
            CC[ns * i:ns * (i + 1), ni * j:ni * (j + 1)] = np.linalg.matrix_power(A, i - j) @ B
    return MM, CC


def dmom(mat, num):
    """utils.py:270-285: block-diagonal replication."""
    r, c = mat.shape
    out = np.zeros((r * num, c * num))
    for i in range(num):
        out[r * i:r * (i + 1), c * i:c * (i + 1)] = mat
    return out


def lqr_gain_from_linearisation(Ac, Bc, Cc, Dc, dt):
    """env.py:344-358 after the linearise call: K = -dlqr(Ad, Bd, Cd'Cd, I)."""
    A, B, C, D = c2d(Ac, Bc, Cc, Dc, dt)
    return -dlqr(A, B, C.T @ C, np.eye(B.shape[1]))


def lqr_action(p_dem, q_dem, r_dem, K, x, u0):
    """env.py:360-371."""
    x_ref = np.copy(x)
    x_ref[4], x_ref[5], x_ref[6] = p_dem, q_dem, r_dem
    return -K @ (x_ref - x) + u0


def rollout_lqr_linear(x0, Ad, Bd, K, x_ref, u0, T, track=None, every=1):
    """The reference's LINEAR-model closed loops for one aircraft (9 states, 3 inputs):
      test_env_mk2.py:54-62 (what main.py:35 runs):  u = _calc_LQR_action(p, q, r, K, x, u0) = -K (x_ref - x) + u0 with
          x_ref = x except x_ref[4:7] = demands (env.py:360-371), K = -dlqr;  x = ssr.Ad @ x + ssr.Bd @ u      -> track = (4, 5, 6)
      test_env.py:553-559:  u = -K' (x - x_ref), K' = dlqr, a fixed reference;  x = A @ x + B @ u             -> track = None (all), K = -K'
    x_ref: the tracked entries are read from it (the others follow the current state).  Returns (x [T/every, 9], u [T/every, 3])."""
    x = np.array(x0, dtype=float)
    xr_given = np.asarray(x_ref, dtype=float)
    idx = list(range(len(x))) if track is None else list(track)
    xs, us = [], []
    for t in range(T):
        xr = np.copy(x)
        xr[idx] = xr_given[idx]
        u = -K @ (xr - x) + u0
        x = Ad @ x + Bd @ u
        if (t + 1) % every == 0:
            xs.append(np.copy(x)), us.append(np.copy(u))
    return np.array(xs), np.array(us)


def setup_OSQP(x_ref, A, B, Q, R, hzn, dt, x, act_states,
               x_lb=MPC_X_LB, x_ub=MPC_X_UB, u_lb=MPC_U_LB, u_ub=MPC_U_UB,
               udot_lb=MPC_UDOT_LB, udot_ub=MPC_UDOT_UB):
    """utils.py:21-167.  Returns P, q, A, l, u with 1-D q/l/u."""
    m, n = len(x), len(act_states)
    xr = np.tile(x_ref, hzn)
    MM, CC = calc_MC(A, B, dt, hzn)
    K = -dlqr(A, B, Q, R)
    Q_bar = scipy.linalg.solve_discrete_lyapunov((A + B @ K).T, Q + K.T @ R @ K)
    QQ = dmom(Q, hzn)
    QQ[-m:, -m:] = Q_bar
    RR = dmom(R, hzn)
    P = 2 * (CC.T @ QQ @ CC + RR)
    q = -2 * ((xr - MM @ x) @ QQ @ CC)
    sl = np.tile(x_lb, hzn) - MM @ x
    su = np.tile(x_ub, hzn) - MM @ x
    cl, cu = np.tile(u_lb, hzn), np.tile(u_ub, hzn)
    rl = np.concatenate((act_states + udot_lb * dt, np.tile(udot_lb, hzn - 1)))
    ru = np.concatenate((act_states + udot_ub * dt, np.tile(udot_ub, hzn - 1)))
    Dm = np.eye(n * hzn)
    for i in range(n, n * hzn):
        Dm[i, i - n] = -1
    Ac = np.concatenate((CC, np.eye(n * hzn), Dm), axis=0)
    return P, q, Ac, np.concatenate((sl, cl, rl)), np.concatenate((su, cu, ru))


def mpc_qp(x_full, Ad, Bd, Cd, hzn, dt, p_dem=0.0, q_dem=0.0, r_dem=0.0):
    """env.py:373-416: QP data for one aircraft state (demands land in x_ref[5:8] -- quirk 8-Q.4)."""
    x = np.asarray(x_full)[MPC_X_IDX]
    act = np.asarray(x_full)[MPC_U_IN_X_IDX]
    x_ref = np.copy(x)
    x_ref[5:8] = [p_dem, q_dem, r_dem]
    Q = Cd.T @ Cd
    R = np.eye(3)
    return setup_OSQP(x_ref, Ad, Bd, Q, R, hzn, dt, x, act)


# ------------------------------------------------------------- QP solvers
RHO_AUTO_SCALE = 2.0      # start value of rho when rho <= 0: RHO_AUTO_SCALE * sqrt(tr P / tr A'A) (same constant as csrc/f16_mpc.hpp)
ADMM_DEFAULTS = dict(rho=0.0, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, eps_prim_inf=1e-4,
                     check_every=25, rho_every=100, max_iter=40000, adaptive_rho=True)      # the builder's opt-in rule


def admm_osqp_style(P, q, A, l, u, **kw):
    """The builder's opt-in settings of the same solver (see admm_osqp below): no equilibration, rows with l = -inf and
    u = +inf dropped, start value rho = RHO_AUTO_SCALE * sqrt(tr P / tr A'A) unless a positive rho is given -- every other
    rule (iteration, termination test, rho update, infeasibility certificate) is admm_osqp's."""
    o = dict(ADMM_DEFAULTS)
    o.update(kw)
    return admm_osqp(P, q, A, l, u, drop_unbounded_rows=True, scaling=0, **o)


# ---- OSQP as the reference invokes it (env.py:420-422: osqp.OSQP().setup(P, q, A, l, u, max_iter=40000, verbose=True,
# polish=False) -> every other setting at its default).  The package is absent (PyPI `osqp`, unpinned: README.md:11), so
# this restates its PUBLISHED algorithm (Stellato et al. 2020 + the 0.6.x sources scaling.c / auxil.c / osqp.c as
# summarised in SURVEY.md Appendix C): Ruiz equilibration (10 passes, D / E / c), rho = 0.1 with the per-row rho vector
# (rho_min on rows without bounds, 1e3 rho on equality rows), sigma 1e-6, alpha 1.6, termination on UNSCALED residuals
# every 25 iterations, adaptive rho from the SCALED residuals.  OSQP's default update interval is wall-clock based; the
# deterministic stand-in is its own no-timer constant ADAPTIVE_RHO_FIXED = 100 iterations (`rho_every`).
OSQP_INFTY, MIN_SCALING, MAX_SCALING = 1e30, 1e-4, 1e4
RHO_MIN, RHO_MAX, RHO_TOL, RHO_EQ_OVER_RHO_INEQ = 1e-6, 1e6, 1e-4, 1e3
OSQP_DEFAULTS = dict(rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, eps_prim_inf=1e-4, scaling=10,
                     check_every=25, rho_every=100, adaptive_rho=True, adaptive_rho_tolerance=5.0, max_iter=40000)


def _limit_scaling(v):
    v = np.where(v < MIN_SCALING, 1.0, v)
    return np.where(v > MAX_SCALING, MAX_SCALING, v)


def osqp_scale(P, q, A, l, u, passes=10):
    """scaling.c:scale_data -- returns (Ps, qs, As, ls, us, D, E, c) with Ps = c D P D, qs = c D q, As = E A D,
    ls = E l, us = E u."""
    n, m = P.shape[0], A.shape[0]
    P, q, A = np.array(P, dtype=float), np.array(q, dtype=float), np.array(A, dtype=float)
    D, E, c = np.ones(n), np.ones(m), 1.0
    for _ in range(passes):
        Dt = np.maximum(np.abs(P).max(axis=0), np.abs(A).max(axis=0) if m else 0.0)     # compute_inf_norm_cols_KKT
        Et = np.abs(A).max(axis=1) if m else np.ones(0)
        Dt, Et = 1.0 / np.sqrt(_limit_scaling(Dt)), 1.0 / np.sqrt(_limit_scaling(Et))
        P = Dt[:, None] * P * Dt[None, :]
        A = Et[:, None] * A * Dt[None, :]
        q = Dt * q
        D, E = D * Dt, E * Et
        ct = float(_limit_scaling(np.abs(P).max(axis=0).mean()))                         # cost scaling
        qn = float(_limit_scaling(np.abs(q).max()))
        ct = 1.0 / max(ct, qn)
        P, q, c = P * ct, q * ct, c * ct
    return P, q, A, E * l, E * u, D, E, c


def admm_osqp(P, q, A, l, u, drop_unbounded_rows=False, **kw):
    """The solve of env.py:420-424 by the published OSQP algorithm (see the block comment above).
    drop_unbounded_rows: rows with l = -inf and u = +inf take part in the equilibration (they are rows of A) but are then
    left out of the iteration instead of being carried with rho_min = 1e-6 (their y stays 0 and A x - z stays 0 on them;
    what is dropped is a 1e-6-weighted term of the KKT matrix) -- this is what the HIP kernels do.
    Returns dict(x, y, iters, r_prim, r_dual (unscaled), rho, infeasible, D, E, c)."""
    o = dict(OSQP_DEFAULTS)
    o.update(kw)
    l = np.maximum(np.asarray(l, dtype=float), -OSQP_INFTY)       # the Python wrapper clips +-inf to +-1e30
    u = np.minimum(np.asarray(u, dtype=float), OSQP_INFTY)
    n = P.shape[0]
    if o["scaling"]:
        Ps, qs, As, ls, us, D, E, c = osqp_scale(P, q, A, l, u, o["scaling"])
    else:
        Ps, qs, As, ls, us = np.array(P, float), np.array(q, float), np.array(A, float), l.copy(), u.copy()
        D, E, c = np.ones(n), np.ones(A.shape[0]), 1.0
    loose = (ls < -OSQP_INFTY * MIN_SCALING) & (us > OSQP_INFTY * MIN_SCALING)
    keep = ~loose if drop_unbounded_rows else np.ones(len(ls), bool)
    As, ls, us, Ek, loose_k = As[keep], ls[keep], us[keep], E[keep], loose[keep]
    m = As.shape[0]
    eq = (us - ls) < RHO_TOL
    rho, sigma, alpha = float(o["rho"]), o["sigma"], o["alpha"]
    if not rho > 0:     # the builder's automatic start value (only meaningful without equilibration): balance P and rho A'A
        rho = float(min(max(RHO_AUTO_SCALE * np.sqrt(np.trace(Ps) / np.trace(As.T @ As)), RHO_MIN), RHO_MAX))

    def rho_vec(r):
        return np.where(loose_k, RHO_MIN, np.where(eq, RHO_EQ_OVER_RHO_INEQ * r, r))

    def factor(rv):
        return scipy.linalg.cho_factor(Ps + sigma * np.eye(n) + As.T @ (rv[:, None] * As))

    rv = rho_vec(rho)
    cho = factor(rv)
    x, z, y = np.zeros(n), np.zeros(m), np.zeros(m)
    Einv, Dinv, cinv = 1.0 / Ek, 1.0 / D, 1.0 / c
    it, rp, rd = 0, np.inf, np.inf
    infeasible = converged = False
    for it in range(1, o["max_iter"] + 1):
        xt = scipy.linalg.cho_solve(cho, sigma * x - qs + As.T @ (rv * z - y))
        zt = As @ xt
        x = alpha * xt + (1 - alpha) * x
        zr = alpha * zt + (1 - alpha) * z
        z_new = np.clip(zr + y / rv, ls, us)
        dy = rv * (zr - z_new)
        y = y + dy
        z = z_new
        if it % o["check_every"] == 0 or it == o["max_iter"]:
            Ax, Px, Aty = As @ x, Ps @ x, As.T @ y
            rp = np.abs(Einv * (Ax - z)).max()                               # compute_pri_res (unscaled)
            eps_p = o["eps_abs"] + o["eps_rel"] * max(np.abs(Einv * z).max(), np.abs(Einv * Ax).max())
            rd = cinv * np.abs(Dinv * (Px + qs + Aty)).max()                 # compute_dua_res (unscaled)
            eps_d = o["eps_abs"] + o["eps_rel"] * cinv * max(np.abs(Dinv * qs).max(), np.abs(Dinv * Aty).max(),
                                                              np.abs(Dinv * Px).max())
            if rp < eps_p and rd < eps_d:
                converged = True
                break
            # is_primal_infeasible
            ndy = np.abs(Ek * dy).max()
            if ndy > o["eps_prim_inf"]:
                supp = np.sum(us * np.maximum(dy, 0) + ls * np.minimum(dy, 0))
                if supp < -o["eps_prim_inf"] * ndy and np.abs(Dinv * (As.T @ dy)).max() < o["eps_prim_inf"] * ndy:
                    infeasible = True
                    break
            if o["adaptive_rho"] and it % o["rho_every"] == 0 and it < o["max_iter"]:
                # compute_rho_estimate: SCALED residuals, normalised
                pr = np.abs(Ax - z).max() / (max(np.abs(z).max(), np.abs(Ax).max()) + 1e-10)
                dr = np.abs(Px + qs + Aty).max() / (max(np.abs(qs).max(), np.abs(Aty).max(), np.abs(Px).max()) + 1e-10)
                new = min(max(rho * np.sqrt(pr / (dr + 1e-10)), RHO_MIN), RHO_MAX)
                if new > rho * o["adaptive_rho_tolerance"] or new < rho / o["adaptive_rho_tolerance"]:
                    rho = new
                    rv = rho_vec(rho)
                    cho = factor(rv)
    yfull = np.zeros(len(keep))
    yfull[keep] = cinv * Ek * y                                              # unscale_solution
    xu = D * x
    if infeasible:
        xu = np.full(n, np.nan)
    return dict(x=xu, y=yfull, iters=it, r_prim=rp, r_dual=rd, rho=rho, infeasible=infeasible, converged=converged,
                D=D, E=E, c=c)


def qp_exact(P, q, A, l, u, tol=1e-9, max_rounds=50):
    """Exact minimiser of  1/2 x'Px + q'x  s.t. l <= Ax <= u  (P > 0) by primal-dual active-set
    iteration on the KKT system, seeded from a tight ADMM run; raises if KKT is not met."""
    n = P.shape[0]
    seed = admm_osqp_style(P, q, A, l, u, eps_abs=1e-10, eps_rel=1e-10, max_iter=400000)
    x, y = seed["x"], seed["y"]
    Ax = A @ x
    act_lo = (Ax - l < 1e-6) & (y < -1e-9) & np.isfinite(l)
    act_hi = (u - Ax < 1e-6) & (y > 1e-9) & np.isfinite(u)
    for _ in range(max_rounds):
        idx = np.nonzero(act_lo | act_hi)[0]
        b = np.where(act_lo[idx], l[idx], u[idx])
        Aa = A[idx]
        k = len(idx)
        KKT = np.block([[P, Aa.T], [Aa, np.zeros((k, k))]])
        sol = np.linalg.lstsq(KKT, np.concatenate((-q, b)), rcond=None)[0] if k else np.linalg.solve(P, -q)
        x = sol[:n]
        lam = np.zeros(A.shape[0])
        if k:
            lam[idx] = sol[n:]
        Ax = A @ x
        viol_lo = (Ax < l - tol) & ~act_lo
        viol_hi = (Ax > u + tol) & ~act_hi
        wrong_lo = act_lo & (lam > tol)
        wrong_hi = act_hi & (lam < -tol)
        if not (viol_lo.any() or viol_hi.any() or wrong_lo.any() or wrong_hi.any()):
            stat = np.abs(P @ x + q + A.T @ lam).max()
            if stat > 1e-7 * max(1.0, np.abs(q).max()):
                raise RuntimeError(f"qp_exact: stationarity {stat}")
            return x, lam
        act_lo = (act_lo | viol_lo) & ~wrong_lo
        act_hi = (act_hi | viol_hi) & ~wrong_hi
    # degenerate active set (linearly dependent rate/command/state rows): fall back to the tight ADMM point,
    # accepted only if its own KKT residuals are at rounding level
    if seed["r_prim"] < 1e-8 and seed["r_dual"] < 1e-8 * max(1.0, np.abs(q).max()):
        return seed["x"], seed["y"]
    raise RuntimeError("qp_exact: active set did not settle")
