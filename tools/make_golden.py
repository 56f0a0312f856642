#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Build-container only.  It (a) loads the reference's prebuilt C/nlplant_xcg25.so and
C/nlplant_xcg35.so with ctypes (CWD must be the reference root because the C opens
"C/<table>.dat" by relative path, hifi_F16_AeroData.c:8), and (b) imports the reference's
env.py / utils.py / parameters.py under empty stub modules for the five packages that are
absent offline (gym, ursina, progressbar, osqp, control.matlab) with np.infty restored
(parameters.py:122 uses it; NumPy 2 removed it).  Nothing of the reference is copied: only
inputs and the outputs it computed are stored.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import ctypes
import os
import sys
import types

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = os.environ.get("F16_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

TABLE_FNS = [  # reference symbol per table id (same order as tools/pack_tables.py HIFI)
    "_Cx", "_Cz", "_Cm", "_Cn", "_Cl", "_Cy", "_Cy_r30", "_Cn_r30", "_Cl_r30", "_Cy_a20", "_Cn_a20", "_Cl_a20",
    "_Cx_lef", "_Cz_lef", "_Cm_lef", "_Cy_lef", "_Cn_lef", "_Cl_lef", "_Cy_a20_lef", "_Cn_a20_lef", "_Cl_a20_lef",
    "_CXq", "_CYr", "_CYp", "_CZq", "_CLr", "_CLp", "_CMq", "_CNr", "_CNp", "_delta_CNbeta", "_delta_CLbeta",
    "_delta_Cm", "_delta_CXq_lef", "_delta_CYr_lef", "_delta_CYp_lef", "_delta_CZq_lef", "_delta_CLr_lef",
    "_delta_CLp_lef", "_delta_CMq_lef", "_delta_CNr_lef", "_delta_CNp_lef", "_eta_el"]
TABLE_NARGS = [3] * 5 + [2] * 16 + [1] * 22
ALPHA1 = np.array([-20, -15, -10, -5, 0, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 55, 60, 70, 80, 90.])
BETA1 = np.array([-30, -25, -20, -15, -10, -8, -6, -4, -2, 0, 2, 4, 6, 8, 10, 15, 20, 25, 30.])
DH1 = np.array([-25, -10, 0, 10, 25.])


def import_reference():
    os.chdir(REF)
    for name in ("gym", "ursina", "progressbar", "osqp", "control", "control.matlab"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["gym"].Env = type("Env", (), {})
    sys.modules["gym"].spaces = types.ModuleType("gym.spaces")
    sys.modules["gym.spaces"] = sys.modules["gym"].spaces
    sys.modules["control"].matlab = sys.modules["control.matlab"]
    if not hasattr(np, "infty"):
        np.infty = np.inf
    sys.path.insert(0, REF)
    import parameters  # noqa
    import env  # noqa
    import utils  # noqa
    return parameters, env, utils


def dptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def random_states(rng, n, lofi=False):
    """In-grid xu[18] samples (units of parameters.py:119)."""
    x = np.zeros((n, 18))
    x[:, 0:2] = rng.uniform(-1e3, 1e3, (n, 2))
    x[:, 2] = rng.uniform(0, 4e4, n)
    x[:, 3:6] = rng.uniform(-1, 1, (n, 3))
    x[:, 6] = rng.uniform(200, 900, n)
    x[:, 7] = np.deg2rad(rng.uniform(-9.5 if lofi else -19, 44, n))
    x[:, 8] = np.deg2rad(rng.uniform(-29, 29, n))
    x[:, 9:12] = rng.uniform(-1, 1, (n, 3))
    x[:, 12] = rng.uniform(1000, 19000, n)
    x[:, 13] = rng.uniform(-24.5, 24.5, n)
    x[:, 14] = rng.uniform(-21, 21, n)
    x[:, 15] = rng.uniform(-29, 29, n)
    x[:, 16] = rng.uniform(0, 25, n)
    x[:, 17] = rng.uniform(-20, 5, n)
    # exact-node cases: beta = 0, el in {0, +-10, 25}, alpha = 0 (exact zeros survive the rad->deg product)
    k = n // 8
    x[:k, 8] = 0.0
    x[k:2 * k, 13] = rng.choice([0.0, -10.0, 10.0, 25.0, -25.0], k)
    x[2 * k:2 * k + k // 2, 7] = 0.0
    x[2 * k + k // 2:3 * k, [7, 8, 13]] = 0.0
    return x


def main():
    os.makedirs(OUT, exist_ok=True)
    parameters, env, utils = import_reference()
    from oracle import mpc_oracle as mo

    libs = {25: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg25.so")),
            35: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg35.so"))}
    rng = np.random.default_rng(20261003)

    # ---------------- G1: every table function
    lib = libs[25]
    pts_all, vals_all = [], []
    for tid, (fn, na) in enumerate(zip(TABLE_FNS, TABLE_NARGS)):
        f = getattr(lib, fn)
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_double] * na
        amax = 45.0 if "lef" in fn else 90.0
        n = 64
        a = rng.uniform(-20, amax, n)
        b = rng.uniform(-30, 30, n)
        e = rng.uniform(-25, 25, n)
        a[:8] = rng.choice(ALPHA1[ALPHA1 <= amax], 8)          # nodes
        b[4:12] = rng.choice(BETA1, 8)
        e[8:16] = rng.choice(DH1, 8)
        a[16], a[17] = -20.0, amax                              # grid edges
        b[18], b[19] = -30.0, 30.0
        e[20], e[21] = -25.0, 25.0
        if fn == "_eta_el":
            vals = [f(e[i]) for i in range(n)]
        elif na == 1:
            vals = [f(a[i]) for i in range(n)]
        elif na == 2:
            vals = [f(a[i], b[i]) for i in range(n)]
        else:
            if fn in ("_Cn", "_Cl"):
                pass  # DH2 grid {-25,0,25}: DH1 nodes +-10 are interior points there
            vals = [f(a[i], b[i], e[i]) for i in range(n)]
        pts_all.append(np.stack([a, b, e], 1))
        vals_all.append(np.array(vals))
    np.savez_compressed(os.path.join(OUT, "g1_tables.npz"), pts=np.array(pts_all), vals=np.array(vals_all),
                        names=np.array(TABLE_FNS))

    # ---------------- G2: Nlplant + atmos
    def nlplant(lib, xu, fi):
        out = np.zeros((len(xu), 18))
        for i in range(len(xu)):
            row = np.ascontiguousarray(xu[i])
            lib.Nlplant(dptr(row), dptr(out[i]), ctypes.c_int(fi))
        return out
    xu_h = random_states(rng, 1000)
    xu_l = random_states(rng, 400, lofi=True)
    x0 = np.copy(parameters.x0)
    xu_h[0] = x0
    xu_l[0] = x0
    alt = rng.uniform(0, 6e4, 64)
    vt = rng.uniform(0.01, 1500, 64)
    alt[0], vt[0] = 35000.0, 500.0
    atm = np.zeros((64, 3))
    for i in range(64):
        libs[25].atmos(ctypes.c_double(alt[i]), ctypes.c_double(vt[i]), dptr(atm[i]))
    np.savez_compressed(os.path.join(OUT, "g2_nlplant.npz"), xu_hifi=xu_h, xu_lofi=xu_l,
                        xdot_hifi_xcg25=nlplant(libs[25], xu_h, 1), xdot_hifi_xcg35=nlplant(libs[35], xu_h, 1),
                        xdot_lofi_xcg25=nlplant(libs[25], xu_l, 0), xdot_lofi_xcg35=nlplant(libs[35], xu_l, 0),
                        atmos_in=np.stack([alt, vt], 1), atmos_out=atm)

    # ---------------- reference F16 objects (trim + linearise at construction, env.py:31-60)
    def make_f16(stab):
        P = parameters
        sv = P.stateVector(P.states, np.copy(P.x0), P.x_units, P.x_ub, P.x_lb, np.copy(P.x0), P.observed_states,
                           P.mpc_states, P.mpc_inputs, P.mpc_controlled_states)
        iv = P.inputVector(P.inputs, np.copy(P.u0), P.u_units, P.u_ub, P.u_lb, P.udot_ub, P.udot_lb, np.copy(P.u0),
                           P.mpc_inputs)
        sp = P.simulationParameters(P.dt, P.time_start, P.time_end, stab, 1)
        ss = P.stateSpace(*[np.zeros((1, 1))] * 8)
        return env.F16(sv, iv, sp, ss, libs[35 if stab else 25])
    f16 = {25: make_f16(0), 35: make_f16(1)}

    # ---------------- G5/G6/G7: trim, linearisations, discretisation, LQR gain
    g567 = {}
    for k, f in f16.items():
        g567[f"trim_x_xcg{k}"] = np.copy(f.x.initial_condition)
        g567[f"A18_xcg{k}"], g567[f"B18_xcg{k}"] = f.ss.Ac, f.ss.Bc
        g567[f"C18_xcg{k}"], g567[f"D18_xcg{k}"] = f.ss.Cc, f.ss.Dc
        g567[f"Ad18_xcg{k}"], g567[f"Bd18_xcg{k}"] = f.ss.Ad, f.ss.Bd
        for nm in ("Ac", "Bc", "Cc", "Dc", "Ad", "Bd", "Cd", "Dd"):
            g567[f"ssr_{nm}_xcg{k}"] = getattr(f.ssr, nm)
        g567[f"K_lqr_xcg{k}"] = f._calc_LQR_gain()
    # LQR action sample (env.py:360-371)
    f = f16[25]
    x9 = f.x._get_mpc_x() + rng.uniform(-0.01, 0.01, 9)
    g567["lqr_action_x9"] = x9
    g567["lqr_action_u"] = f._calc_LQR_action(0.1, -0.05, 0.02, g567["K_lqr_xcg25"], x9, f.u.initial_condition[1:])
    np.savez_compressed(os.path.join(OUT, "g567_trim_lin_lqr.npz"), **g567)

    # ---------------- G3: _calc_xdot / _calc_xdot_na through the reference Python
    xs = random_states(rng, 256)
    us = np.stack([rng.uniform(0, 20000, 256), rng.uniform(-30, 30, 256), rng.uniform(-25, 25, 256),
                   rng.uniform(-35, 35, 256)], 1)
    us[:64] = xs[:64, 12:16] + rng.uniform(-1, 1, (64, 4))       # unsaturated rates
    g3 = dict(x=xs, u=us)
    for k, f in f16.items():
        g3[f"xdot_xcg{k}"] = np.array([f._calc_xdot(xs[i], us[i]) for i in range(256)])
        x_full = np.copy(f.x.values)
        x9s = xs[:, mo.MPC_X_IDX]
        u3s = xs[:, mo.MPC_U_IN_X_IDX]
        g3[f"na_x_full_xcg{k}"] = x_full
        g3[f"xdot_na_xcg{k}"] = np.array([f._calc_xdot_na(x9s[i], u3s[i]) for i in range(256)])
    g3["na_x9"], g3["na_u3"] = x9s, u3s
    np.savez_compressed(os.path.join(OUT, "g3_calc_xdot.npz"), **g3)

    # ---------------- G4: Euler rollouts through F16.step (env.py:105-130)
    g4 = {}
    for k, f in f16.items():
        f.reset()
        u = np.copy(f.u.values)
        traj = []
        for t in range(1000):
            f.step(u)
            if (t + 1) % 50 == 0:
                traj.append(np.copy(f.x.values))
        g4[f"trim_traj_xcg{k}"] = np.array(traj)
        g4[f"trim_u_xcg{k}"] = u
        # perturbed starts + stepped commands
        starts, trajs, ucmd = [], [], []
        for j in range(8):
            f.reset()
            xs0 = np.copy(f.x.values)
            xs0[2] += rng.uniform(-3000, 8000)
            xs0[6] += rng.uniform(-150, 150)
            xs0[3:6] += rng.uniform(-0.15, 0.15, 3)
            xs0[7] += rng.uniform(-0.03, 0.12)
            xs0[8] += rng.uniform(-0.05, 0.05)
            xs0[9:12] += rng.uniform(-0.2, 0.2, 3)
            f.x.values = np.copy(xs0)
            u = np.copy(f.u.initial_condition) + np.array([rng.uniform(-500, 3000), rng.uniform(-3, 3),
                                                           rng.uniform(-5, 5), rng.uniform(-5, 5)])
            tr = []
            for t in range(300):
                f.step(u)
                if (t + 1) % 25 == 0:
                    tr.append(np.copy(f.x.values))
            starts.append(xs0), trajs.append(np.array(tr)), ucmd.append(u)
        g4[f"pert_x0_xcg{k}"], g4[f"pert_traj_xcg{k}"], g4[f"pert_u_xcg{k}"] = map(np.array, (starts, trajs, ucmd))
        f.reset()
    np.savez_compressed(os.path.join(OUT, "g4_rollout.npz"), **g4)

    # ---------------- G8/G9: MPC matrices + QP data + exact minimisers
    g8 = {}
    # Cannon C21 toy system used by notes_examples/example_2_1.py:28-49
    At, Bt, Ct = np.array([[1.1, 2], [0, 0.95]]), np.array([[0], [0.0787]]), np.array([[-1.0, 1.0]])
    MMt, CCt = utils.calc_MC(At, Bt, 1, 4)
    Qt, Rt = Ct.T @ Ct, np.eye(1) * 0.01
    g8["toy_MM"], g8["toy_CC"] = MMt, CCt
    g8["toy_H"] = CCt.T @ utils.dmom(Qt, 4) @ CCt + utils.dmom(Rt, 4)
    g8["toy_K"] = utils.dlqr(At, Bt, Qt, Rt)
    for k, f in f16.items():
        f.reset()
        for N in (4, 10, 30):
            x = f.x._get_mpc_x()
            act = f.x._get_mpc_act_states()
            x_ref = np.copy(x)
            x_ref[5:8] = [0.0, 0.0, 0.0]
            A, B, C = f.ssr.Ad, f.ssr.Bd, f.ssr.Cd
            Pm, q, Ac, l, u = utils.setup_OSQP(x_ref, A, B, C.T @ C, np.eye(3), N, f.paras.dt, x, act,
                                               f.x._vec_mpc_x_lb, f.x._vec_mpc_x_ub, f.u._vec_mpc_u_lb,
                                               f.u._vec_mpc_u_ub, f.u._vec_mpc_udot_lb, f.u._vec_mpc_udot_ub)
            tag = f"xcg{k}_N{N}"
            g8[f"P_{tag}"], g8[f"q_{tag}"], g8[f"A_{tag}"] = Pm, q.ravel(), Ac
            g8[f"l_{tag}"], g8[f"u_{tag}"] = l.ravel(), u.ravel()
            xs_, lam = mo.qp_exact(Pm, q.ravel(), Ac, l.ravel(), u.ravel())
            g8[f"xstar_{tag}"], g8[f"lam_{tag}"] = xs_, lam
        MM, CC = utils.calc_MC(f.ssr.Ad, f.ssr.Bd, f.paras.dt, 10)
        g8[f"MM10_xcg{k}"], g8[f"CC10_xcg{k}"] = MM, CC
    # drop the big dense constraint matrices' redundancy: A is rebuilt from CC in tests for N=30
    np.savez_compressed(os.path.join(OUT, "g8_mpc_qp.npz"), **g8)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


G10_FILES = [  # (file under the reference root, cg location it was produced with, surface doublet in degrees)
    ("C/ele_0.100ail_0.100rud_0.100_hifimodel_alt10000_vel300.txt", 0.30, 0.1),
    ("Nguyen_m/ele_0.000ail_0.000rud_0.000_hifimodel_alt10000_vel500.txt", 0.30, 0.0),
    ("Nguyen_m/ele_0.000ail_0.000rud_0.000_hifimodel_alt10000_vel600.txt", 0.30, 0.0),
    ("Nguyen_m/ele_0.000ail_0.000rud_0.000_hifimodel_alt10000_vel700.txt", 0.25, 0.0)]


def g10_time_histories(seconds=10.0):
    """G10: the time histories the reference HOLDS as data files -- output of its Simulink driver (Nguyen_m/runF16Sim.m:100-150:
    trim, then 30 s of the nonlinear hifi model at 1 ms, a -1 / +2 / -1 surface doublet between 1, 3 and 5 s, one line per 0.1 s:
    time, 12 states, nx ny nz mach qbar ps, thrust and the three surface positions, every value '%8.5f,').  No reference code
    runs here: the first `seconds` of each file are stored as numbers, with the text of the header and of the first row for the
    writer's format test.  The cg location of each run is not recorded in the file; it is the one at which the file's first row
    is an equilibrium of the plant (0.30 for three of them, 0.25 for the 700 ft/s run).  The C/...alt5000_vel1000 file is left
    out: 1000 ft/s is outside the envelope of env.py:117-124."""
    out = {}
    for k, (rel, xcg, dis) in enumerate(G10_FILES):
        lines = open(os.path.join(REF, rel)).read().split("\n")
        rows = [l for l in lines if l.strip() and l.strip()[0].isdigit()]
        a = np.array([[float(v) for v in l.strip().strip(",").split(",")] for l in rows])
        n = int(round(seconds / 0.1)) + 1
        out[f"rows_{k}"] = a[:n]
        out[f"xcg_{k}"] = xcg
        out[f"doublet_{k}"] = dis
        out[f"name_{k}"] = os.path.basename(rel)
        if k == 0:
            out["header_line"] = [l for l in lines if l.startswith("time,")][0]
            out["first_row_text"] = rows[0]
            out["second_row_text"] = rows[1]
    np.savez_compressed(os.path.join(OUT, "g10_time_histories.npz"), **out)
    print("g10_time_histories.npz", os.path.getsize(os.path.join(OUT, "g10_time_histories.npz")))


def g11_state_space():
    """G11: the linearised 18-state models the reference HOLDS as a data file (Nguyen_m/StateSpace_alt10000_vel700.txt: A, B of the
    hifi and of the lofi model at 10,000 ft / 700 ft/s, printed '%8.5f,'; states in the order of parameters.py:116-119 incl. the
    actuator and flap states) and the trim points they were taken at (first rows of the 700 ft/s time-history files of the same
    directory).  Numbers only; no reference code runs."""
    import re
    txt = open(os.path.join(REF, "Nguyen_m", "StateSpace_alt10000_vel700.txt")).read()
    out = {}
    for m in re.finditer(r"(\w+) = \n((?:[ \-\d\.,e]+\n)+)", txt):
        out[m.group(1)] = np.array([[float(v) for v in l.strip().strip(",").split(",")] for l in m.group(2).strip().split("\n")])
    for tag, rel in (("hi", "ele_0.000ail_0.000rud_0.000_hifimodel_alt10000_vel700.txt"),
                     ("lo", "ele_0.000ail_0.000rud_0.000_lofimodel_alt10000_vel700_LTI.txt")):
        rows = [l for l in open(os.path.join(REF, "Nguyen_m", rel)).read().split("\n") if l.strip() and l.strip()[0].isdigit()]
        out["trim_row_" + tag] = np.array([float(v) for v in rows[0].strip().strip(",").split(",")])
    import scipy.io                      # MATLAB_SS.mat: what the reference's own test loads (test_env.py:186) -- the LOFI model of the
    mat = scipy.io.loadmat(os.path.join(REF, "MATLAB_SS.mat"))      # same flight condition in full precision (= A_lo, B_lo to 5e-6)
    out["A_mat"], out["B_mat"] = mat["A"], mat["B"]
    np.savez_compressed(os.path.join(OUT, "g11_state_space.npz"), **out)
    print("g11_state_space.npz", os.path.getsize(os.path.join(OUT, "g11_state_space.npz")), sorted(out))


def make_f16_objects(parameters, env, libs):
    def make_f16(stab):
        P = parameters
        sv = P.stateVector(P.states, np.copy(P.x0), P.x_units, P.x_ub, P.x_lb, np.copy(P.x0), P.observed_states,
                           P.mpc_states, P.mpc_inputs, P.mpc_controlled_states)
        iv = P.inputVector(P.inputs, np.copy(P.u0), P.u_units, P.u_ub, P.u_lb, P.udot_ub, P.udot_lb, np.copy(P.u0),
                           P.mpc_inputs)
        sp = P.simulationParameters(P.dt, P.time_start, P.time_end, stab, 1)
        ss = P.stateSpace(*[np.zeros((1, 1))] * 8)
        return env.F16(sv, iv, sp, ss, libs[35 if stab else 25])
    return {25: make_f16(0), 35: make_f16(1)}


def clr_reads_zero(lib):
    """The reference's `_CLr` interpolates memory it never filled (hifi_F16_AeroData.c:964-972): every fixture of this
    repository is taken in a process where that memory reads ~0 (G2 was) -- checked, not assumed."""
    f = lib._CLr
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_double]
    return all(abs(f(a)) < 1e-290 for a in np.linspace(-20, 89, 45))


def g12_lqr_loop_and_g8b_weights():
    """G12: the reference's nonlinear closed loop under its LQR controller (test_env_mk2.py:70-85: K = _calc_LQR_gain() once,
    then per step u = _calc_LQR_action(p, q, r, K, x._get_mpc_x(), u.initial_condition[1:]); u.values[1:] = u; step(u.values)),
    300 steps, both xcg builds: from trim with rate demands, and from perturbed starts.  Every 25th state + the last action.
    G8b: utils.setup_OSQP (utils.py:21-167) called with weights, reference and bounds OTHER than the ones env.py:373-424 hard-wires:
    the weights the author left commented out at env.py:391-403 with R = 0.01 I, and a dense SPD Q / R with a free reference and
    tightened / widened boxes."""
    parameters, env, utils = import_reference()
    libs = {25: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg25.so")),
            35: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg35.so"))}
    f16 = make_f16_objects(parameters, env, libs)
    for k in libs:
        if not clr_reads_zero(libs[k]):
            raise SystemExit("this process's _CLr does not read ~0 (uninitialised heap): run the script again")
    rng = np.random.default_rng(20261005)
    g12 = {}
    for k, f in f16.items():
        f.reset()
        K = f._calc_LQR_gain()
        g12[f"K_xcg{k}"] = K
        g12[f"u0_xcg{k}"] = np.copy(f.u.initial_condition)
        cases_x0, cases_dem, cases_traj, cases_u = [], [], [], []
        for j in range(6):
            f.reset()
            x0 = np.copy(f.x.values)
            if j == 0:
                dem = np.zeros(3)                                   # test_env_mk2.py:37-39 as written: hold the trim
            elif j == 1:
                dem = np.array([0.1, -0.05, 0.02])                  # the sample of fixture G7's action
            else:
                dem = rng.uniform(-0.15, 0.15, 3)
                x0[3:6] += rng.uniform(-0.1, 0.1, 3)
                x0[6] += rng.uniform(-60, 60)
                x0[7] += rng.uniform(-0.02, 0.06)
                x0[8] += rng.uniform(-0.03, 0.03)
                x0[9:12] += rng.uniform(-0.15, 0.15, 3)
            f.x.values = np.copy(x0)
            f.u.values = np.copy(f.u.initial_condition)
            tr = []
            for t in range(300):
                u = f._calc_LQR_action(dem[0], dem[1], dem[2], K, f.x._get_mpc_x(), f.u.initial_condition[1:])
                f.u.values[1:] = u
                f.step(f.u.values)
                if (t + 1) % 25 == 0:
                    tr.append(np.copy(f.x.values))
            cases_x0.append(x0), cases_dem.append(dem), cases_traj.append(np.array(tr)), cases_u.append(np.copy(f.u.values))
            f.u.values = np.copy(f.u.initial_condition)
        g12[f"x0_xcg{k}"], g12[f"dem_xcg{k}"] = np.array(cases_x0), np.array(cases_dem)
        g12[f"traj_xcg{k}"], g12[f"u_last_xcg{k}"] = np.array(cases_traj), np.array(cases_u)
        f.reset()
    for k in libs:
        if not clr_reads_zero(libs[k]):
            raise SystemExit("_CLr stopped reading ~0 during the run: run the script again")
    np.savez_compressed(os.path.join(OUT, "g12_lqr_loop.npz"), **g12)
    print("g12_lqr_loop.npz", os.path.getsize(os.path.join(OUT, "g12_lqr_loop.npz")))

    g8b = {}
    Qa = np.diag([0.01, 0.0, 0.01, 0.01, 0.0, 1.0, 1.0, 1.0, 0.0])      # env.py:391-401 (the tenth line indexes past a 9 x 9 Q)
    Ra = np.eye(3) * 0.01                                               # env.py:403
    M = rng.uniform(-1, 1, (9, 9))
    Qb = M @ M.T / 9 + np.diag(rng.uniform(0.05, 1.0, 9))
    M3 = rng.uniform(-1, 1, (3, 3))
    Rb = M3 @ M3.T / 3 + np.diag(rng.uniform(0.1, 2.0, 3))
    g8b["Qa"], g8b["Ra"], g8b["Qb"], g8b["Rb"] = Qa, Ra, Qb, Rb
    for k, f in f16.items():
        f.reset()
        x = f.x._get_mpc_x()
        act = f.x._get_mpc_act_states()
        A, B = f.ssr.Ad, f.ssr.Bd
        for tag, Q, R in (("a", Qa, Ra), ("b", Qb, Rb)):
            if tag == "a":
                x_ref = np.copy(x)
                x_ref[5:8] = [0.05, -0.02, 0.01]
                bnds = (f.x._vec_mpc_x_lb, f.x._vec_mpc_x_ub, f.u._vec_mpc_u_lb, f.u._vec_mpc_u_ub,
                        f.u._vec_mpc_udot_lb, f.u._vec_mpc_udot_ub)
            else:
                x_ref = x + rng.uniform(-0.05, 0.05, 9)              # a free reference, not "x with three entries replaced"
                xlb = np.array([-np.inf, -np.inf, -10., -15., -200., -60., -30., -np.inf, 0.])
                xub = np.array([np.inf, np.inf, 40., 15., 200., 60., 30., np.inf, 20.])
                bnds = tuple(np.asarray(v)[:, None] for v in (                # vertical vectors, as utils.py:57-68 wants them
                    xlb, xub, np.array([-20., -15., -25.]), np.array([22., 18., 28.]),
                    np.array([-50., -70., -100.]), np.array([55., 75., 110.])))
            g8b[f"xref_{tag}_xcg{k}"] = x_ref
            for nm, v in zip(("xlb", "xub", "ulb", "uub", "rlb", "rub"), bnds):
                g8b[f"{nm}_{tag}"] = np.asarray(v, dtype=float).ravel()
            for N in (4, 10, 30):
                Pm, q, Ac, l, u = utils.setup_OSQP(x_ref, A, B, Q, R, N, f.paras.dt, x, act, *bnds)
                t = f"{tag}_xcg{k}_N{N}"
                g8b[f"P_{t}"], g8b[f"q_{t}"], g8b[f"l_{t}"], g8b[f"u_{t}"] = Pm, q.ravel(), l.ravel(), u.ravel()
                if N < 30:
                    g8b[f"A_{t}"] = Ac
            g8b[f"K_{tag}_xcg{k}"] = utils.dlqr(A, B, Q, R)             # utils.py:219 with the same weights
        g8b[f"x_full_xcg{k}"] = np.copy(f.x.values)
        g8b[f"Ad_xcg{k}"], g8b[f"Bd_xcg{k}"], g8b[f"Cd_xcg{k}"] = f.ssr.Ad, f.ssr.Bd, f.ssr.Cd
    np.savez_compressed(os.path.join(OUT, "g8b_mpc_qp_weights.npz"), **g8b)
    print("g8b_mpc_qp_weights.npz", os.path.getsize(os.path.join(OUT, "g8b_mpc_qp_weights.npz")))


def g13_round5():
    """Round-5 fixtures, generated by RUNNING the reference (no osqp needed).
    G3b: NaN through the actuator models -- `_calc_xdot` (env.py:65-103; np.clip of utils.py:303-330 PROPAGATES NaN) with NaN in
    each surface command, in the thrust command and in the flap state lf1, +-inf and far-out-of-box commands, and ONE `step`
    (env.py:105-130) under a NaN elevator command: what the reference does with the NaN command OSQP hands back for an infeasible
    QP (env.py:420-424).  (Every case keeps the plant's own inputs xu[0:17] finite: with NaN INSIDE them the reference's interpn
    finds no bracket, mexndinterp.c:125-138, and reads an undefined cell -- which is why a second step is not taken.)
    G13: the linear-model closed loops -- what main.py:35 runs (test_env_mk2.py:46-62 `LQR(linear=True)`: x = ssr.Ad @ x + ssr.Bd @ u
    under `_calc_LQR_action`, 10,000 steps at zero demands as written, plus rate demands and a perturbed start) and
    test_env.py:501-576 `test_LQR_lin` (f16 model: K = dlqr(A, B, I, I), u = -K (x - x_ref), x = A x + B u; and its toy double
    integrator).  Every 100th state / action of the long runs, every step of the toy."""
    parameters, env, utils = import_reference()
    libs = {25: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg25.so")),
            35: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg35.so"))}
    f16 = make_f16_objects(parameters, env, libs)
    for k in libs:
        if not clr_reads_zero(libs[k]):
            raise SystemExit("this process's _CLr does not read ~0 (uninitialised heap): run the script again")
    rng = np.random.default_rng(20261006)
    nan = float("nan")
    g3b = {}
    for k, f in f16.items():
        f.reset()
        x0 = np.copy(f.x.values)
        u0 = np.copy(f.u.values)
        xs, us, xd = [], [], []
        for case in range(10):
            x, u = np.copy(x0), np.copy(u0)
            x[3:6] += rng.uniform(-0.05, 0.05, 3)
            x[13:16] += rng.uniform(-1, 1, 3)
            if case < 4:
                u[case] = nan                   # NaN thrust / elevator / aileron / rudder command
            elif case == 4:
                u[1:] = nan                     # what env.py:424 returns for an infeasible QP
            elif case == 5:
                u[1] = np.inf                   # np.clip(inf) = the bound
            elif case == 6:
                u[2] = -np.inf
            elif case == 7:
                x[17] = nan                     # lf1 (not a plant input: xu[0:17]) -> both flap derivatives
            elif case == 8:
                u[:] = [1e9, -1e9, 1e9, -1e9]   # far outside the command boxes
            xs.append(np.copy(x)), us.append(np.copy(u)), xd.append(np.copy(f._calc_xdot(x, u)))
        g3b[f"x_xcg{k}"], g3b[f"u_xcg{k}"], g3b[f"xdot_xcg{k}"] = np.array(xs), np.array(us), np.array(xd)
        # one step under a NaN elevator command (the envelope test of env.py:117-124 lets NaN through: every comparison is False)
        f.reset()
        f.u.values[1] = nan
        f.step(f.u.values)
        g3b[f"step1_xcg{k}"] = np.copy(f.x.values)
        g3b[f"step_u_xcg{k}"] = np.copy(f.u.values)
        f.reset()
    for k in libs:
        if not clr_reads_zero(libs[k]):
            raise SystemExit("_CLr stopped reading ~0 during the run: run the script again")
    np.savez_compressed(os.path.join(OUT, "g3b_nan_actuators.npz"), **g3b)
    print("g3b_nan_actuators.npz", os.path.getsize(os.path.join(OUT, "g3b_nan_actuators.npz")))

    from scipy.signal import cont2discrete
    g13 = {}
    T = 10000                                                     # test_env_mk2.py:28-30: time_end 10 at dt 0.001
    for k, f in f16.items():
        f.reset()
        K = f._calc_LQR_gain()                                    # test_env_mk2.py:37
        g13[f"K_xcg{k}"], g13[f"Ad_xcg{k}"], g13[f"Bd_xcg{k}"] = K, np.copy(f.ssr.Ad), np.copy(f.ssr.Bd)
        x_init = f.x._get_mpc_x()
        u0 = np.copy(f.u.initial_condition[1:])
        g13[f"u0_xcg{k}"] = u0
        x0s, dems, xt, ut = [], [], [], []
        for case in range(3):
            x = np.copy(x_init)
            dem = np.zeros(3)                                     # test_env_mk2.py:40-42 as written
            if case == 1:
                dem = np.array([0.1, -0.05, 0.02])
            elif case == 2:
                dem = np.array([-0.05, 0.08, 0.03])
                x[:7] += rng.uniform(-0.05, 0.05, 7)
            x0s.append(np.copy(x)), dems.append(dem)
            xs_, us_ = [], []
            for idx in range(T):                                  # test_env_mk2.py:54-62
                u = f._calc_LQR_action(dem[0], dem[1], dem[2], K, x, u0)
                x = f.ssr.Ad @ x + f.ssr.Bd @ u
                if (idx + 1) % 100 == 0:
                    xs_.append(np.copy(x)), us_.append(np.copy(u))
            xt.append(np.array(xs_)), ut.append(np.array(us_))
        g13[f"x0_xcg{k}"], g13[f"dem_xcg{k}"] = np.array(x0s), np.array(dems)
        g13[f"xtraj_xcg{k}"], g13[f"utraj_xcg{k}"] = np.array(xt), np.array(ut)
        # test_env.py:501-576 with f16=True (the trim of the object as built, env.py:42: same call as test_env.py:506)
        f.reset()
        A, B, C, D = f.linearise(f.x._get_mpc_x(), f.u._get_mpc_u(), _calc_xdot=f._calc_xdot_na, get_obs=f._get_obs_na)
        A, B, C, D = cont2discrete((A, B, C, D), f.paras.dt)[0:4]
        x = f.x._get_mpc_x()[:, None]
        x_ref = np.copy(x)
        Kl = utils.dlqr(A, B, np.eye(9), np.eye(3))
        x = x + rng.uniform(-0.02, 0.02, (9, 1))                  # (as written x = x_ref and u = 0 for ever: start off the reference)
        g13[f"lin_x0_xcg{k}"], g13[f"lin_xref_xcg{k}"], g13[f"lin_K_xcg{k}"] = x[:, 0].copy(), x_ref[:, 0].copy(), Kl
        g13[f"lin_A_xcg{k}"], g13[f"lin_B_xcg{k}"] = A, B
        xs_, us_ = [], []
        for i in range(T):                                        # test_env.py:553-559
            u = - Kl @ (x - x_ref)
            x = A @ x + B @ u
            if (i + 1) % 100 == 0:
                xs_.append(x[:, 0].copy()), us_.append(u[:, 0].copy())
        g13[f"lin_xtraj_xcg{k}"], g13[f"lin_utraj_xcg{k}"] = np.array(xs_), np.array(us_)
    # test_env.py:528-541: the toy double integrator (dt 0.1, 3 s)
    x = np.array([3, 1])[np.newaxis].T
    x_ref = np.array([-3, 0])[np.newaxis].T
    A = np.array([[1, 1.0], [0, 1]])
    B = np.array([0.0, 1])[:, None]
    Kt = utils.dlqr(A, B, np.array([[1.0, 0.0], [0.0, 0.0]]), np.array([[1.0]]))
    xs_, us_ = [], []
    for i in range(30):
        u = - Kt @ (x - x_ref)
        x = A @ x + B @ u
        xs_.append(x[:, 0].copy()), us_.append(u[:, 0].copy())
    g13["toy_K"], g13["toy_xtraj"], g13["toy_utraj"] = Kt, np.array(xs_), np.array(us_)
    np.savez_compressed(os.path.join(OUT, "g13_linear_loops.npz"), **g13)
    print("g13_linear_loops.npz", os.path.getsize(os.path.join(OUT, "g13_linear_loops.npz")))


def g14_dynamic_lqr(steps=150):
    """G14: the reference's per-step RE-LINEARISED LQR loop (test_env.py:625-687 `test_LQR_dynamic_nl`; SURVEY.md 8f-2's pattern): every
    step  A, B = linearise at the CURRENT state (env.py:294-342, reduced model) -> cont2discrete -> K = dlqr(A, B, I, 1e4 I)
    (utils.py:219);  cmd = -K (x9 - x_ref) with x_ref = the initial x9 (no trim offset: the loop as written commands the surfaces towards
    zero);  u.values[1:] = cmd;  step(u.values).  Both xcg builds, `steps` steps from the object's trim point: every command, every 10th
    state, the gain of steps 0 / 50 / last.  (The eigenvalue prints and the second linearisation of the loop body do not enter the state.)"""
    parameters, env, utils = import_reference()
    from scipy.signal import cont2discrete
    libs = {25: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg25.so")),
            35: ctypes.CDLL(os.path.join(REF, "C", "nlplant_xcg35.so"))}
    f16 = make_f16_objects(parameters, env, libs)
    for k in libs:
        if not clr_reads_zero(libs[k]):
            raise SystemExit("this process's _CLr does not read ~0 (uninitialised heap): run the script again")
    g = {}
    for k, f in f16.items():
        f.reset()
        g[f"x0_xcg{k}"], g[f"u0_xcg{k}"] = np.copy(f.x.values), np.copy(f.u.values)
        Q, R = np.eye(9), np.eye(3) * 10000                     # test_env.py:641-642
        x_ref = np.copy(f.x._get_mpc_x())                       # :655
        cmds, xs, Ks = [], [], {}
        for idx in range(steps):
            Ac, Bc, Cc, Dc = f.linearise(f.x._get_mpc_x(), f.u._get_mpc_u(), _calc_xdot=f._calc_xdot_na, get_obs=f._get_obs_na)
            A, B, C, D = cont2discrete((Ac, Bc, Cc, Dc), f.paras.dt)[0:4]
            K = utils.dlqr(A, B, Q, R)
            cmd = (- K @ (f.x._get_mpc_x() - x_ref))
            f.u.values[1:] = cmd
            f.step(f.u.values)
            cmds.append(np.copy(cmd))
            if idx in (0, 50, steps - 1):
                Ks[idx] = np.copy(K)
            if (idx + 1) % 10 == 0:
                xs.append(np.copy(f.x.values))
        g[f"cmd_xcg{k}"], g[f"x_xcg{k}"] = np.array(cmds), np.array(xs)
        for idx, K in Ks.items():
            g[f"K{idx}_xcg{k}"] = K
        f.reset()
    for k in libs:
        if not clr_reads_zero(libs[k]):
            raise SystemExit("_CLr stopped reading ~0 during the run: run the script again")
    np.savez_compressed(os.path.join(OUT, "g14_dynamic_lqr.npz"), **g)
    print("g14_dynamic_lqr.npz", os.path.getsize(os.path.join(OUT, "g14_dynamic_lqr.npz")))


if __name__ == "__main__":
    if "--g14" in sys.argv:
        g14_dynamic_lqr()
    elif "--g13" in sys.argv:
        g13_round5()
    elif "--g12" in sys.argv:
        g12_lqr_loop_and_g8b_weights()
    elif "--g10" in sys.argv:
        g10_time_histories()
        g11_state_space()
    else:
        main()
        g10_time_histories()
        g11_state_space()
