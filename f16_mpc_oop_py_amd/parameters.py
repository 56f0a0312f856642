"""Constants and index maps of the reference's parameters.py (file:line cited per item).

Unlike the reference this module does NOT load a shared object at import time
(parameters.py:108-114 does); the HIP library is loaded by `lib.load()` on first use, and the
choice between nlplant_xcg25.so / nlplant_xcg35.so (stab_flag, parameters.py:31) becomes the
run-time `xcg` argument.
"""
import numpy as np
from numpy import pi

dt, time_start, time_end = 0.001, 0., 10.          # parameters.py:22
fi_flag = 1                                         # :26   1 hifi, 0 lofi
stab_flag = 0                                       # :31   0 -> xcg 0.25, 1 -> xcg 0.35


def xcg_of(stab_flag_):
    """stab_flag -> xcg (README.md:59-63, C/nlplant.c:34)."""
    return 0.35 if stab_flag_ else 0.25


m2f = 3.28084                                       # :102
f2m = 1 / m2f
# initial condition, parameters.py:36-55,105
x0 = np.array([0. * m2f, 0. * m2f, 3048. * m2f, 0., 0., 0., 213.36 * m2f, 1.0721 * pi / 180, 0., 0., 0., 0.,
               2886.6468, -2.0385, -0.087577, -0.03877, 0.3986, -1.0721 * pi / 180 * 180 / pi])
u0 = np.copy(x0[12:16])

states = ['npos', 'epos', 'h', 'phi', 'theta', 'psi', 'V', 'alpha', 'beta', 'p', 'q', 'r', 'T', 'dh', 'da', 'dr',
          'lf2', 'lf1']                             # :116
inputs = ['T', 'dh', 'da', 'dr']                    # :117
x_units = ['ft', 'ft', 'ft', 'rad', 'rad', 'rad', 'ft/s', 'rad', 'rad', 'rad/s', 'rad/s', 'rad/s', 'lb', 'deg', 'deg',
           'deg', 'deg', 'deg']                     # :119
u_units = ['lb', 'deg', 'deg', 'deg']

inf = np.inf
# :59-95,122-123 (mixed units exactly as in the reference)
x_ub = [inf, inf, 100000, inf, inf, inf, 900, 90, 30, 300, 100, 50, 19000, 25, 21.5, 30, 25, inf]
x_lb = [-inf, -inf, 0, -inf, -inf, -inf, 0, -20., -30., -300, -100, -50, 1000, -25, -21.5, -30., 0., -inf]
u_ub = [19000, 25, 21.5, 30]                        # :125
u_lb = [1000, -25, -21.5, -30.]                     # :126
udot_ub = [10000, 60, 80, 120]                      # :128
udot_lb = [-10000, -60, -80, -120]                  # :129

observed_states = ['h', 'phi', 'theta', 'alpha', 'beta', 'p', 'q', 'r', 'lf2', 'lf1']   # :134
mpc_states = ['phi', 'theta', 'alpha', 'beta', 'p', 'q', 'r', 'lf1', 'lf2']              # :135
mpc_inputs = ['dh', 'da', 'dr']                                                         # :136
mpc_controlled_states = ['p', 'q', 'r']                                                 # :137

# index maps, parameters.py:158-183,198-210
obs_x_idx = [states.index(s) for s in observed_states]          # [2,3,4,7,8,9,10,11,16,17]
mpc_x_idx = [states.index(s) for s in mpc_states]               # [3,4,7,8,9,10,11,17,16]
mpc_u_states_idx = [states.index(s) for s in mpc_inputs]        # [13,14,15]
mpc_u_in_x_idx = mpc_u_states_idx
mpc_u_idx = [inputs.index(s) for s in mpc_inputs]               # [1,2,3]
mpc_obs_x_idx = [i for i, s in enumerate(mpc_states) if s in observed_states]   # [0..8]
vec_mpc_x_lb = np.array([x_lb[i] for i in mpc_x_idx])
vec_mpc_x_ub = np.array([x_ub[i] for i in mpc_x_idx])
vec_mpc_u_lb = np.array([u_lb[i] for i in mpc_u_idx])
vec_mpc_u_ub = np.array([u_ub[i] for i in mpc_u_idx])
vec_mpc_udot_lb = np.array([udot_lb[i] for i in mpc_u_idx])
vec_mpc_udot_ub = np.array([udot_ub[i] for i in mpc_u_idx])
