"""Batched mirror of the reference's `F16` environment (env.py:29-436).

Same method names and argument meaning as the reference class, over a leading batch dimension B:
`step / reset / get_obs / _calc_xdot / _calc_xdot_na / linearise / _calc_LQR_gain /
_calc_LQR_action / _calc_MPC_action (calc_MPC_action)`.  Where the reference holds one
`x.values[18]` NumPy vector and calls `nlplant.Nlplant` through ctypes once per evaluation
(env.py:100), this class holds the states of B aircraft resident in HBM, state-major
([18, B] fp64, so that a wavefront's 64 lanes read contiguous memory), and calls the batched
entry points of libf16hip.so through the same ctypes mechanism.  torch is used only to own
device memory and streams.  There is no CPU path in this class.
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib
from . import parameters as P


def _vp(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class F16Batch:
    def __init__(self, x0, u0=None, *, stab_flag=None, xcg=None, fi_flag=P.fi_flag, dt=P.dt, device="cuda:0",
                 flags=0, context=None):
        """x0: [B,18] (or [18]) initial states in the reference's state order/units
        (parameters.py:116-119); u0: [B,4] inputs, default = x0[:,12:16] (env.py:43).
        stab_flag / xcg: parameters.py:31 (1 -> 0.35) or an explicit value."""
        if not torch.cuda.is_available():
            raise _lib.F16HipError("F16Batch needs an AMD GPU (no CPU fallback)")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.ctx = context or _lib.Context(self.device.index or 0)
        self.lib = self.ctx.lib
        self.xcg = float(xcg) if xcg is not None else P.xcg_of(P.stab_flag if stab_flag is None else stab_flag)
        self.fi_flag = int(fi_flag)
        self.dt = float(dt)
        self.flags = int(flags)
        # host arrays, or tensors already on the device (from_trim: the trim states never leave HBM)
        as2d = lambda a: (a if a.dim() == 2 else a.unsqueeze(0)).to(torch.float64) if isinstance(a, torch.Tensor) \
            else torch.as_tensor(np.atleast_2d(np.asarray(a, dtype=np.float64)))
        x0 = as2d(x0)
        self.B = x0.shape[0]
        assert x0.shape[1] == 18
        u0 = x0[:, 12:16] if u0 is None else as2d(u0)
        self._x_init = self._soa(x0)            # x.initial_condition
        self._u_init = self._soa(u0)            # u.initial_condition
        self._x = self._x_init.clone()          # x.values  [18,B]
        self._u = self._u_init.clone()          # u.values  [4,B]
        self.status = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.ssr = None                         # (Ad,Bd,Cd) per aircraft once linearised (env.py:49-60)

    # ------------------------------------------------------------------ env.py:198-292
    @staticmethod
    def trim(h_t, v_t, *, stab_flag=None, xcg=None, fi_flag=P.fi_flag, device="cuda:0", flags=0, maxiter=50000, context=None):
        """Batched F16.trim: straight-and-level trim at altitudes h_t [B] (ft) and airspeeds v_t [B] (ft/s) by the
        reference's Nelder-Mead, all conditions in one launch.  Returns (x_trim [B,18], info dict)."""
        if not torch.cuda.is_available():
            raise _lib.F16HipError("trim needs an AMD GPU (no CPU fallback)")
        dev = torch.device(device)
        torch.cuda.set_device(dev)
        ctx = context or _lib.Context(dev.index or 0)
        xcg = float(xcg) if xcg is not None else P.xcg_of(P.stab_flag if stab_flag is None else stab_flag)
        h = torch.as_tensor(np.atleast_1d(np.asarray(h_t, dtype=np.float64)), device=dev)
        v = torch.as_tensor(np.atleast_1d(np.asarray(v_t, dtype=np.float64)), device=dev)
        B = h.shape[0]
        assert v.shape[0] == B
        xt = torch.empty((18, B), dtype=torch.float64, device=dev)
        cost = torch.empty(B, dtype=torch.float64, device=dev)
        iters = torch.zeros(B, dtype=torch.int32, device=dev)
        nfev = torch.zeros(B, dtype=torch.int32, device=dev)
        st = torch.zeros(B, dtype=torch.int32, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(ctx.lib.f16_trim_batch(ctx.handle, _vp(h), _vp(v), _vp(xt), _vp(cost), _vp(iters), _vp(nfev), _vp(st), B, B,
                                          xcg, int(fi_flag), int(flags), int(maxiter), None, stream), ctx.lib)
        return xt.t(), dict(cost=cost, iters=iters, nfev=nfev, status=st, context=ctx)

    @classmethod
    def from_trim(cls, h_t, v_t, **kw):
        """env.py:42-44: construct the batch at its trim points (x.initial_condition = trim, u = x[12:16])."""
        tk = {k: kw[k] for k in ("stab_flag", "xcg", "fi_flag", "device", "flags") if k in kw}
        x, info = cls.trim(h_t, v_t, **tk)            # [B,18] view of the device-resident [18,B] result
        env = cls(x, None, context=info["context"], **{k: v for k, v in kw.items() if k != "maxiter"})
        env.trim_info = info
        return env

    # ------------------------------------------------------------------ helpers
    def _soa(self, a, rows=None):
        """[B,k] (or [k], broadcast) host/device array -> contiguous state-major [k,B] fp64 on the GPU."""
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64))
        t = t.to(device=self.device, dtype=torch.float64)
        if t.dim() == 1:
            t = t.unsqueeze(0).expand(self.B, -1)
        if rows is not None:
            assert t.shape[1] == rows, (tuple(t.shape), rows)
        out = t.t().contiguous()
        # (a [k,B] device tensor viewed as [B,k] comes back as the SAME storage: the resident copies must not alias their source)
        return out.clone() if isinstance(a, torch.Tensor) and out.data_ptr() == a.data_ptr() else out

    @property
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        _lib.check(rc, self.lib)

    # x.values / u.values as [B,18] / [B,4] views of the resident state
    @property
    def x_values(self):
        return self._x.t()

    @property
    def u_values(self):
        return self._u.t()

    def set_state(self, x):
        self._x.copy_(self._soa(x, 18))

    def set_input(self, u):
        self._u.copy_(self._soa(u, 4))

    # ------------------------------------------------------------------ env.py:132-150
    def reset(self):
        self._x.copy_(self._x_init)
        self._u.copy_(self._u_init)
        self.status.zero_()
        return self.get_obs(self._x, self._u)

    def get_obs(self, x=None, u=None):
        """Observed states x[_obs_x_idx] (parameters.py:134,160) -> [B,10]."""
        x = self._x if x is None else x
        return x[P.obs_x_idx].t()

    def _get_obs_na(self, x9, u=None):
        return x9

    def debug_table_lookup(self, tid, alpha, beta=None, el=None):
        """One of the reference's 43 table functions (C/hifi_F16_AeroData.c:109-1861) evaluated by the device lookup code
        at the given points (degrees).  Test entry.  Returns (values, status bits)."""
        a = np.ascontiguousarray(alpha, dtype=np.float64)
        b = np.zeros_like(a) if beta is None else np.ascontiguousarray(beta, dtype=np.float64)
        e = np.zeros_like(a) if el is None else np.ascontiguousarray(el, dtype=np.float64)
        out, st = np.zeros_like(a), np.zeros(a.shape, dtype=np.int32)
        hp = lambda v: ctypes.c_void_p(v.ctypes.data)
        self._check(self.lib.f16_debug_table_lookup(self.ctx.handle, int(tid), hp(a), hp(b), hp(e), a.size, hp(out), hp(st)))
        return out, st

    # ------------------------------------------------------------------ env.py:65-103
    def _calc_xdot(self, x=None, u=None):
        """xdot [B,18] for states x [B,18] and inputs u [B,4] (defaults: resident x.values/u.values)."""
        xs = self._x if x is None else self._soa(x, 18)
        us = self._u if u is None else self._soa(u, 4)
        out = torch.empty_like(xs)
        st = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self._check(self.lib.f16_xdot_batch(self.ctx.handle, _vp(xs), _vp(us), _vp(out), _vp(st), self.B, self.B,
                                            self.xcg, self.fi_flag, self.flags, self._stream))
        self.last_status = st
        return out.t()

    def nlplant(self, xu):
        """C/nlplant.c:23 for B aircraft: xu [B,>=17] -> xdot [B,18] (12..17 = nx,ny,nz,mach,qbar,ps)."""
        xs = self._soa(torch.as_tensor(np.asarray(xu, dtype=np.float64))[:, :18] if not isinstance(xu, torch.Tensor) else xu[:, :18])
        out = torch.empty((18, self.B), dtype=torch.float64, device=self.device)
        st = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self._check(self.lib.f16_nlplant_batch(self.ctx.handle, _vp(xs), _vp(out), _vp(st), self.B, self.B, self.xcg,
                                               self.fi_flag, self.flags, self._stream))
        self.last_status = st
        return out.t()

    # ------------------------------------------------------------------ env.py:105-130
    def step(self, action=None, auto_reset=False):
        """One explicit-Euler step of every aircraft (env.py:126).  Returns (obs [B,10], reward, done, info) like
        the reference; where the reference exit()s on an envelope violation (env.py:121-124) the aircraft is frozen
        and `done[b]` is set (status bit F16_ST_ENVELOPE).
        auto_reset (vectorised-env convention): aircraft that are done are put back on their initial
        condition (env.py:132-135 `reset` for those aircraft only) after the step; the returned observation is the
        post-reset one and `info["terminal_observation"]` holds the last pre-reset observation of every aircraft."""
        self.rollout(1, action)
        done = (self.status & _lib.F16_ST["ENVELOPE"]) != 0
        reward = torch.ones(self.B, dtype=torch.float64, device=self.device)
        info = {"fidelity": "high" if self.fi_flag == 1 else "low", "status": self.status.clone() if auto_reset else self.status}
        if auto_reset:
            info["terminal_observation"] = self.get_obs().clone()
            self._x[:, done] = self._x_init[:, done]
            self.status[done] = 0
        return self.get_obs(), reward, done, info

    def rollout(self, nsteps, action=None, traj_every=None):
        """nsteps Euler steps in ONE launch with the state held in registers (the reference's
        `for ...: self.step(u)` loops, test_env.py:456-462).  traj_every=k stores the state after every k-th
        step and returns it as [nsteps//k, 18, B] (state-major)."""
        us = self._u if action is None else self._soa(action, 4)
        traj = None
        if traj_every:
            assert nsteps % traj_every == 0
            traj = torch.empty((nsteps // traj_every, 18, self.B), dtype=torch.float64, device=self.device)
        self._check(self.lib.f16_rollout(self.ctx.handle, _vp(self._x), _vp(us), _vp(traj), _vp(self.status), self.B,
                                         self.B, int(nsteps), int(traj_every or 1), self.dt, self.xcg, self.fi_flag,
                                         self.flags, self._stream))
        return traj

    # ------------------------------------------------------------------ env.py:152-193
    def _get_mpc_x(self):
        return self._x[P.mpc_x_idx].t()

    def _get_mpc_u(self):
        return self._u[P.mpc_u_idx].t()

    def _get_mpc_act_states(self):
        return self._x[P.mpc_u_states_idx].t()

    def _calc_xdot_na(self, x9, u3):
        """x9 [B,9] MPC states, u3 [B,3] actuator positions -> xdot9 [B,9] (other states from x.values)."""
        x9s, u3s = self._soa(x9, 9), self._soa(u3, 3)
        out = torch.empty_like(x9s)
        st = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self._check(self.lib.f16_xdot_na_batch(self.ctx.handle, _vp(self._x), _vp(x9s), _vp(u3s), _vp(out), _vp(st),
                                               self.B, self.B, self.xcg, self.fi_flag, self.flags, self._stream))
        self.last_status = st
        return out.t()

    # ------------------------------------------------------------------ env.py:294-358
    def linearise(self, x=None, u=None, _calc_xdot=None, get_obs=None, eps=1e-5):
        """env.py:294 `linearise(self, x, u, _calc_xdot=None, get_obs=None)`: forward-difference linearisation (eps 1e-5,
        env.py:319) of every aircraft at its own point, with the reference's dispatch on the model function:
          * default / `self._calc_xdot` + `self.get_obs`: the 18-state model, x [B,18], u [B,4]
            -> A [B,18,18], B [B,18,4], C [B,10,18], D [B,10,4]                                        (env.py:45)
          * `self._calc_xdot_na` + `self._get_obs_na`: the reduced model, x = x9 [B,9] (self._get_mpc_x()), u = u3 [B,3]
            (self._get_mpc_u()); the other states come from x.values as in env.py:172-177
            -> A [B,9,9], B [B,9,3], C [B,9,9], D [B,9,3]                                              (env.py:49)
        x / u default to the resident values.  Only these two models exist on the device; another callable raises."""
        if _calc_xdot is None or _calc_xdot == self._calc_xdot:
            if get_obs is not None and get_obs != self.get_obs:
                raise ValueError("the 18-state model is linearised with get_obs (env.py:312-313)")
            r = self.linearise_full(eps, discretise=False, x=x, u=u)
            return r["Ac"], r["Bc"], r["Cc"], r["Dc"]
        if _calc_xdot != self._calc_xdot_na:
            raise ValueError("linearise knows the reference's two models: _calc_xdot and _calc_xdot_na")
        if get_obs is not None and get_obs != self._get_obs_na:
            raise ValueError("the reduced model is linearised with _get_obs_na (env.py:49)")
        xs, us = self._x, self._u
        if x is not None or u is not None:          # scatter the given x9 / u3 over a copy of the resident state (env.py:172-177)
            xs, us = self._x.clone(), self._u.clone()
            if x is not None:
                xs[P.mpc_x_idx] = self._soa(x, 9)
            if u is not None:
                us[P.mpc_u_idx] = self._soa(u, 3)
        return self._linearise_na(eps, xs, us)

    def _linearise_na(self, eps=1e-5, xs=None, us=None):
        """The reduced (9-state) linearisation at the resident point (or at state-major xs [18,B] / us [4,B])."""
        xs = self._x if xs is None else xs
        us = self._u if us is None else us
        Ac = torch.empty((81, self.B), dtype=torch.float64, device=self.device)
        Bc = torch.empty((27, self.B), dtype=torch.float64, device=self.device)
        Cc = torch.empty((81, self.B), dtype=torch.float64, device=self.device)
        st = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self._check(self.lib.f16_linearise_batch(self.ctx.handle, _vp(xs), _vp(us), _vp(Ac), _vp(Bc), _vp(Cc), _vp(st),
                                                 self.B, self.B, eps, self.xcg, self.fi_flag, self.flags, self._stream))
        self.last_status = st
        self._lin = (Ac, Bc, Cc)
        return (Ac.t().reshape(self.B, 9, 9), Bc.t().reshape(self.B, 9, 3), Cc.t().reshape(self.B, 9, 9),
                torch.zeros((self.B, 9, 3), dtype=torch.float64, device=self.device))

    def linearise_full(self, eps=1e-5, discretise=True, x=None, u=None):
        """env.py:45-46: 18-state linearisation (default _calc_xdot/get_obs) at every aircraft's own (x, u) and its
        zero-order-hold discretisation.  Returns dict(Ac [B,18,18], Bc [B,18,4], Cc [B,10,18], Dc, Ad, Bd)."""
        B = self.B
        xs = self._x if x is None else self._soa(x, 18)
        us = self._u if u is None else self._soa(u, 4)
        Ac = torch.empty((324, B), dtype=torch.float64, device=self.device)
        Bc = torch.empty((72, B), dtype=torch.float64, device=self.device)
        Cc = torch.empty((180, B), dtype=torch.float64, device=self.device)
        st = torch.zeros(B, dtype=torch.int32, device=self.device)
        self._check(self.lib.f16_linearise_full_batch(self.ctx.handle, _vp(xs), _vp(us), _vp(Ac), _vp(Bc), _vp(Cc),
                                                      _vp(st), B, B, eps, self.xcg, self.fi_flag, self.flags, self._stream))
        self.last_status = st
        out = dict(Ac=Ac.t().reshape(B, 18, 18), Bc=Bc.t().reshape(B, 18, 4), Cc=Cc.t().reshape(B, 10, 18),
                   Dc=torch.zeros((B, 10, 4), dtype=torch.float64, device=self.device))
        if discretise:
            Ad, Bd = torch.empty_like(Ac), torch.empty_like(Bc)
            self._check(self.lib.f16_c2d_full_batch(self.ctx.handle, _vp(Ac), _vp(Bc), _vp(Ad), _vp(Bd), B, B, self.dt,
                                                    self._stream))
            out["Ad"], out["Bd"] = Ad.t().reshape(B, 18, 18), Bd.t().reshape(B, 18, 4)
        return out

    def discretise(self, Ac=None, Bc=None):
        """scipy.signal.cont2discrete(..., dt) zero-order hold (env.py:50,351) per aircraft."""
        if Ac is None:
            Ac, Bc, _ = self._lin
        else:
            Ac = Ac.reshape(self.B, 81).t().contiguous()
            Bc = Bc.reshape(self.B, 27).t().contiguous()
        Ad, Bd = torch.empty_like(Ac), torch.empty_like(Bc)
        self._check(self.lib.f16_c2d_batch(self.ctx.handle, _vp(Ac), _vp(Bc), _vp(Ad), _vp(Bd), self.B, self.B, self.dt,
                                           self._stream))
        return Ad, Bd

    def build_ssr(self, eps=1e-5):
        """env.py:49-60: reduced discrete model per aircraft, frozen until called again."""
        self.release_MPC_plan()                 # a new model invalidates a prepared plan
        self._linearise_na(eps)
        Ad, Bd = self.discretise()
        self.ssr = (Ad, Bd, self._lin[2])
        return self.ssr

    def _calc_LQR_gain(self, Q=None, R=None):
        """env.py:344-358: linearise -> ZOH -> K = -dlqr(Ad,Bd,Cd'Cd,I).  Returns K [B,3,9].
        Q [9,9] / R [3,3]: the weights utils.py:219 `dlqr(A, B, Q, R)` takes (default: env.py's Q = Cd'Cd, R = I)."""
        Ad, Bd, Cd = self.build_ssr()
        K = torch.empty((27, self.B), dtype=torch.float64, device=self.device)
        st = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        w = _lib.make_weights(Q=Q, R=R)
        self._check(self.lib.f16_lqr_batch_w(self.ctx.handle, _vp(Ad), _vp(Bd), _vp(Cd), ctypes.byref(w) if w else None, _vp(K), None,
                                             _vp(st), self.B, self.B, self._stream))
        self.last_status = st
        return K.t().reshape(self.B, 3, 9)

    def _calc_LQR_action(self, p_dem, q_dem, r_dem, K, x, u0):
        """env.py:360-371: u = -K (x_ref - x) + u0 with x_ref[4:7] = demands.  x [B,9], u0 [B,3]."""
        x = torch.as_tensor(x, device=self.device, dtype=torch.float64)
        u0 = torch.as_tensor(u0, device=self.device, dtype=torch.float64)
        x_ref = x.clone()
        x_ref[:, 4], x_ref[:, 5], x_ref[:, 6] = p_dem, q_dem, r_dem
        return -(K @ (x_ref - x).unsqueeze(-1)).squeeze(-1) + u0

    def _demands(self, p_dem, q_dem, r_dem):
        """(p, q, r) demands, scalars or [B], as a state-major [3,B] device tensor."""
        dem = torch.empty((3, self.B), dtype=torch.float64, device=self.device)
        for k, v in enumerate((p_dem, q_dem, r_dem)):
            if isinstance(v, (int, float)):
                dem[k].fill_(float(v))                # scalar demand: a fill kernel, no host-to-device copy
            else:
                dem[k] = torch.as_tensor(v, dtype=torch.float64, device=self.device)
        return dem

    def rollout_LQR(self, nsteps, p_dem, q_dem, r_dem, K=None, u0=None, traj_every=None, linear=False):
        """The reference's LQR loops (test_env_mk2.py:25-88 `LQR(linear=...)`; flight_sim.py:139,181) as ONE launch.
        linear=False (test_env_mk2.py:70-85): per step `u = _calc_LQR_action(p_dem, q_dem, r_dem, K, x._get_mpc_x(),
        u.initial_condition[1:])`, `u.values[1:] = u`, `step(u.values)`, the state in registers for all nsteps.  K [B,3,9] defaults
        to `_calc_LQR_gain()` at the current point; u0 [B,4] defaults to the CURRENT thrust command u.values[0] (the loop never
        writes it) with u.initial_condition[1:] as the LQR offset; demands scalars or [B].  u.values ends up holding the last
        action, as in the reference.  traj_every=k returns the states after every k-th step, [nsteps//k, 18, B].
        linear=True (test_env_mk2.py:46-62, what main.py:35 runs): the same law on the frozen reduced model,
        `x = ssr.Ad @ x + ssr.Bd @ u` from x = x._get_mpc_x(); x.values / u.values are not touched (the reference's loop works on
        locals).  Returns (x_storage [nsteps//k, 9, B], u_storage [nsteps//k, 3, B]) with k = traj_every or 1."""
        if K is None:
            K = self._calc_LQR_gain()
        Ks = torch.as_tensor(K, device=self.device, dtype=torch.float64).reshape(self.B, 27).t().contiguous()
        dem = self._demands(p_dem, q_dem, r_dem)
        if linear:
            if self.ssr is None:
                self.build_ssr()
            Ad, Bd, _ = self.ssr
            xref = torch.zeros((9, self.B), dtype=torch.float64, device=self.device)
            xref[4:7] = dem                                                  # env.py:365-367
            u03 = self._u_init[1:4].contiguous() if u0 is None else self._soa(u0, 4)[1:4].contiguous()
            return self.rollout_linear(self._x[P.mpc_x_idx], Ad, Bd, Ks, xref, u03, nsteps, track=(4, 5, 6), traj_every=traj_every,
                                       state_major=True)
        if u0 is None:       # test_env_mk2.py:76-82: the offset is u.initial_condition[1:], the thrust the CURRENT u.values[0]
            u0s = torch.cat((self._u[0:1], self._u_init[1:4]), 0).contiguous()
        else:
            u0s = self._soa(u0, 4)
        traj = None
        if traj_every:
            assert nsteps % traj_every == 0
            traj = torch.empty((nsteps // traj_every, 18, self.B), dtype=torch.float64, device=self.device)
        self._check(self.lib.f16_rollout_lqr(self.ctx.handle, _vp(self._x), _vp(u0s), _vp(Ks), _vp(dem), _vp(traj), _vp(self._u),
                                             _vp(self.status), self.B, self.B, int(nsteps), int(traj_every or 1), self.dt,
                                             self.xcg, self.fi_flag, self.flags, self._stream))
        return traj

    def rollout_linear(self, x9, Ad, Bd, K, x_ref, u0, nsteps, track=None, traj_every=None, state_major=False):
        """`u = -K (x_ref - x) + u0; x = Ad x + Bd u` for nsteps on per-aircraft 9-state / 3-input linear models (C-ABI
        f16_rollout_lqr_linear): the loop of test_env_mk2.py:54-62 (track = (4, 5, 6): the reference follows the current state
        except the three rate demands, env.py:360-371; K = -dlqr as `_calc_LQR_gain` returns it) and of test_env.py:553-559
        (track=None: a fixed full reference; pass K = -dlqr(A, B, Q, R)).  x9 [B,9], Ad [B,9,9], Bd [B,9,3], K [B,3,9], x_ref [B,9],
        u0 [B,3] or None -- or, with state_major=True, the device layout itself ([9,B], [81,B], [27,B], [27,B], [9,B], [3,B]: what
        build_ssr and the gain kernel hold).  Returns (x_storage [nsteps//k, 9, B], u_storage [nsteps//k, 3, B]), k = traj_every or 1;
        the final state is kept in `last_linear_state` [B,9]."""
        def sm(a, rows):                      # -> contiguous state-major [rows, B]
            t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float64))
            t = t.to(device=self.device, dtype=torch.float64)
            if state_major:
                assert tuple(t.shape) == (rows, self.B), (tuple(t.shape), rows)
                return t.contiguous()
            return t.reshape(self.B, rows).t().contiguous()
        xs = sm(x9, 9).clone()
        Ads, Bds, Ks, xr = sm(Ad, 81), sm(Bd, 27), sm(K, 27), sm(x_ref, 9)
        u0s = sm(u0, 3) if u0 is not None else None
        k = int(traj_every or 1)
        assert nsteps % k == 0
        trx = torch.empty((nsteps // k, 9, self.B), dtype=torch.float64, device=self.device)
        tru = torch.empty((nsteps // k, 3, self.B), dtype=torch.float64, device=self.device)
        mask = 0x1FF if track is None else sum(1 << int(j) for j in track)
        self._check(self.lib.f16_rollout_lqr_linear(self.ctx.handle, _vp(xs), _vp(Ads), _vp(Bds), _vp(Ks), _vp(xr), _vp(u0s), _vp(trx),
                                                    _vp(tru), self.B, self.B, int(nsteps), k, mask, self._stream))
        self.last_linear_state = xs.t()
        return trx, tru

    # ------------------------------------------------------------------ env.py:373-424
    def prepare_MPC(self, hzn, settings=None, warm_start=False, weights=None):
        """Prepare the model-only part of calc_MPC_action for horizon hzn from the frozen reduced model self.ssr
        (env.py:49-60 freezes it; the reference still rebuilds the QP on every call): DARE, terminal weight, prediction
        blocks, P, A'A, start rho and the KKT factorisation stay on the device.  `_calc_MPC_action(..., use_plan=True)`
        then only forms the state-dependent vectors and iterates; results are bit-identical to the one-shot call."""
        if self.ssr is None:
            self.build_ssr()
        self.release_MPC_plan()
        Ad, Bd, Cd = self.ssr
        s = _lib.QPSettings()
        self.lib.f16_qp_default_settings(ctypes.byref(s))
        for k, v in (settings or {}).items():
            setattr(s, k, v)
        h = ctypes.c_void_p()
        w = _lib.make_weights(**weights) if weights else None           # (utils.py:21 Q, R and the six bound vectors; fixed for the plan)
        self._check(self.lib.f16_mpc_plan_create_w(self.ctx.handle, ctypes.byref(h), _vp(Ad), _vp(Bd), _vp(Cd),
                                                   ctypes.byref(w) if w else None, self.B, self.B, int(hzn), self.dt, ctypes.byref(s),
                                                   self._stream))
        self._plan, self._plan_hzn = h, int(hzn)
        self._plan_default_settings = int((settings or {}).get("scaling", 10)) > 0       # (what f16_rollout_mpc takes: equilibrated solves)
        if warm_start:      # OSQP's in-object default; the reference starts cold on every call (new object), so: opt-in
            self._check(self.lib.f16_mpc_plan_warm_start(h, 1))
        return self

    def release_MPC_plan(self):
        if getattr(self, "_plan", None) is not None:
            self.lib.f16_mpc_plan_destroy(self._plan)
        self._plan, self._plan_hzn = None, None

    def __del__(self):
        try:
            self.release_MPC_plan()
        except Exception:
            pass

    @staticmethod
    def solver_modes():
        """Named QP-solver settings (overrides of f16_qp_default_settings) for benchmarks and tests."""
        return {"osqp_defaults": None,                                        # what env.py:420-422 invokes (the library default)
                "osqp_defaults_rho_every_25": dict(rho_every=25),             # OSQP's wall-clock interval typically lands here
                "builder_rule": dict(scaling=0, rho=0.0),                     # no equilibration, rho0 = 2 sqrt(tr P / tr A'A)
                "rho_0p1_unscaled": dict(scaling=0, rho=0.1)}

    def setup_OSQP(self, p_dem, q_dem, r_dem, hzn, b=0, weights=None, x_ref=None):
        """The QP of aircraft b in the reference's own form (utils.py:21-167 `setup_OSQP`): dense host arrays
        (P [n,n], q [n], A [15 hzn, n], l, u) with n = 3 hzn and the reference's row order (9 hzn state rows, 3 hzn
        command rows, 3 hzn rate rows; unbounded rows carry +-inf), built on the device from the frozen reduced
        model and the current state -- for callers that hand the QP to a solver of their own, and for the tests.
        weights: dict(Q=, R=, x_lb=, x_ub=, u_lb=, u_ub=, udot_lb=, udot_ub=) -- the arguments of utils.py:21 that env.py fills
        with constants (any subset; any bound pattern); x_ref [B,9] (or [9]): the reference itself instead of "x with
        x[5:8] = demands" (env.py:380-383)."""
        if self.ssr is None:
            self.build_ssr()
        Ad, Bd, Cd = self.ssr
        dem = torch.empty((3, self.B), dtype=torch.float64, device=self.device)
        for k, v in enumerate((p_dem, q_dem, r_dem)):
            dem[k] = torch.as_tensor(v, dtype=torch.float64, device=self.device)
        n, rows = 3 * int(hzn), 15 * int(hzn)
        P, q, A = np.zeros((n, n)), np.zeros(n), np.zeros((rows, n))
        l, u = np.zeros(rows), np.zeros(rows)
        hp = lambda a: ctypes.c_void_p(a.ctypes.data)
        w = _lib.make_weights(**weights) if weights else None
        xr = self._soa(x_ref, 9) if x_ref is not None else None
        self._check(self.lib.f16_mpc_qp_debug_w(self.ctx.handle, _vp(Ad), _vp(Bd), _vp(Cd), _vp(self._x), _vp(dem), _vp(xr),
                                                ctypes.byref(w) if w else None, int(b), self.B, int(hzn), self.dt, hp(P), hp(q), hp(A),
                                                hp(l), hp(u)))
        return P, q, A, l, u

    def _calc_MPC_action(self, p_dem, q_dem, r_dem, hzn, settings=None, return_info=False, relinearise=False,
                         use_plan=False, weights=None, x_ref=None):
        """First MPC move [B,3] (dh,da,dr commands) for demands p,q,r (scalars or [B]) over horizon hzn, from the
        frozen reduced model self.ssr (env.py:385-387) and the current state.  The QP of utils.py:21-167 is solved
        on the GPU by OSQP-style ADMM (the reference calls the `osqp` package, env.py:420-422).
        weights / x_ref: as in setup_OSQP (the solvers keep the reference's pattern of bounded rows; a plan fixes its weights at
        prepare_MPC, x_ref is per call)."""
        # relinearise=True: SURVEY.md 8f-2 -- the reduced model is re-derived at the CURRENT state on every call (the
        # reference freezes it at construction, env.py:49-60; its test_env.py:625-687 loops re-linearise per step)
        if self.ssr is None or relinearise:
            self.build_ssr()
        Ad, Bd, Cd = self.ssr
        if torch.is_tensor(p_dem) and p_dem.dim() == 2:       # demands already on the device as [3,B] (graph capture)
            dem = p_dem
        else:
            dem = torch.empty((3, self.B), dtype=torch.float64, device=self.device)
            for k, v in enumerate((p_dem, q_dem, r_dem)):
                if isinstance(v, (int, float)):
                    dem[k].fill_(float(v))            # scalar demand: a fill kernel, no host-to-device copy
                else:
                    dem[k] = torch.as_tensor(v, dtype=torch.float64, device=self.device)
        ucmd = torch.empty((3, self.B), dtype=torch.float64, device=self.device)
        info = torch.empty((4, self.B), dtype=torch.float64, device=self.device)
        useq = torch.empty((3 * hzn, self.B), dtype=torch.float64, device=self.device) if return_info else None
        st = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        if use_plan:
            if relinearise or settings or weights:
                raise ValueError("a prepared plan fixes the model, the QP settings and the weights (prepare_MPC)")
            if getattr(self, "_plan", None) is None or self._plan_hzn != int(hzn):
                self.prepare_MPC(hzn)
            xr = self._soa(x_ref, 9) if x_ref is not None else None
            self._check(self.lib.f16_mpc_plan_solve_w(self._plan, _vp(self._x), _vp(dem), _vp(xr), _vp(ucmd), _vp(useq), _vp(info),
                                                      _vp(st), self._stream))
            self.last_status, self.last_iters = st, info[0]
            self.status |= st          # sticky: F16_ST_QP_INFEASIBLE / QP_MAXITER of ANY step stay visible after a closed loop
            if return_info:
                return ucmd.t(), dict(iters=info[0], r_prim=info[1], r_dual=info[2], rho=info[3], u_seq=useq.t(), status=st)
            return ucmd.t()
        s = _lib.QPSettings()
        self.lib.f16_qp_default_settings(ctypes.byref(s))
        for k, v in (settings or {}).items():
            setattr(s, k, v)
        w = _lib.make_weights(**weights) if weights else None
        xr = self._soa(x_ref, 9) if x_ref is not None else None
        self._check(self.lib.f16_mpc_batch_w(self.ctx.handle, _vp(Ad), _vp(Bd), _vp(Cd), _vp(self._x), _vp(dem), _vp(xr),
                                             ctypes.byref(w) if w else None, _vp(ucmd), _vp(useq), _vp(info), _vp(st), self.B, self.B,
                                             int(hzn), self.dt, ctypes.byref(s), self._stream))
        self.last_status, self.last_iters = st, info[0]
        self.status |= st
        if return_info:
            return ucmd.t(), dict(iters=info[0], r_prim=info[1], r_dual=info[2], rho=info[3], u_seq=useq.t(), status=st)
        return ucmd.t()

    calc_MPC_action = _calc_MPC_action

    def rollout_MPC(self, nsteps, p_dem, q_dem, r_dem, hzn, traj_every=None, return_info=False, hold_command=False):
        """The reference's closed MPC loop (test_env.py:480-495; BASELINE config 5) as ONE launch (C-ABI f16_rollout_mpc): per step
        `cmd = _calc_MPC_action(p_dem, q_dem, r_dem, hzn); u.values[1:] = cmd; step(u.values)` from the frozen reduced model
        (env.py:49-60) with OSQP's default settings (every solve cold, as the reference's -- or warm from the step before when the plan was
        prepared with warm_start=True).  Work items are (step, aircraft) pairs taken from one queue by one wavefront per
        SIMD, so no step waits for another aircraft's solve -- the host loop `dist.closed_loop_mpc_rollout` (the checker of this
        call) joins the batch after every solve.  Uses the prepared plan of horizon hzn (prepare_MPC; made here if absent).
        Returns the states after every traj_every-th step [nsteps//k, 18, B] (None without traj_every); with return_info also
        dict(cmd [nsteps, 3, B]: what calc_MPC_action returned per step, iters [nsteps, B]).  x.values / u.values / status are
        updated in place; u.values ends up holding the last command.  hold_command: F16_FLAG_HOLD_COMMAND."""
        if getattr(self, "_plan", None) is None or self._plan_hzn != int(hzn):
            self.prepare_MPC(hzn)
        dem = p_dem if (torch.is_tensor(p_dem) and p_dem.dim() == 2) else self._demands(p_dem, q_dem, r_dem)
        traj = None
        if traj_every:
            assert nsteps % traj_every == 0
            traj = torch.empty((nsteps // traj_every, 18, self.B), dtype=torch.float64, device=self.device)
        cmd = torch.empty((nsteps, 3, self.B), dtype=torch.float64, device=self.device) if return_info else None
        its = torch.empty((nsteps, self.B), dtype=torch.int32, device=self.device) if return_info else None
        flags = self.flags | (_lib.F16_FLAG_HOLD_COMMAND if hold_command else 0)
        self._check(self.lib.f16_rollout_mpc(self._plan, _vp(self._x), _vp(self._u), _vp(dem), _vp(traj), _vp(cmd), _vp(its),
                                             _vp(self.status), int(nsteps), int(traj_every or 1), self.xcg, self.fi_flag, flags,
                                             self._stream))
        if return_info:
            return traj, dict(cmd=cmd, iters=its)
        return traj

    def _calc_constr_checking_hzn(self, max_hzn=150, settings=None, return_info=False):
        """env.py:426-436: the first move of calc_MPC_action(0, 0, 0, N) for every horizon N = 1..max_hzn (the reference
        fills u[:, N-1] for one aircraft; here [B, 3, max_hzn]).  One library call (f16_mpc_hzn_sweep): the long horizons
        are solved by a single launch over every (horizon, aircraft) pair; each slice equals _calc_MPC_action(0, 0, 0, N)."""
        if self.ssr is None:
            self.build_ssr()
        Ad, Bd, Cd = self.ssr
        dem = torch.zeros((3, self.B), dtype=torch.float64, device=self.device)
        ucmd = torch.empty((max_hzn, 3, self.B), dtype=torch.float64, device=self.device)
        info = torch.empty((max_hzn, 4, self.B), dtype=torch.float64, device=self.device)
        st = torch.zeros((max_hzn, self.B), dtype=torch.int32, device=self.device)
        s = _lib.QPSettings()
        self.lib.f16_qp_default_settings(ctypes.byref(s))
        for k, v in (settings or {}).items():
            setattr(s, k, v)
        self._check(self.lib.f16_mpc_hzn_sweep(self.ctx.handle, _vp(Ad), _vp(Bd), _vp(Cd), _vp(self._x), _vp(dem), _vp(ucmd),
                                               _vp(info), _vp(st), self.B, self.B, 1, int(max_hzn), self.dt, ctypes.byref(s),
                                               self._stream))
        self.last_status = st
        out = ucmd.permute(2, 1, 0)
        if return_info:
            return out, dict(iters=info[:, 0], r_prim=info[:, 1], r_dual=info[:, 2], rho=info[:, 3], status=st)
        return out
