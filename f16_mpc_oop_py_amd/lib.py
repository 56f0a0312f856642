"""ctypes binding of libf16hip.so (the C-ABI declared in include/f16_hip.h).

This is the same mechanism the reference uses for its plant: `ctypes.CDLL(<.so>)`
(parameters.py:108-114).  There is no CPU fallback: if the HIP library is missing or no GPU
is visible every entry point raises.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.environ.get("F16HIP_SO", os.path.join(HERE, "libf16hip.so"))   # override only for A/B experiments
SOURCES = ["f16_api.hip", "f16_dynamics.hip", "f16_control.hip", "f16_mpc_solve.hip", "f16_mpc_wave.hip", "f16_mpc_big.hip", "f16_trim.hip", "f16_debug.hip", "f16_tables.cpp"]
# the expression-exact (F16_STRICT) build of the plant alone: checker-side evidence that the device lookups and the plant
# reproduce the reference bit for bit where no libm call is involved (tests/test_gpu_dynamics.py); never used by the product
STRICT_SO_PATH = os.path.join(HERE, "libf16hip_strict.so")
STRICT_SOURCES = ["f16_api.hip", "f16_dynamics.hip", "f16_debug.hip", "f16_tables.cpp"]
# Default build: FMA contraction on, tan = sin/cos, tfac^4.14 = tfac^4 * exp(0.14 log tfac), branch-free sincos (each <= 2 ulp away from
# the strict form; measured: xdot max rel. error vs the CPU restatement unchanged at 5e-14, -16 % kernel time).
# F16_STRICT=1 builds the expression-by-expression variant (no contraction, libm tan/pow): 97 % of xdot outputs
# then agree with the reference restatement bit for bit.
if os.environ.get("F16_STRICT"):
    _NUMERICS = ["-ffp-contract=off"]
else:
    _NUMERICS = ["-ffp-contract=fast", "-DF16_FAST_TAN", "-DF16_FAST_POW", "-DF16_FAST_TRIG", "-DF16_FAST_DIV"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result"] + _NUMERICS

F16_ST = dict(ALPHA1=1, ALPHA2=2, BETA=4, EL=8, ENVELOPE=16, NONFINITE=32, QP_MAXITER=64, QP_INFEASIBLE=128)
F16_ST_ENV_STATE = lambda k: 1 << (8 + k)      # with ENVELOPE: state k was outside its box (env.py:117-124)
F16_ST_LOOP_STALL = 1 << 26                    # f16_rollout_mpc gave up waiting for the aircraft's previous step (a guard; never observed)
F16_FLAG_FIX_CLR = 1
F16_FLAG_NO_ENVELOPE = 2
F16_FLAG_ONE_LANE = 4          # rollouts: the one-lane-per-aircraft kernel whatever the batch size (results independent of B)
F16_FLAG_HOLD_COMMAND = 8      # closed MPC loops: a step without a command (infeasible QP, state not finite) keeps the previous one


class F16HipError(RuntimeError):
    pass


class QPSettings(ctypes.Structure):
    _fields_ = [("rho", ctypes.c_double), ("sigma", ctypes.c_double), ("alpha", ctypes.c_double),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double), ("eps_prim_inf", ctypes.c_double),
                ("max_iter", ctypes.c_int),
                ("check_every", ctypes.c_int), ("rho_every", ctypes.c_int), ("adaptive_rho", ctypes.c_int),
                ("scaling", ctypes.c_int)]


class MPCWeights(ctypes.Structure):
    """include/f16_hip.h `f16_mpc_weights`: the arguments of utils.py:21 setup_OSQP / utils.py:219 dlqr that env.py fills with
    constants."""
    _fields_ = [("q_from_cd", ctypes.c_int), ("Q", ctypes.c_double * 81), ("R", ctypes.c_double * 9),
                ("x_lb", ctypes.c_double * 9), ("x_ub", ctypes.c_double * 9), ("u_lb", ctypes.c_double * 3), ("u_ub", ctypes.c_double * 3),
                ("udot_lb", ctypes.c_double * 3), ("udot_ub", ctypes.c_double * 3)]


def make_weights(Q=None, R=None, x_lb=None, x_ub=None, u_lb=None, u_ub=None, udot_lb=None, udot_ub=None):
    """MPCWeights from env.py's constants with the given entries replaced (None = all defaults -> None: the plain entry points)."""
    if all(v is None for v in (Q, R, x_lb, x_ub, u_lb, u_ub, udot_lb, udot_ub)):
        return None
    import numpy as np
    w = MPCWeights()
    load().f16_mpc_default_weights(ctypes.byref(w))
    if Q is not None:
        w.q_from_cd = 0
        w.Q[:] = list(np.asarray(Q, dtype=np.float64).reshape(81))
    if R is not None:
        w.R[:] = list(np.asarray(R, dtype=np.float64).reshape(9))
    for name, v, n in (("x_lb", x_lb, 9), ("x_ub", x_ub, 9), ("u_lb", u_lb, 3), ("u_ub", u_ub, 3), ("udot_lb", udot_lb, 3), ("udot_ub", udot_ub, 3)):
        if v is not None:
            getattr(w, name)[:] = list(np.asarray(v, dtype=np.float64).reshape(n))
    return w


def _fingerprint():
    """Hash of everything the binary depends on: every source / header under csrc/, the public header, the compiler
    flags (F16_STRICT, F16_HIPCC_EXTRA included).  Stored beside the .so at build time; load() refuses a stale binary."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", ".hpp", ".inc")))
    for f in files + [os.path.join("..", "..", "include", "f16_hip.h")]:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(HIPCC_FLAGS + os.environ.get("F16_HIPCC_EXTRA", "").split()).encode())
    return h.hexdigest()


def _compile_and_link(srcs, flags, out, verbose=False, stamp=None, fp=None, force=False):
    """One object per source, compiled in parallel and cached under build/obj/ by a hash of (source, every header, flags);
    then one link.  A change to one kernel file recompiles that file only.  One builder at a time per tree (file lock);
    the stamp is re-checked after the lock is taken and written inside it, so that the ranks of a multi-GPU run that all
    find a stale library do not relink in turn."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    import fcntl
    lock = open(os.path.join(objdir, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and stamp and fp and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == fp:
            return                      # another process built it while this one waited for the lock (force: relink regardless)
        _compile_and_link_locked(srcs, flags, out, verbose, hipcc, objdir)
        if stamp and fp:
            with open(stamp + ".tmp", "w") as f:
                f.write(fp + "\n")
            os.replace(stamp + ".tmp", stamp)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _compile_and_link_locked(srcs, flags, out, verbose, hipcc, objdir):
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hdr = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".hpp", ".inc")):
            hdr.update(open(os.path.join(CSRC, f), "rb").read())
    hdr.update(open(os.path.join(HERE, "..", "include", "f16_hip.h"), "rb").read())
    cflags = [f for f in flags if f != "-shared"]
    hdr.update(" ".join(cflags).encode())

    def one(src):
        h = hashlib.sha256(hdr.digest() + open(src, "rb").read()).hexdigest()[:24]
        obj = os.path.join(objdir, os.path.basename(src) + "." + h + ".o")
        if not os.path.exists(obj):
            cmd = [hipcc] + cflags + ["-c", "-o", obj + ".tmp", src]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into the in-tree libf16hip.so (hipcc cross-compiles without a GPU).  Rebuilds
    when the recorded fingerprint (sources + flags) differs from the tree's."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    fp = _fingerprint()
    stamp = SO_PATH + ".stamp"
    if not force and os.path.exists(SO_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == fp:
        return SO_PATH
    _compile_and_link(srcs, HIPCC_FLAGS + os.environ.get("F16_HIPCC_EXTRA", "").split(), SO_PATH, verbose, stamp, fp, force)
    return SO_PATH


def build_strict(force=False):
    """The F16_STRICT variant of the plant sources (no FMA contraction, IEEE divisions, libm tan/pow) as a second,
    test-only library."""
    import hashlib
    srcs = [os.path.join(CSRC, s) for s in STRICT_SOURCES]
    h = hashlib.sha256(_fingerprint().encode() + b"strict").hexdigest()
    stamp = STRICT_SO_PATH + ".stamp"
    if not force and os.path.exists(STRICT_SO_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == h:
        return STRICT_SO_PATH
    flags = [f for f in HIPCC_FLAGS if f not in _NUMERICS] + ["-ffp-contract=off"]
    _compile_and_link(srcs, flags, STRICT_SO_PATH, False, stamp, h, force)
    return STRICT_SO_PATH


def strict_path():
    """Path of the up-to-date strict library, or an error naming the build command -- for callers that must not compile
    (GPU test processes: see load())."""
    import hashlib
    h = hashlib.sha256(_fingerprint().encode() + b"strict").hexdigest()
    stamp = STRICT_SO_PATH + ".stamp"
    if os.path.exists(STRICT_SO_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == h:
        return STRICT_SO_PATH
    raise F16HipError(f"{STRICT_SO_PATH} is missing or stale: run `python -c 'import __graft_entry__ as g; g.build()'`")


_LIB = None


def load():
    """Load libf16hip.so once.  torch is imported first so that both share ONE HIP runtime
    (same libamdhip64 SONAME): device pointers from torch tensors are valid in our kernels."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  (must precede CDLL, see docstring)
    # load() never compiles: it may be called from a process that has already initialised the GPU (bench.run after
    # torch.cuda.set_device, or anything under `rocprofv3 -- python3 ...`, whose preload is inherited by children), and
    # the hipcc -> clang -> lld chain would be an exec hop of such a process.  A stale or missing binary is an error that
    # names the build command; F16_AUTOBUILD=1 opts back in for interactive use, and is ignored under a profiler preload.
    stamp = SO_PATH + ".stamp"
    build_cmd = "python -c 'import __graft_entry__ as g; g.build()'"
    stale = not os.path.exists(SO_PATH) or not os.path.exists(stamp) or open(stamp).read().strip() != _fingerprint()
    if stale and "F16HIP_SO" not in os.environ:
        profiled = bool(os.environ.get("LD_PRELOAD")) or any(k.startswith(("ROCP_", "ROCPROFILER_")) for k in os.environ)
        if os.environ.get("F16_AUTOBUILD") == "1" and not profiled:
            build()
        else:
            raise F16HipError(f"{SO_PATH} is missing or older than csrc/ (or built with other flags): run `{build_cmd}` "
                              f"first (load() does not compile; F16_AUTOBUILD=1 opts in outside a profiler)")
    if not os.path.exists(SO_PATH):
        raise F16HipError(f"{SO_PATH} is missing: run `{build_cmd}`")
    L = ctypes.CDLL(SO_PATH)
    vp, d, i, l, u = ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_long, ctypes.c_uint
    L.f16_create.argtypes = [ctypes.POINTER(vp), i]
    L.f16_destroy.argtypes = [vp]
    L.f16_last_error.restype = ctypes.c_char_p
    L.f16_table_image_doubles.restype = ctypes.c_size_t
    L.f16_debug_read_tables.argtypes = [vp, vp]
    L.f16_table_image_i32_ints.restype = ctypes.c_size_t
    L.f16_debug_read_tables_i32.argtypes = [vp, vp]
    L.Nlplant.argtypes = [vp, vp, i]
    L.Nlplant.restype = None
    L.atmos.argtypes = [d, d, vp]
    L.atmos.restype = None
    L.f16_dropin_config.argtypes = [d, u]
    L.f16_dropin_config.restype = None
    L.f16_xdot_batch.argtypes = [vp, vp, vp, vp, vp, l, l, d, i, u, vp]
    L.f16_nlplant_batch.argtypes = [vp, vp, vp, vp, l, l, d, i, u, vp]
    L.f16_rollout.argtypes = [vp, vp, vp, vp, vp, l, l, i, i, d, d, i, u, vp]
    L.f16_rollout_lqr.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, l, l, i, i, d, d, i, u, vp]
    L.f16_rollout_lqr_linear.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, l, l, i, i, u, vp]
    L.f16_xdot_na_batch.argtypes = [vp, vp, vp, vp, vp, vp, l, l, d, i, u, vp]
    L.f16_debug_table_lookup.argtypes = [vp, i, vp, vp, vp, i, vp, vp]
    if hasattr(L, "f16_trim_batch"):
        L.f16_trim_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, l, l, d, i, u, i, vp, vp]
    if hasattr(L, "f16_linearise_batch"):
        L.f16_linearise_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, l, l, d, d, i, u, vp]
        L.f16_c2d_batch.argtypes = [vp, vp, vp, vp, vp, l, l, d, vp]
        L.f16_linearise_full_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, l, l, d, d, i, u, vp]
        L.f16_c2d_full_batch.argtypes = [vp, vp, vp, vp, vp, l, l, d, vp]
        L.f16_lqr_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, l, l, vp]
        L.f16_qp_default_settings.argtypes = [ctypes.POINTER(QPSettings)]
        L.f16_qp_default_settings.restype = None
        L.f16_mpc_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, l, l, i, d, ctypes.POINTER(QPSettings), vp]
        L.f16_mpc_hzn_sweep.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, l, l, i, i, d, ctypes.POINTER(QPSettings), vp]
        L.f16_mpc_qp_debug.argtypes = [vp, vp, vp, vp, vp, vp, l, l, i, d, vp, vp, vp, vp, vp]
        L.f16_debug_spd_inverse.argtypes = [vp, vp, vp, i, l, vp]
        L.f16_mpc_plan_create.argtypes = [vp, ctypes.POINTER(vp), vp, vp, vp, l, l, i, d, ctypes.POINTER(QPSettings), vp]
        L.f16_mpc_plan_solve.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        wp = ctypes.POINTER(MPCWeights)
        L.f16_mpc_default_weights.argtypes = [wp]
        L.f16_mpc_default_weights.restype = None
        L.f16_lqr_batch_w.argtypes = [vp, vp, vp, vp, wp, vp, vp, vp, l, l, vp]
        L.f16_mpc_batch_w.argtypes = [vp, vp, vp, vp, vp, vp, vp, wp, vp, vp, vp, vp, l, l, i, d, ctypes.POINTER(QPSettings), vp]
        L.f16_mpc_plan_create_w.argtypes = [vp, ctypes.POINTER(vp), vp, vp, vp, wp, l, l, i, d, ctypes.POINTER(QPSettings), vp]
        L.f16_mpc_plan_solve_w.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.f16_mpc_qp_debug_w.argtypes = [vp, vp, vp, vp, vp, vp, vp, wp, l, l, i, d, vp, vp, vp, vp, vp]
        L.f16_rollout_mpc.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, d, i, u, vp]
        L.f16_mpc_plan_warm_start.argtypes = [vp, i]
        L.f16_mpc_plan_destroy.argtypes = [vp]
        L.f16_mpc_plan_destroy.restype = None
    _LIB = L
    return L


def check(rc, L=None):
    if rc != 0:
        L = L or load()
        raise F16HipError(f"libf16hip error {rc}: {L.f16_last_error().decode()}")


class Context:
    """Owns one f16_ctx (device table image) on one GPU."""

    def __init__(self, device=0):
        self.lib = load()
        self.handle = ctypes.c_void_p()
        check(self.lib.f16_create(ctypes.byref(self.handle), int(device)), self.lib)
        self.device = int(device)

    def close(self):
        if self.handle:
            self.lib.f16_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
