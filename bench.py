#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: F-16 env steps/s (+ MPC solves/s), batch 4096, hifi model.

One bench "step" = one pass of the hot path over one batch: a 1000-Euler-step open-loop rollout of
B = 4096 hifi aircraft per GPU with every state stored (BASELINE config 2; [1000,18,4096] fp64 trajectory),
i.e. 4,096,000 aircraft-steps per launch.  value = aircraft-steps/s over all ranks.  Inputs are resident in HBM
before the timed region.  Multi-GPU: aircraft are independent, so ranks own disjoint batch shards (weak scaling,
no data-path collective); the max-over-ranks time is taken with one scalar all-reduce.

Extra keys on the same JSON line: "roofline" (dominant kernel k_rollout, HBM-bound by SURVEY 8(d) accounting,
timed with HIP events on the launch stream), "cpu_baseline" (the C oracle on the host cores, bounded sample),
"mpc" (batched calc_MPC_action throughput, N=30, xcg 0.35 -- BASELINE config 4).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_STORED_STEP = 144           # 18 doubles written per stored trajectory sample


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="aircraft per GPU")
    ap.add_argument("--euler-steps", type=int, default=1000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-mpc", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the large-batch roofline leg")
    ap.add_argument("--large-batch", type=int, default=262144)
    ap.add_argument("--mpc-hzn", type=int, default=30)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config2_states

    from f16_mpc_oop_py_amd import dist as fdist
    rank, world, local = fdist.init_from_env()
    local = local % max(torch.cuda.device_count(), 1)      # (only differs in a rehearsal with more ranks than GPUs)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    B, T = args.batch, args.euler_steps
    x0_all, u0_all = config2_states(B * world)        # global batch, contiguous shards
    x0, u0 = x0_all[rank * B:(rank + 1) * B], u0_all[rank * B:(rank + 1) * B]
    env = F16Batch(x0, u0, device=dev)
    traj = torch.empty((T, 18, B), dtype=torch.float64, device=dev)

    import ctypes
    from f16_mpc_oop_py_amd.env import _vp

    def one_pass():
        env._x.copy_(env._x_init)
        rc = env.lib.f16_rollout(env.ctx.handle, _vp(env._x), _vp(env._u), _vp(traj), _vp(env.status), B, B, T, 1,
                                 env.dt, env.xcg, env.fi_flag, env.flags, env._stream)
        assert rc == 0

    for _ in range(args.warmup):
        one_pass()
    barrier()
    # per-launch kernel time with HIP events on the stream the kernel is launched on (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for s, e in ev:
        env._x.copy_(env._x_init)
        s.record()
        rc = env.lib.f16_rollout(env.ctx.handle, _vp(env._x), _vp(env._u), _vp(traj), _vp(env.status), B, B, T, 1,
                                 env.dt, env.xcg, env.fi_flag, env.flags, env._stream)
        e.record()
        assert rc == 0
    barrier()
    elapsed = fdist.max_over_ranks(time.perf_counter() - t0, dev)
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    assert int(env.status.max()) == 0, "an aircraft left the envelope"
    assert bool(torch.isfinite(traj[-1]).all())

    steps_total = world * B * T * args.steps
    value = steps_total / elapsed
    achieved = B * T * BYTES_PER_STORED_STEP / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": "F16 env steps/sec (hifi Nguyen model, explicit Euler dt=1ms, batch 4096 per GPU)",
        "value": value, "unit": "aircraft-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: open-loop {T}-step rollout, B={B}/GPU, hifi, xcg=0.25, "
                               f"trajectory [T,18,B] stored every step", "batch_per_gpu": B, "euler_steps": T,
                   "parallelism": f"batch-sharded x{world}, no collective in the timed region"},
        "roofline": {"bound": "hbm", "kernel": "k_rollout_q" if B <= 4096 else ("k_rollout_4w" if B <= 16384 else "k_rollout"), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel_ms": kern_ms, "bytes_per_launch": B * T * BYTES_PER_STORED_STEP,
                     "note": "B=4096: 256 workgroups of 16 aircraft (four lanes per aircraft, four role wavefronts), one per CU; bound by the per-step dependency chain (lookup round trips + fp64 issue), not HBM (DESIGN.md 4)"},
    }

    out["roofline"]["traffic"] = recorded_traffic(B, T)
    if world > 1:
        # the one data-path collective -- all-gather of the trajectory shards -- timed on its own
        barrier()
        t0 = time.perf_counter()
        full = fdist.all_gather_trajectories(traj)
        barrier()
        tg = fdist.max_over_ranks(time.perf_counter() - t0, dev)
        out["allgather"] = {"ms": tg * 1e3, "bytes_received_per_gpu": int(full.numel() * 8),
                            "GB/s_per_gpu": full.numel() * 8 / tg / 1e9,
                            "steps_per_s_including_collation": world * B * T / (elapsed / args.steps + tg)}
        del full
    del traj
    if not args.no_large:
        out["roofline_large_batch"] = bench_large(args, dev, rank)
    if not args.no_mpc:
        out["mpc"] = bench_mpc(args, dev, rank, world, fdist, barrier)
        out["config5_closed_loop"] = bench_closed_loop(args, dev, rank, world, fdist, barrier)
        if rank == 0:
            out["trim"] = bench_trim(dev)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(x0, u0, T)
        if "mpc" in out:
            out["mpc"]["cpu_baseline"] = cpu_baseline_mpc(args.mpc_hzn)
        out["config1_dropin_loop"] = config1_dropin_loop()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        barrier()
        dist.destroy_process_group()


def recorded_traffic(B, T):
    """HBM bytes per k_rollout launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE/WRITE_SIZE,
    corrected as MI355X_MICROARCH.md prescribes); recorded under profiles/ by tools/pmc_summary.py.  None if the
    recorded run was a different workload."""
    path = os.path.join(REPO, "profiles", "traffic_k_rollout.json")
    try:
        rec = json.load(open(path))
        if rec.get("batch") == B and rec.get("euler_steps") == T:
            return rec["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def bench_large(args, dev, rank):
    """SURVEY.md 8(d) caveat: at B=4096 there are 64 wavefronts for 1024 SIMDs, so the HBM fraction of the same kernel
    family is also reported where the chip is full: B=262,144 aircraft per GPU, 200 Euler steps, every state stored
    (7.5 GB trajectory).  Same accounting (144 B per stored aircraft-step), HIP events on the launch stream."""
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config2_states
    from f16_mpc_oop_py_amd.env import _vp
    B, T = args.large_batch, 200
    x0, u0 = config2_states(4096)                      # the verified config-2 set, tiled
    reps = (B + 4095) // 4096
    x0, u0 = np.tile(x0, (reps, 1))[:B], np.tile(u0, (reps, 1))[:B]
    env = F16Batch(x0, u0, device=dev)
    traj = torch.empty((T, 18, B), dtype=torch.float64, device=dev)
    ts = []
    for i in range(2 + 5):
        env._x.copy_(env._x_init)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = env.lib.f16_rollout(env.ctx.handle, _vp(env._x), _vp(env._u), _vp(traj), _vp(env.status), B, B, T, 1,
                                 env.dt, env.xcg, env.fi_flag, env.flags, env._stream)
        e.record()
        assert rc == 0
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(s.elapsed_time(e))
    assert int(env.status.max()) == 0 and bool(torch.isfinite(traj[-1]).all())
    ms = float(np.mean(ts))
    gbs = B * T * BYTES_PER_STORED_STEP / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_rollout<512>" if B >= 131072 else "k_rollout<256>", "batch_per_gpu": B, "euler_steps": T, "kernel_ms": ms,
            "steps_per_s": B * T / (ms * 1e-3), "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": B * T * BYTES_PER_STORED_STEP,
            "note": "two waves per SIMD on all 256 CUs; fp64/VALU issue-bound (~1,700 instructions per aircraft-step)"}


def bench_mpc(args, dev, rank, world, fdist, barrier):
    """BASELINE config 4: B=4096/GPU, xcg=0.35, N=30, one calc_MPC_action per aircraft, (A,B) from each aircraft's
    own linearisation.  Same barrier + max-over-ranks timing rule as the dynamics leg."""
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config4_states
    B = args.batch
    x0, u0 = config4_states(B * world)
    env = F16Batch(x0[rank * B:(rank + 1) * B], u0[rank * B:(rank + 1) * B], xcg=0.35, device=dev)
    env.build_ssr()
    env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn)
    n = max(2, args.steps // 4)
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        u, info = env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn, return_info=True)
    barrier()
    dt = fdist.max_over_ranks(time.perf_counter() - t0, dev) / n
    it = info["iters"].cpu().numpy()
    assert fdist.or_status(info["status"]) == 0
    # the same without the dispatch-order heuristic (workgroups in the caller's order instead of longest-first by the
    # previous call's iteration counts): what a first call on an unseen batch costs
    os.environ["F16_MPC_DISPATCH_ORDER"] = "0"
    env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn)
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn)
    barrier()
    dco = fdist.max_over_ranks(time.perf_counter() - t0, dev) / n
    del os.environ["F16_MPC_DISPATCH_ORDER"]
    # linearise + ZOH + LQR chain (BASELINE config 3)
    env._calc_LQR_gain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        env._calc_LQR_gain()
    barrier()
    dl = fdist.max_over_ranks(time.perf_counter() - t0, dev) / n
    # the same solves with OSQP's fixed start value rho = 0.1 (the literal config-4 setting of SURVEY.md 8d; the default
    # above starts from 2 sqrt(tr P / tr A'A) because the QP is not Ruiz-scaled -- DESIGN.md 4)
    env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn, settings=dict(rho=0.1))
    barrier()
    t0 = time.perf_counter()
    for _ in range(2):
        _, info01 = env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn, settings=dict(rho=0.1), return_info=True)
    barrier()
    d01 = fdist.max_over_ranks(time.perf_counter() - t0, dev) / 2
    it01 = info01["iters"].cpu().numpy()
    flop_per_solve = 5.2e6 + 1.15e5 * float(np.mean(it))          # SURVEY.md 8(d) dense-form accounting
    mfma = None
    try:        # fp64 matrix-core counters of this same workload, recorded from the rocprofv3 --pmc pass (profiles/)
        rec = json.load(open(os.path.join(REPO, "profiles", "mfma_mpc.json")))
        if rec.get("batch") == B and rec.get("hzn") == args.mpc_hzn:
            mfma = {k: rec[k] for k in ("k_mpc_fast", "k_mpc<true> (build)", "source")}
    except Exception:
        pass
    return {"metric": "MPC solves/sec (calc_MPC_action, N=%d, batch %d per GPU, xcg=0.35)" % (args.mpc_hzn, B),
            "value": world * B / dt, "unit": "solves/s", "ms_per_batch": dt * 1e3,
            "admm_iters": {"min": float(it.min()), "median": float(np.median(it)), "max": float(it.max())},
            "roofline": {"bound": "fp64 vector/MFMA (78.6 TF/s)", "achieved": flop_per_solve * B / dt / 1e12,
                         "peak": 78.6, "unit": "TFLOP/s", "frac": flop_per_solve * B / dt / 78.6e12,
                         "note": "dense-form FLOP accounting of SURVEY 8(d) over the measured time; the kernels use the "
                                 "Toeplitz recursion, so issued FLOPs are lower"},
            "admm_iters_mean": float(np.mean(it)), "mfma": mfma,
            "dispatch": {"order": "longest-first by the previous call's iteration counts (any order gives the same results)",
                         "value_in_caller_order": world * B / dco, "ms_per_batch_in_caller_order": dco * 1e3},
            "rho_start_0p1": {"value": world * B / d01, "unit": "solves/s", "ms_per_batch": d01 * 1e3,
                              "admm_iters": {"min": float(it01.min()), "median": float(np.median(it01)),
                                             "max": float(it01.max()), "mean": float(it01.mean())},
                              "status_or": int(fdist.or_status(info01["status"]))},
            "linearise_zoh_lqr_per_s": world * B / dl}


def bench_trim(dev, B=4096):
    """straight-and-level trim of B flight conditions in one launch (the reference's Nelder-Mead,
    env.py:198-292, one wavefront lane per condition); the reference takes 0.56 s for one condition."""
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    rng = np.random.default_rng(7)
    h = rng.uniform(5e3, 3e4, B)
    v = rng.uniform(450.0, 800.0, B)
    F16Batch.trim(h[:64], v[:64], device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, info = F16Batch.trim(h, v, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cost = info["cost"].cpu().numpy()
    return {"conditions": B, "ms": dt * 1e3, "trims_per_s": B / dt, "cost_max": float(cost.max()),
            "nfev_mean": float(info["nfev"].double().mean()), "nfev_max": int(info["nfev"].max()),
            "reference_s_per_trim": 0.56,
            "note": "one lane per condition: the launch lasts as long as its slowest member (conditions that are not "
                    "trimmable run Nelder-Mead to the reference's maxiter = 50,000 iterations)"}


def bench_closed_loop(args, dev, rank, world, fdist, barrier, B=8192, T=20):
    """BASELINE config 5 shape per GPU: B = 8192 aircraft, closed loop (calc_MPC_action N = 30, then one Euler step),
    T steps, trajectory kept on the device ([T,18,B]; the one all-gather is timed in the open-loop leg).  The model is
    frozen at construction as in the reference (env.py:49-60), so the model-only part of the QP is a prepared plan;
    the one-shot figure (everything rebuilt per call, as the reference does) is measured beside it."""
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config4_states
    x0, u0 = config4_states(B * world)
    sl = slice(rank * B, (rank + 1) * B)
    res = {}
    for name, use_plan, steps in (("prepared_plan", True, T), ("one_shot", False, max(4, T // 4)),
                                  ("prepared_plan_warm_start", True, T), ("prepared_plan_warm_start_check5", True, T)):
        env = F16Batch(x0[sl], u0[sl], xcg=0.35, device=dev)
        env.build_ssr()
        if use_plan:
            env.prepare_MPC(args.mpc_hzn, settings=dict(check_every=5) if name.endswith("check5") else None,
                            warm_start="warm_start" in name)
        fdist.closed_loop_mpc_rollout(env, steps=2, hzn=args.mpc_hzn, gather=False, use_plan=use_plan)
        barrier()
        t0 = time.perf_counter()
        traj = fdist.closed_loop_mpc_rollout(env, steps=steps, hzn=args.mpc_hzn, gather=False, use_plan=use_plan)
        barrier()
        dt = fdist.max_over_ranks(time.perf_counter() - t0, dev)
        assert bool(torch.isfinite(traj).all()) and fdist.or_status(env.status) & ~(64 | 128) == 0
        res[name] = {"aircraft_steps_per_s": world * B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps}
        del env, traj
    res["batch_per_gpu"] = B
    res["hzn"] = args.mpc_hzn
    res["note"] = ("prepared_plan / one_shot start every solve cold, as the reference does (a new OSQP object per call); "
                   "warm_start is the opt-in extension (OSQP's in-object default); check5 = the same with the termination test "
                   "every 5 iterations instead of OSQP's 25 (a warm-started solve needs fewer than 25)")
    return res


def cpu_baseline(x0, u0, T):
    """The C restatement of the reference path (oracle/, kind 'port') on the host cores: bounded sample of the same
    workload (first n aircraft x T steps), single thread and all cores."""
    from oracle import mpc_oracle as mo
    ora = mo.COracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)
    n1 = 256
    t0 = time.perf_counter()
    ora.rollout(x0[:n1], u0[:n1], T, store=True, nthreads=1)
    t1 = time.perf_counter() - t0
    nall = min(len(x0), 256 * cores)
    t0 = time.perf_counter()
    ora.rollout(x0[:nall], u0[:nall], T, store=True, nthreads=cores)
    tall = time.perf_counter() - t0
    return {"value": nall * T / tall, "unit": "aircraft-steps/s", "cores": cores, "kind": "port",
            "single_thread_value": n1 * T / t1,
            "sample": f"first {nall} aircraft x {T} steps of the same workload on {cores} threads "
                      f"(single-thread figure: first {n1} aircraft); oracle/libf16_oracle.so (C restatement)"}


def cpu_baseline_mpc(hzn, n=12):
    """the MPC CPU baseline is the same-algorithm numpy restatement (oracle/, kind 'port') on the same
    config-4 flight conditions -- OSQP itself cannot be timed anywhere in this pipeline (not installable offline).
    Bounded sample: n aircraft, single thread; whole chain per solve = linearise + ZOH + setup_OSQP + ADMM."""
    from oracle import mpc_oracle as mo
    from f16_mpc_oop_py_amd.workload import config4_states
    ora = mo.COracle()
    x0, _ = config4_states(n)
    t0 = time.perf_counter()
    its = []
    for b in range(n):
        Ac, Bc, Cc, Dc = ora.linearise_na(x0[b], xcg=0.35)
        Ad, Bd, Cd, _ = mo.c2d(Ac, Bc, Cc, Dc, 0.001)
        P, q, A, l, u = mo.mpc_qp(x0[b], Ad, Bd, Cd, hzn, 0.001)
        its.append(mo.admm_osqp_style(P, q, A, l, u)["iters"])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
            "sample": f"{n} config-4 aircraft, N={hzn}: C linearise + scipy ZOH/DARE + numpy setup_OSQP + numpy ADMM "
                      f"(same rules as the kernel; mean {float(np.mean(its)):.0f} iterations); one Python thread, numpy/BLAS may use the cores listed"}


def config1_dropin_loop(steps=2000):
    """BASELINE config 1 shape (1 aircraft, lofi, open-loop Euler through the reference-style caller): a Python loop
    over the drop-in `Nlplant` / `atmos` symbols of libf16hip.so (host pointers, one aircraft per call on the GPU) next
    to the same loop over the C restatement on the CPU.  us per step, lower is better; not the product's use case."""
    import ctypes
    from f16_mpc_oop_py_amd import lib
    from f16_mpc_oop_py_amd import parameters as P
    from oracle import mpc_oracle as mo
    L = lib.load()
    L.f16_dropin_config(0.25, 0)
    x = np.array(P.x0, dtype=np.float64) if hasattr(P, "x0") else None
    if x is None:
        from f16_mpc_oop_py_amd.workload import config2_states
        x = config2_states(1)[0][0].copy()
    xdot = np.zeros(18)
    vp = ctypes.c_void_p

    def loop(fn):
        xx = x.copy()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(vp(xx.ctypes.data), vp(xdot.ctypes.data), 0)
            xx[:12] += xdot[:12] * 0.001
        return (time.perf_counter() - t0) / steps * 1e6

    gpu_us = loop(L.Nlplant)
    ora = mo.COracle()
    cpu_us = loop(ora.lib.Nlplant) if hasattr(ora, "lib") and hasattr(ora.lib, "Nlplant") else None
    return {"steps": steps, "fidelity": "lofi", "us_per_step_dropin_gpu_symbol": gpu_us,
            "us_per_step_cpu_restatement_symbol": cpu_us, "unit": "us/step"}


if __name__ == "__main__":
    main()
