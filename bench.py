#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: F-16 env steps/s (+ MPC solves/s), batch 4096, hifi model.

One bench "step" = one pass of the hot path over one batch: a 1000-Euler-step open-loop rollout of
B = 4096 hifi aircraft per GPU with every state stored (BASELINE config 2; [1000,18,4096] fp64 trajectory),
i.e. 4,096,000 aircraft-steps per launch.  value = aircraft-steps/s over all ranks.  Inputs are resident in HBM
before the timed region.  Multi-GPU: aircraft are independent, so ranks own disjoint batch shards (weak scaling,
no data-path collective); the max-over-ranks time is taken with one scalar all-reduce.

Launching.  `python bench.py --gpus N` starts N ranks ITSELF when it was not started by a launcher (RANK unset): the
parent process -- which never imports torch or touches the GPU -- spawns N fresh children, one per GPU
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), relays rank 0's JSON line
and exits with the children's status.  Under `python -m torch.distributed.run ... bench.py --gpus N` the environment is
already set and each process is a rank.  `n_gpus` in the JSON is the world size the process group actually has.

Extra keys on the same JSON line: "roofline" (dominant kernel, HBM-bound by SURVEY 8(d) accounting, timed with HIP
events on the launch stream), "cpu_baseline" (the C oracle on the host cores, bounded sample), "mpc" (batched
calc_MPC_action throughput, N=30, xcg 0.35 -- BASELINE config 4), "config5_closed_loop" (BASELINE config 5 per GPU:
8192 aircraft, T=100 closed-loop steps, then the timed all-gather of the [T,18,8192] shards).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_STORED_STEP = 144           # 18 doubles written per stored trajectory sample


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="aircraft per GPU")
    ap.add_argument("--euler-steps", type=int, default=1000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-mpc", action="store_true")
    ap.add_argument("--no-large", action="store_true", help="skip the large-batch roofline leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the closed-loop leg")
    ap.add_argument("--allgather-direct", action="store_true",
                    help="multi-GPU runs: also time the all-gather as direct point-to-point sends / receives")
    ap.add_argument("--large-batch", type=int, default=262144)
    ap.add_argument("--mpc-hzn", type=int, default=30)
    ap.add_argument("--config5-batch", type=int, default=8192, help="aircraft per GPU in the closed-loop leg")
    ap.add_argument("--config5-steps", type=int, default=100)
    ap.add_argument("--rank-timeout", type=float, default=900.0,
                    help="self-launched multi-rank runs (--gpus N without a launcher): seconds after which the parent kills "
                         "the ranks it started and exits non-zero with their stderr tails")
    ap.add_argument("--dry-run-stall-rank", type=int, default=-1,
                    help="(test) --dry-run: this rank sleeps instead of entering the first collective")
    ap.add_argument("--force-group", action="store_true",
                    help="form a process group even with one rank (F16_DIST_FORCE_GROUP=1): the all-gather and the scalar "
                         "reductions then execute through RCCL on a one-GPU box, and the `allgather` block is printed")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch path only: rendezvous, shard bookkeeping and the collectives on CPU tensors (no GPU needed)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(n, argv, deadline_s):
    """Parent of a multi-rank run.  Must not import torch / touch the GPU: children are fresh processes (never an exec of
    a process that has initialised HIP).  Rank 0's stdout is relayed, the other ranks' stdout goes to stderr; every rank's
    stderr is relayed AND its tail kept.  The run is bounded in time: when `deadline_s` expires (a rank stalled in its
    first collective would otherwise hold the caller until an external limit) or a rank dies, the children started here
    -- exactly those -- are killed, the ranks' stderr tails are printed and the parent exits non-zero."""
    import collections
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, tails, threads = [], [], []
    chunks = []

    def pump(stream, tail, sink):
        for line in iter(stream.readline, b""):
            tail.append(line)
            if sink is not None:
                sink.write(line.decode(errors="replace"))
                sink.flush()

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        procs.append(p)
        tails.append(collections.deque(maxlen=40))
        out_tail = collections.deque(maxlen=100000) if r == 0 else collections.deque(maxlen=40)
        if r == 0:
            chunks = out_tail
        threads.append(threading.Thread(target=pump, args=(p.stdout, out_tail, None if r == 0 else sys.stderr), daemon=True))
        threads.append(threading.Thread(target=pump, args=(p.stderr, tails[r], sys.stderr), daemon=True))
    for t in threads:
        t.start()
    t_end = time.monotonic() + deadline_s
    rc, why = 0, None
    while any(p.poll() is None for p in procs):
        failed = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        if failed:                              # a rank died: the others would wait in a collective until its timeout
            rc, why = procs[failed[0]].returncode, f"rank {failed[0]} exited with status {procs[failed[0]].returncode}"
            time.sleep(2.0)
        elif time.monotonic() > t_end:
            rc, why = 124, f"no result within --rank-timeout {deadline_s:.0f} s (ranks still running: " \
                           f"{[i for i, p in enumerate(procs) if p.poll() is None]})"
        if why:
            for p in procs:
                if p.poll() is None:
                    p.kill()                    # exactly the children started above
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    for t in threads:
        t.join(5)
    sys.stdout.write(b"".join(chunks).decode(errors="replace"))
    sys.stdout.flush()
    if why:
        sys.stderr.write(f"bench.py launcher: {why}; killed the remaining ranks\n")
        for r in range(n):
            sys.stderr.write(f"---- rank {r} stderr tail ----\n" + b"".join(tails[r]).decode(errors="replace"))
        sys.stderr.flush()
    return rc or (1 if why else 0)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.rank_timeout))
    if args.force_group:
        os.environ["F16_DIST_FORCE_GROUP"] = "1"
    if args.dry_run:
        return dry_run(args)
    run(args)


def dry_run(args):
    """The multi-rank plumbing of run() without the kernels: process group from the environment, contiguous shards,
    the one data-path collective (all-gather of trajectory shards, both layouts) and the scalar reductions."""
    import torch
    import torch.distributed as dist
    from f16_mpc_oop_py_amd import dist as fdist
    rank, world, local = fdist.init_from_env()
    if rank == args.dry_run_stall_rank:          # (test of the launcher's deadline: a rank that never arrives)
        time.sleep(3600)
    T, Bl = 4, 6
    lo, hi = fdist.shard_bounds(world * Bl, world, rank)
    full = torch.arange(T * 18 * world * Bl, dtype=torch.float64).reshape(T, 18, world * Bl)
    mine = full[:, :, lo:hi].contiguous()
    flat = fdist.all_gather_trajectories(mine, chunk_bytes=18 * Bl * world * 8 * 3)     # three samples per chunk: ragged tail
    ranks = fdist.all_gather_trajectories(mine, layout="ranks")
    ok = bool(torch.equal(flat, full)) and bool(torch.equal(ranks.reshape(T, 18, -1), full))
    t = fdist.max_over_ranks(1.0 + rank)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": fdist.world_size(), "requested_gpus": args.gpus,
                          "allgather_ok": ok, "max_over_ranks": t,
                          "backend": dist.get_backend() if fdist.group_active() else None, "versions": versions()}))
    if fdist.group_active():
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


def versions():
    """torch / HIP / RCCL versions of the process group that actually ran (rank 0 prints them with the result)."""
    import torch
    v = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None)}
    try:
        v["rccl"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        v["rccl"] = None
    return v


# ------------------------------------------------------------------------------------------------ the bench
def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config2_states, in_grid_on_gpu
    from f16_mpc_oop_py_amd import dist as fdist
    from f16_mpc_oop_py_amd.env import _vp

    rank, world, local = fdist.init_from_env()
    local = local % max(torch.cuda.device_count(), 1)      # (only differs in a rehearsal with more ranks than GPUs)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    grouped = fdist.group_active()              # (world > 1, or one rank with --force-group)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    B, T = args.batch, args.euler_steps
    # global batch, contiguous shards; 8(d)'s rule: a candidate that leaves the grid within T steps is resampled from
    # the same seed stream (every rank applies it to the whole global batch on its own GPU: same batch everywhere)
    x0_all, u0_all = config2_states(B * world, accept=in_grid_on_gpu(0.25, steps=T, device=dev))
    x0, u0 = x0_all[rank * B:(rank + 1) * B], u0_all[rank * B:(rank + 1) * B]
    env = F16Batch(x0, u0, device=dev)
    traj = torch.empty((T, 18, B), dtype=torch.float64, device=dev)

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
    kname = "k_rollout_q" if B <= 8192 else ("k_rollout_4w" if B <= 16384 else "k_rollout")
    out = {
        "metric": "F16 env steps/sec (hifi Nguyen model, explicit Euler dt=1ms, batch 4096 per GPU)",
        "value": value, "unit": "aircraft-steps/s", "n_gpus": fdist.world_size(), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE config 2: open-loop {T}-step rollout, B={B}/GPU, hifi, xcg=0.25, "
                               f"trajectory [T,18,B] stored every step", "batch_per_gpu": B, "euler_steps": T,
                   "parallelism": f"batch-sharded x{world}, no collective in the timed region",
                   "requested_gpus": args.gpus},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel_ms": kern_ms, "bytes_per_launch": B * T * BYTES_PER_STORED_STEP,
                     "note": "B=4096: 256 workgroups of 16 aircraft (four lanes per aircraft, four role wavefronts), one per CU; bound by the per-step dependency chain (lookup round trips + fp64 issue), not HBM (DESIGN.md 4)"},
        "versions": versions(), "backend": dist.get_backend() if grouped else None,
        "scaling_note": "1/2/4/8-GPU values exist only where the driver ran this command on an 8-GPU node; the builder's "
                        "box has one GPU (multi-rank paths rehearsed there with F16_DIST_BACKEND=gloo)",
    }
    tr, src = recorded_traffic(B, T)
    out["roofline"]["traffic"] = tr
    out["roofline"]["traffic_source"] = src
    out["roofline"]["issue"] = recorded_issue("k_rollout_q", kern_ms * 1e-3, batch=B, euler_steps=T)
    if grouped:
        # the one data-path collective -- all-gather of the trajectory shards -- timed on its own
        barrier()
        t0 = time.perf_counter()
        full = fdist.all_gather_trajectories(traj, layout="ranks")
        barrier()
        tg = fdist.max_over_ranks(time.perf_counter() - t0, dev)
        out["allgather"] = {"ms": tg * 1e3, "bytes_received_per_gpu": int(full.numel() * 8),
                            "GB/s_per_gpu": full.numel() * 8 / tg / 1e9, "layout": "[T,18,W,B/W] view of the receive buffer",
                            "algo": "collective (all_gather_into_tensor)",
                            "steps_per_s_including_collation": world * B * T / (elapsed / args.steps + tg)}
        del full
        if args.allgather_direct:
            # opt-in (--allgather-direct): the same collation as W - 1 direct sends / receives per rank (one peer per
            # xGMI link), for comparison; not in the default run (unrehearsed on RCCL hardware: a stall would cost the
            # scaling run)
            barrier()
            t0 = time.perf_counter()
            full = fdist.all_gather_trajectories(traj, layout="ranks", algo="direct")
            barrier()
            td = fdist.max_over_ranks(time.perf_counter() - t0, dev)
            out["allgather"]["direct_p2p"] = {"ms": td * 1e3, "GB/s_per_gpu": full.numel() * 8 / td / 1e9}
            del full
    del traj
    out["lqr_closed_loop"] = bench_lqr(args, dev, rank, world, fdist, barrier, value)
    if not args.no_large:
        out["roofline_large_batch"] = bench_large(args, dev, rank)
    if not args.no_mpc:
        out["mpc"] = bench_mpc(args, dev, rank, world, fdist, barrier)
        if rank == 0:
            out["trim"] = bench_trim(dev)
            out["hzn_sweep"] = bench_hzn_sweep(dev)
    if not args.no_config5:
        out["config5_closed_loop"] = bench_closed_loop(args, dev, rank, world, fdist, barrier)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(x0, u0, T)
        if "mpc" in out:
            out["mpc"]["cpu_baseline"] = cpu_baseline_mpc(args.mpc_hzn)
        out["config1_reference_style_loop"] = config1_reference_style_loop()
    # The second half of BASELINE's metric (MPC solves/s) and config 5, compact, INSIDE the object the driver's record keeps verbatim
    # (`roofline`; the full `mpc` / `config5_closed_loop` blocks stay beside it).  The first-call figure leads: BASELINE config 4 as
    # written is one call per aircraft on an unseen batch.
    if "mpc" in out:
        m = out["mpc"]
        r = m.get("roofline", {})
        out["roofline"]["mpc"] = {
            "workload": "BASELINE config 4: B=%d/GPU, N=%d, xcg 0.35, osqp defaults, one calc_MPC_action per aircraft" % (args.batch, args.mpc_hzn),
            "first_call_solves_per_s": m.get("first_call_value"), "repeated_solves_per_s": m.get("repeated_call_value"),
            "ms_per_batch_first_call": m.get("ms_per_batch"), "ms_per_batch_repeated": m.get("repeated_call_ms_per_batch"),
            "iters_mean": m.get("admm_iters", {}).get("mean"),
            "frac_issued": (r.get("issued_flop_per_launch") or 0.0) / (m["repeated_call_ms_per_batch"] * 1e-3) / 78.6e12 if r.get("issued_flop_per_launch") else None,
            "frac_dense_form": r.get("frac"), "mfma_busy_frac": r.get("mfma_busy_frac"),
            "cpu_solves_per_s": (m.get("cpu_baseline") or {}).get("value"), "cpu_cores": (m.get("cpu_baseline") or {}).get("cores")}
    if "config5_closed_loop" in out:
        c = out["config5_closed_loop"]
        out["roofline"]["config5"] = {
            "workload": "BASELINE config 5 per GPU: B=%d, N=%d, T=%d closed-loop steps, reference settings, cold start" % (c["batch_per_gpu"], c["hzn"], c["steps"]),
            "aircraft_steps_per_s_one_launch": c["headline"]["aircraft_steps_per_s"],
            "aircraft_steps_per_s_host_loop": c["host_loop"]["aircraft_steps_per_s"],
            "aircraft_steps_per_s_one_launch_hold_command": c["fused_hold_command"]["aircraft_steps_per_s"],
            "iters_mean": c["headline"]["iters_mean"], "aircraft_infeasible_at_some_step": c["headline"]["aircraft_infeasible_at_some_step"],
            "aircraft_not_finite_at_the_end": c["headline"]["aircraft_not_finite_at_the_end"],
            "one_launch_equals_host_loop_bit_for_bit": c["one_launch_equals_host_loop_bit_for_bit"]}
        try:        # issued fp64 FLOPs of the one-launch loop, RECORDED from rocprofv3 counter passes of the same workload (tools/c5_prof_summary.py)
            rec5 = json.load(open(os.path.join(REPO, "profiles", "config5_counters.json")))
            if rec5.get("batch") == c["batch_per_gpu"] and rec5.get("steps") == c["steps"] and rec5.get("hzn") == c["hzn"]:
                sec = c["batch_per_gpu"] * c["steps"] / (c["headline"]["aircraft_steps_per_s"] / world)
                out["roofline"]["config5"].update({
                    "bound": "fp64 vector issue of one wavefront per SIMD (the ADMM iteration)", "issued_flop_per_launch": rec5["issued_flop_per_launch"],
                    "achieved_tflops": rec5["issued_flop_per_launch"] / sec / 1e12, "peak_tflops": 78.6,
                    "frac_issued": rec5["issued_flop_per_launch"] / sec / 78.6e12, "counters_source": "recorded: profiles/config5_counters.json (" + str(rec5.get("source")) + ")"})
        except Exception:
            pass
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if grouped:
        barrier()
        dist.destroy_process_group()


SIMDS, ISSUE_PER_SIMD = 1024, 0.6e9       # 256 CUs x 4 SIMDs; one VALU wave-instruction per 4 cycles at 2.4 GHz (MI355X_MICROARCH.md)


def recorded_issue(kernel, seconds, **match):
    """Secondary roofline for kernels that are bound by instruction issue, not by HBM: VALU wave-instructions per launch
    (SQ_INSTS_VALU of a rocprofv3 --pmc pass of this same command, RECORDED in profiles/issue_valu.json by tools/*_pmc_summary.py)
    over what 1024 SIMDs can issue in the launch's measured duration.  None if the recorded run was another workload."""
    try:
        rec = json.load(open(os.path.join(REPO, "profiles", "issue_valu.json")))[kernel]
    except Exception:
        return None
    if any(rec.get(k) != v for k, v in match.items()):
        return None
    src = str(rec.get("source") or "")
    if src.startswith("profiles/"):      # the counter file the record cites must exist and hold rows: no figure from evidence that is not there
        try:
            with open(os.path.join(REPO, src)) as f:
                if sum(1 for _ in f) < 2:
                    return None
        except OSError:
            return None
    insts = rec["insts_valu_per_launch"]
    return {"bound": "VALU issue", "achieved": insts / seconds / 1e9, "peak": SIMDS * ISSUE_PER_SIMD / 1e9, "unit": "G wave-instructions/s",
            "frac": insts / seconds / (SIMDS * ISSUE_PER_SIMD), "insts_valu_per_launch": insts,
            "source": "recorded: profiles/issue_valu.json (" + str(rec.get("source")) + ")",
            "note": "an fp64 FMA with three register operands occupies a SIMD for 6 cycles, not 4, when one wavefront is resident "
                    "(tools/micro/issue_mix.hip, profiles/r04_issue_mix.log): 1.0 is not reachable by fp64 code at one wave per SIMD"}


def recorded_traffic(B, T):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of this same command
    (FETCH_SIZE/WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes), RECORDED under profiles/ by
    tools/pmc_summary.py -- not measured in this run (counters need rocprofv3).  (None, None) if the recorded run was a
    different workload."""
    path = os.path.join(REPO, "profiles", "traffic_k_rollout.json")
    try:
        rec = json.load(open(path))
        if rec.get("batch") == B and rec.get("euler_steps") == T:
            return rec["hbm_bytes_per_launch"], "recorded: profiles/traffic_k_rollout.json (rocprofv3 --pmc pass of this command)"
    except Exception:
        pass
    return None, None


def bench_lqr(args, dev, rank, world, fdist, barrier, open_loop_value):
    """The reference's running controller (flight_sim.py:139,181; test_env_mk2.py:70-85): closed loop under the LQR law, the action
    of env.py:360-371 computed inside the rollout kernel every step (f16_rollout_lqr).  Same batch, steps and stored trajectory as
    the headline open-loop leg; gains from each aircraft's own linearisation (outside the timed region)."""
    import numpy as np
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config2_states
    from f16_mpc_oop_py_amd.env import _vp
    B, T = args.batch, args.euler_steps
    x0, u0 = config2_states(B * world)
    env = F16Batch(x0[rank * B:(rank + 1) * B], u0[rank * B:(rank + 1) * B], device=dev)
    K = env._calc_LQR_gain().reshape(B, 27).t().contiguous()
    dem = torch.zeros((3, B), dtype=torch.float64, device=dev)
    traj = torch.empty((T, 18, B), dtype=torch.float64, device=dev)
    n = max(3, args.steps // 2)

    def one():
        env._x.copy_(env._x_init)
        rc = env.lib.f16_rollout_lqr(env.ctx.handle, _vp(env._x), _vp(env._u_init), _vp(K), _vp(dem), _vp(traj), _vp(env._u), _vp(env.status),
                                     B, B, T, 1, env.dt, env.xcg, env.fi_flag, env.flags, env._stream)
        assert rc == 0
    one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    barrier()
    dt = fdist.max_over_ranks(time.perf_counter() - t0, dev) / n
    v = world * B * T / dt
    return {"value": v, "unit": "aircraft-steps/s", "ms_per_rollout": dt * 1e3, "ratio_to_open_loop": v / open_loop_value,
            "frozen_aircraft": int(((env.status & 16) != 0).sum()), "finite": bool(torch.isfinite(traj[-1]).all()),
            "note": "f16_rollout_lqr: u[1:4] = -K (x_ref - x9) + u0[1:4] per step inside the kernel, thrust held; K per aircraft"}


def bench_large(args, dev, rank):
    """SURVEY.md 8(d) caveat: at B=4096 there are 64 wavefronts for 1024 SIMDs, so the HBM fraction of the same kernel
    family is also reported where the chip is full: B=262,144 aircraft per GPU, 200 Euler steps, every state stored
    (7.5 GB trajectory).  Same accounting (144 B per stored aircraft-step), HIP events on the launch stream."""
    import numpy as np
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
    res = {"bound": "hbm", "kernel": "k_rollout_i<512>" if B >= 131072 else "k_rollout<256>", "batch_per_gpu": B, "euler_steps": T, "kernel_ms": ms,
           "steps_per_s": B * T / (ms * 1e-3), "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": B * T * BYTES_PER_STORED_STEP, "traffic": None,
           "note": "two waves per SIMD on all 256 CUs; fp64/VALU issue-bound"}
    res["issue"] = recorded_issue("k_rollout_i", ms * 1e-3, batch=B, euler_steps=T)
    try:
        rec = json.load(open(os.path.join(REPO, "profiles", "traffic_k_rollout_large.json")))
        if rec.get("batch") == B and rec.get("euler_steps") == T:
            res["traffic"] = rec["hbm_bytes_per_launch"]
            res["traffic_source"] = "recorded: profiles/traffic_k_rollout_large.json"
    except Exception:
        pass
    return res


def bench_mpc(args, dev, rank, world, fdist, barrier):
    """BASELINE config 4: B=4096/GPU, xcg=0.35, N=30, one calc_MPC_action per aircraft, (A,B) from each aircraft's
    own linearisation.  Same barrier + max-over-ranks timing rule as the dynamics leg.  The headline `value` is measured
    with the solver settings the reference's call implies (osqp defaults, env.py:420-422: Ruiz equilibration, rho = 0.1,
    adaptive rho); the builder's own start-value rule (no scaling, rho0 = 2 sqrt(tr P / tr A'A)) is reported beside it."""
    import numpy as np
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config4_states
    B = args.batch
    x0, u0 = config4_states(B * world)
    env = F16Batch(x0[rank * B:(rank + 1) * B], u0[rank * B:(rank + 1) * B], xcg=0.35, device=dev)
    env.build_ssr()
    n = max(2, args.steps // 4)

    def timed(settings, reps):
        env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn, settings=settings)
        env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn, settings=settings)      # (second call: dispatch order known)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            _, info = env._calc_MPC_action(0.0, 0.0, 0.0, args.mpc_hzn, settings=settings, return_info=True)
        barrier()
        dt = fdist.max_over_ranks(time.perf_counter() - t0, dev) / reps
        it = info["iters"].cpu().numpy()
        return dt, it, int(fdist.or_status(info["status"]))

    rec = None
    try:        # counters of this same workload, RECORDED from rocprofv3 passes of tools/profile_round.sh (profiles/)
        rec = json.load(open(os.path.join(REPO, "profiles", "mfma_mpc.json")))
        if rec.get("batch") != B or rec.get("hzn") != args.mpc_hzn:
            rec = None
    except Exception:
        rec = None

    def leg(settings, reps, headline=False):
        dt, it, st = timed(settings, reps)
        flop = 5.2e6 + 1.15e5 * float(np.mean(it))          # SURVEY.md 8(d) dense-form accounting
        roof = {"bound": "fp64 vector/MFMA (78.6 TF/s)", "achieved": flop * B / dt / 1e12, "peak": 78.6,
                "unit": "TFLOP/s", "frac": flop * B / dt / 78.6e12,
                "note": "achieved = dense-form FLOP accounting of SURVEY 8(d) (5.2 M + 0.115 M x iterations per solve) over the "
                        "measured time"}
        if headline and rec and "solve_kernel" in rec:
            k = rec["solve_kernel"]
            # what the solver kernel actually issues (recorded SQ counters of the headline settings): fp64 vector FLOPs =
            # 64 lanes x (2 FMA + ADD + MUL) wave-instructions, matrix-core FLOPs = 512 x MOPS; MFMA-busy = busy cycles over
            # (1024 SIMDs x kernel cycles)
            roof["issued_flop_per_launch"] = k.get("issued_flop_per_launch")
            roof["issued_over_dense"] = (k.get("issued_flop_per_launch") or 0.0) / (flop * B) if flop else None
            roof["mfma_flop_per_launch"] = k.get("mfma_flop_per_launch")
            roof["mfma_busy_frac"] = k.get("mfma_busy_frac")
            roof["mfma_tflops"] = k.get("mfma_tflops")
            roof["kernel"] = k.get("name")
            roof["kernel_ms_rocprof"] = k.get("avg_ms")
            roof["counters_source"] = "recorded: profiles/mfma_mpc.json (" + str(rec.get("source")) + ")"
            roof["note"] += ("; issued FLOPs and the MFMA-busy fraction are counter-derived (recorded rocprofv3 passes, headline "
                             "settings only): the matrix cores are used for the Gram product and the KKT factorisations only, "
                             "the iterations are fp64 vector work")
        return {"value": world * B / dt, "unit": "solves/s", "ms_per_batch": dt * 1e3, "status_or": st,
                "admm_iters": {"min": float(it.min()), "median": float(np.median(it)), "max": float(it.max()),
                               "mean": float(it.mean())},
                "roofline": roof}

    modes = env.solver_modes()
    res = {"metric": "MPC solves/sec (calc_MPC_action, N=%d, batch %d per GPU, xcg=0.35)" % (args.mpc_hzn, B)}
    head = "osqp_defaults" if "osqp_defaults" in modes else next(iter(modes))
    legs = {name: leg(s, n, headline=(name == head)) for name, s in modes.items()}
    res.update(legs[head])
    res["settings"] = head
    res["other_settings"] = {k: v for k, v in legs.items() if k != head}
    assert res["status_or"] == 0
    # the same without the dispatch-order heuristic (workgroups in the caller's order instead of longest-first by the
    # previous call's iteration counts): what a first call on an unseen batch costs
    os.environ["F16_MPC_DISPATCH_ORDER"] = "0"
    dco, _, _ = timed(modes[head], n)
    os.environ["F16_MPC_DISPATCH_ORDER"] = "first"      # every call treated as a first one: order from the QPs themselves (||q||_inf)
    dfirst, _, _ = timed(modes[head], n)
    del os.environ["F16_MPC_DISPATCH_ORDER"]
    res["dispatch"] = {"order": "longest-first by the previous call's iteration counts (any order gives the same results); a first call: "
                                "longest-first by ||q||_inf of the QPs just built",
                       "queue": ("one workgroup per SIMD takes the next aircraft of that order from an atomic counter"
                                 if os.environ.get("F16_MPC_WAVE_QUEUE", "1")[:1] != "0" else
                                 "F16_MPC_WAVE_QUEUE=0: one workgroup per aircraft, dealt round-robin to 32 (XCD, shader engine) partitions by the hardware"),
                       "value_in_caller_order": world * B / dco, "ms_per_batch_in_caller_order": dco * 1e3,
                       "value_first_call": world * B / dfirst, "ms_per_batch_first_call": dfirst * 1e3}
    if world == 1 and os.environ.get("F16_MPC_WAVE_QUEUE", "1")[:1] != "0":
        # the same repeated call with one workgroup per aircraft dealt by the hardware (the switch is read once per process: a child)
        code = ("import sys, time, torch; sys.path.insert(0, %r)\n"
                "from f16_mpc_oop_py_amd import F16Batch\nfrom f16_mpc_oop_py_amd.workload import config4_states\n"
                "x0, u0 = config4_states(%d); env = F16Batch(x0, u0, xcg=0.35); env.build_ssr()\n"
                "for _ in range(2): env._calc_MPC_action(0.0, 0.0, 0.0, %d)\n"
                "torch.cuda.synchronize(); t0 = time.perf_counter()\n"
                "for _ in range(%d): env._calc_MPC_action(0.0, 0.0, 0.0, %d)\n"
                "torch.cuda.synchronize(); print((time.perf_counter() - t0) / %d)\n" % (REPO, B, args.mpc_hzn, n, args.mpc_hzn, n))
        try:
            r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, F16_MPC_WAVE_QUEUE="0"), capture_output=True, text=True, timeout=300)
            dhw = float(r.stdout.strip().splitlines()[-1])
            res["dispatch"]["value_workgroup_per_aircraft_dealt_by_the_hardware"] = B / dhw
            res["dispatch"]["ms_per_batch_workgroup_per_aircraft"] = dhw * 1e3
        except Exception as e:      # (a diagnostic leg: never fails the bench)
            res["dispatch"]["value_workgroup_per_aircraft_dealt_by_the_hardware"] = None
            res["dispatch"]["hardware_dispatch_leg_error"] = repr(e)[:200]
    res["first_call_value"] = world * B / dfirst
    res["repeated_call_value"], res["repeated_call_ms_per_batch"] = res["value"], res["ms_per_batch"]
    res["value"], res["ms_per_batch"] = res["first_call_value"], dfirst * 1e3      # the headline IS config 4 as written: one call on an unseen batch
    res["call_pattern_note"] = ("`value` = `first_call_value`: BASELINE config 4 as written is ONE calc_MPC_action per aircraft on an unseen "
                                "batch (no history: the workgroups are ordered by ||q||_inf of the QPs just built; `dispatch.value_in_caller_order`: "
                                "no ordering at all); `repeated_call_value`: the same batch called again (workgroups ordered longest-first by the "
                                "previous call's iteration counts: the host closed loop's pattern; rounds 1-4 reported this one as `value`)")
    iss = recorded_issue("k_mpc_wave", res["repeated_call_ms_per_batch"] * 1e-3, batch=B, hzn=args.mpc_hzn)
    if iss:
        iss["note"] = "seconds = build + solve of this run; " + iss["note"]
        res["roofline"]["issue"] = iss
    # linearise + ZOH + LQR chain (BASELINE config 3)
    env._calc_LQR_gain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        env._calc_LQR_gain()
    barrier()
    dl = fdist.max_over_ranks(time.perf_counter() - t0, dev) / n
    res["linearise_zoh_lqr_per_s"] = world * B / dl
    res["solver"] = ("one wavefront per aircraft (k_mpc_wave: equilibrated solves, N <= 30); F16_MPC_WAVE=0 selects the 512-lane "
                     "workgroup per aircraft (k_mpc_fast)") if os.environ.get("F16_MPC_WAVE", "1") != "0" else "512-lane workgroup per aircraft (k_mpc_fast)"
    res["parity_note"] = ("the reference delegates this solve to the `osqp` package, which cannot run in this pipeline: the solver "
                          "restates OSQP's published algorithm with its default settings (deterministic rho interval 100) and is pinned "
                          "to the unique minimiser and to a CPU twin of the same rules -- throughput and iteration counts are "
                          "properties of this restatement (DESIGN.md 2)")
    if rec:
        res["mfma"] = {k: rec[k] for k in rec if k not in ("batch", "hzn")}
        res["mfma"]["source"] = "recorded: profiles/mfma_mpc.json (" + str(rec.get("source", "rocprofv3 --pmc")) + ")"
    return res


def bench_trim(dev, B=4096):
    """straight-and-level trim of B flight conditions in one call (the reference's Nelder-Mead,
    env.py:198-292); the reference takes 0.56 s for one condition."""
    import numpy as np
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
    nfev = info["nfev"].cpu().numpy()
    st = info["status"].cpu().numpy()
    conv = (st & 64) == 0                     # the rest cannot be trimmed: scipy, too, runs them to maxiter = 50,000 (env.py:273)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    F16Batch.trim(h[conv], v[conv], device=dev)
    torch.cuda.synchronize()
    dtc = time.perf_counter() - t0
    return {"conditions": B, "ms": dt * 1e3, "trims_per_s": B / dt, "cost_median": float(np.median(cost)),
            "cost_max": float(cost.max()), "nfev_mean": float(nfev.mean()), "nfev_max": int(nfev.max()),
            "not_converged": int((~conv).sum()), "iterations_max": int(info["iters"].max()),
            "trimmable_only": {"conditions": int(conv.sum()), "ms": dtc * 1e3, "nfev_max": int(nfev[conv].max())},
            "reference_s_per_trim": 0.56,
            "note": "sixteen lanes per condition evaluate every candidate of a Nelder-Mead iteration at once: one plant "
                    "evaluation of latency per iteration; a condition whose iteration reaches a fixed point (the ones that cannot be "
                    "trimmed: they would repeat it to the reference's maxiter = 50,000) is accounted for instead of run -- same "
                    "results bit for bit (tests)"}


def bench_hzn_sweep(dev, B=64, max_hzn=150):
    """The reference's horizon sweep (env.py:426-436: calc_MPC_action(0, 0, 0, N) for N = 1..150 on the same states), for B
    aircraft, as one library call (f16_mpc_hzn_sweep); first call on these states (no scheduling history is used)."""
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config4_states
    x0, u0 = config4_states(B)
    env = F16Batch(x0, u0, xcg=0.35, device=dev)
    env.build_ssr()
    env._calc_MPC_action(0, 0, 0, 33)                      # (library warm-up: kernel attributes, pool)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sw, inf = env._calc_constr_checking_hzn(max_hzn=max_hzn, return_info=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = inf["iters"]
    t0 = time.perf_counter()
    sw2 = env._calc_constr_checking_hzn(max_hzn=max_hzn)   # the same sweep again: its queue is ordered by the first one's counts
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    assert torch.equal(torch.nan_to_num(sw, nan=1e300), torch.nan_to_num(sw2, nan=1e300))
    # algorithmic bytes of the long horizons' solve launch: every iteration streams the padded half of the KKT inverse once
    # ((n^2 / 2 + ~32 n) doubles, n = 3 N: csrc/f16_mpc_big.hip hoff); factorisations, set-up and the vectors are not counted
    def half_bytes(N):
        n, q = 3 * N, (3 * N) >> 6
        return 8.0 * (2048 * q * (q + 1) + (n - 64 * q) * (q + 1) * 64 + 64)
    itn = it.sum(dim=1).cpu().numpy()                      # iterations per horizon, all aircraft
    stream = float(sum(itn[N - 1] * half_bytes(N) for N in range(33, max_hzn + 1)))
    roof = {"bound": "Infinity-Cache stream (NOT an HBM roofline)", "kernel": "k_mpc_big (one launch over the pairs N = 33..%d)" % max_hzn,
            "achieved": stream / dt2 / 1e9, "peak": None, "unit": "GB/s", "frac": None, "bytes_per_launch": stream, "traffic": None,
            "note": "achieved = iterations x padded-half bytes of the KKT inverse over the whole REPEATED call; the 256 inverses in flight "
                    "(<= 0.93 MB each, 238 MB) sit in the 256 MB Infinity Cache, so this is a cache-stream rate set by what ONE CU can take "
                    "in (tools/micro/cu_stream.hip), not a fraction of HBM bandwidth; counters: profiles/r03_sweep.json"}
    return {"aircraft": B, "max_hzn": max_hzn, "seconds": dt, "seconds_repeated_call": dt2, "solves": int(it.numel()),
            "solves_per_s": it.numel() / dt, "stream": roof,
            "aircraft_iterations": float(it.sum()), "iterations_max": int(it.max()),
            "certified_infeasible": int((inf["status"] & 128).ne(0).sum()), "settings": "osqp_defaults",
            "note": "horizons <= 32 one call after the other; 33..150: one build launch per horizon, then ONE launch of the "
                    "long-horizon solver over every (horizon, aircraft) pair taken from a work queue, longest horizons first; the "
                    "call lasts as long as its slowest pair started late (here N = 103 at max_iter = 40,000); with a launch per horizon: 61 s; "
                    "seconds_repeated_call: the queue ordered costliest-first by the iteration counts of the previous sweep on this "
                    "stream (scheduling history only, same results)"}


def bench_closed_loop(args, dev, rank, world, fdist, barrier):
    """BASELINE config 5 per GPU: B = 8192 aircraft, closed loop (calc_MPC_action N = 30, then one Euler step), T = 100
    steps with the trajectory [T,18,B] kept on the device, then the ONE data-path collective: the all-gather of the
    shards, timed on its own.  The model is frozen at construction as in the reference (env.py:49-60), so
    the model-only part of the QP is a prepared plan; shorter legs measure the variants beside it."""
    import torch
    from f16_mpc_oop_py_amd import F16Batch
    from f16_mpc_oop_py_amd.workload import config4_states
    B, T = args.config5_batch, args.config5_steps
    x0, u0 = config4_states(B * world)
    sl = slice(rank * B, (rank + 1) * B)
    res = {}
    short = max(4, T // 5)

    def leg(name, use_plan, steps, keep_traj=False, fused=False, hold=False, one_lane=False):
        env = F16Batch(x0[sl], u0[sl], xcg=0.35, device=dev)
        env.build_ssr()
        if use_plan:
            st = dict(check_every=5) if name.endswith("check5") else {}
            if "builder_rule" in name:
                st.update(F16Batch.solver_modes()["builder_rule"])
            env.prepare_MPC(args.mpc_hzn, settings=st or None, warm_start="warm_start" in name)
        kw = dict(hzn=args.mpc_hzn, gather=False, use_plan=use_plan, fused=fused, hold_command=hold, one_lane=one_lane)
        fdist.closed_loop_mpc_rollout(env, steps=2, stats={}, **kw)    # (stats: the reductions' first launch too)
        env.reset()
        stats = {}
        barrier()
        t0 = time.perf_counter()
        traj = fdist.closed_loop_mpc_rollout(env, steps=steps, stats=stats, **kw)
        barrier()
        dt = fdist.max_over_ranks(time.perf_counter() - t0, dev)
        # Per-aircraft conditions of this synthetic batch (include/f16_hip.h, f16_rollout_mpc): a QP that OSQP's rules certify
        # infeasible returns a NaN command (bit 128); the reference's actuator models propagate it (np.clip), the surface states turn
        # NaN (bit 32) and that aircraft is not solved for any more -- unless the previous command is held (`hold`).  Everybody else
        # must be finite and unflagged.
        st_ = env.status
        flagged = (st_ & (32 | 128)) != 0
        assert int((st_[~flagged] & ~64).abs().max()) == 0 if bool((~flagged).any()) else True     # (a NaN surface state also reads as "off the grid": bits 1..8 of flagged aircraft)
        assert bool(torch.isfinite(traj[:, :, ~flagged]).all())
        if hold:
            assert bool(torch.isfinite(traj).all())
        r = {"aircraft_steps_per_s": world * B * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
             "iters_mean": stats["iters_mean"], "longest_solve_iters_mean_over_steps": stats["iters_max_mean"],
             "aircraft_infeasible_at_some_step": int(fdist.sum_over_ranks(float(((st_ & 128) != 0).sum()), dev)),
             "aircraft_not_finite_at_the_end": int(fdist.sum_over_ranks(float(((st_ & 32) != 0).sum()), dev)),
             "aircraft_hit_max_iter": int(fdist.sum_over_ranks(float(((st_ & 64) != 0).sum()), dev))}
        return (r, traj, dt) if keep_traj else (r, None, dt)

    # every variant over the SAME number of steps (the first steps of a closed loop need the most iterations: legs of different
    # length are not comparable)
    for name, use_plan in (("prepared_plan", True), ("one_shot", False), ("prepared_plan_warm_start", True),
                           ("prepared_plan_warm_start_check5", True), ("prepared_plan_builder_rule", True),
                           ("prepared_plan_builder_rule_warm_start", True)):
        res[name] = leg(name, use_plan, short)[0]
    res["fused"] = leg("fused", True, short, fused=True)[0]
    res["fused_warm_start"] = leg("fused_warm_start", True, short, fused=True)[0]      # (opt-in: step t starts from the solution of step t - 1)
    # the headline: config 5 as written -- reference settings (OSQP defaults, cold start per solve), all T steps -- through the ONE-
    # launch closed loop (f16_rollout_mpc); the host loop (six launches per step, a join after every solve) over the same T beside it
    r, traj, dt = leg("fused", True, T, keep_traj=True, fused=True)
    res["headline_leg"] = "fused (f16_rollout_mpc: one launch, (step, aircraft) pairs from a work queue)"
    res["headline"] = r
    # the host loop over the same T steps, stepping with the one-lane kernel the fused loop steps with (F16_FLAG_ONE_LANE): the two
    # trajectories must then be IDENTICAL, bit for bit, at full size -- every aircraft, every step, NaN patterns included
    rh, trajh, _ = leg("prepared_plan", True, T, keep_traj=True, one_lane=True)
    res["host_loop"] = rh
    same = bool(torch.equal(torch.nan_to_num(traj, nan=1e300), torch.nan_to_num(trajh, nan=1e300)))
    res["one_launch_equals_host_loop_bit_for_bit"] = same
    assert same, "f16_rollout_mpc and the host loop disagree"
    del trajh
    res["fused_hold_command"] = leg("fused_hold", True, T, fused=True, hold=True)[0]
    barrier()
    t0 = time.perf_counter()
    full = fdist.all_gather_trajectories(traj, layout="ranks")
    barrier()
    tg = fdist.max_over_ranks(time.perf_counter() - t0, dev)
    res["allgather"] = {"ms": tg * 1e3, "bytes_received_per_gpu": int(full.numel() * 8) if fdist.group_active() else 0,
                        "shape": list(full.shape), "world": fdist.world_size()}
    res["aircraft_steps_per_s"] = world * B * T / dt
    res["aircraft_steps_per_s_including_collation"] = world * B * T / (dt + tg)
    del full, traj
    res["batch_per_gpu"] = B
    res["steps"] = T
    res["hzn"] = args.mpc_hzn
    res["short_leg_steps"] = short
    res["note"] = ("`headline` = config 5 as written (reference settings, cold start, all `steps`) as ONE launch; `host_loop` = the same loop "
                   "as six launches per step with a join after every solve (a step then ends with its longest solve: "
                   "`longest_solve_iters_mean_over_steps`); `fused_hold_command` = the one-launch loop with F16_FLAG_HOLD_COMMAND (an "
                   "aircraft whose QP is certified infeasible keeps its previous command and is solved for at every later step; by "
                   "default it receives OSQP's NaN, its surface states turn NaN as in the reference, and its later solves are skipped: "
                   "`aircraft_not_finite_at_the_end` -- their steps still count as steps).  The other variants run `short_leg_steps`: "
                   "prepared_plan / one_shot / fused start every solve cold, as the reference does (a new OSQP object per call; with OSQP's "
                   "defaults a plan saves the QP build only -- the equilibration depends on q); "
                   "warm_start is the opt-in extension (OSQP's in-object default; `fused_warm_start`: the same inside the one-launch loop, the solution handed "
                   "from wavefront to wavefront with the state); check5 = the same with the termination test "
                   "every 5 iterations instead of OSQP's 25 (a warm-started solve needs fewer than 25); builder_rule = the opt-in "
                   "solver settings (no equilibration, start value of rho from the traces; KKT factorisation cached in the plan)")
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


def cpu_baseline_mpc(hzn):
    """the MPC CPU baseline is the C restatement of the same chain (oracle/f16_mpc_oracle.c, kind
    'port': linearise + ZOH + DARE + setup_OSQP + OSQP-style ADMM with the reference's implied settings) on the same
    config-4 flight conditions, single thread and all cores -- OSQP itself cannot be timed anywhere in this pipeline
    (not installable offline)."""
    import numpy as np
    from oracle import mpc_oracle as mo
    from f16_mpc_oop_py_amd.workload import config4_states
    ora = mo.COracle()
    cores = min(len(os.sched_getaffinity(0)), 64)
    n1, nall = 48, 48 * cores
    x0, _ = config4_states(max(nall, n1))
    t0 = time.perf_counter()
    r1 = ora.mpc_batch(x0[:n1], hzn, xcg=0.35, nthreads=1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    ra = ora.mpc_batch(x0[:nall], hzn, xcg=0.35, nthreads=cores)
    ta = time.perf_counter() - t0
    return {"value": nall / ta, "unit": "solves/s", "cores": cores, "kind": "port", "single_thread_value": n1 / t1,
            "admm_iters_mean": float(np.mean(ra["iters"])),
            "sample": f"first {nall} config-4 aircraft, N={hzn}, on {cores} threads (single-thread figure: first {n1}): "
                      f"C linearise + ZOH + DARE + dense setup_OSQP + Cholesky ADMM, settings = the GPU headline's"}


def config1_reference_style_loop(steps=10000):
    """BASELINE config 1 (1 aircraft, lofi, open-loop Euler through the reference-style caller): the
    reference's `step` (env.py:105-130) written out against a ctypes handle exposing `Nlplant` / `atmos`
    (env.py:65-103: four actuator models, upd_lef through atmos(), Nlplant, Euler) for 10,000 steps -- once with the handle
    = the C restatement on the CPU (what the reference's own .so does), once with the handle = libf16hip.so's drop-in
    symbols (one aircraft per call on the GPU).  us per step, lower is better; not the product's use case."""
    import ctypes
    import numpy as np
    from f16_mpc_oop_py_amd import lib
    from f16_mpc_oop_py_amd import parameters as P
    from oracle import mpc_oracle as mo
    L = lib.load()
    L.f16_dropin_config(0.25, 0)
    ora = mo.COracle()                              # (runs f16o_init)
    ora.lib.f16o_set_xcg(0.25)
    # plain CDLL handles, as the reference makes them (parameters.py:108-114): no argtypes, c_double / c_void_p objects
    h_cpu = ctypes.CDLL(os.path.join(REPO, "oracle", "libf16_oracle.so"))
    h_gpu = ctypes.CDLL(lib.SO_PATH)
    vp, cd, ci = ctypes.c_void_p, ctypes.c_double, ctypes.c_int
    x_lb, x_ub = np.array(P.x_lb, dtype=float), np.array(P.x_ub, dtype=float)

    def loop(handle, nsteps):
        x = np.array(P.x0, dtype=np.float64)
        u = np.array(P.u0, dtype=np.float64)
        xdot = np.zeros(18)
        coeff = np.zeros(3)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            if ((x < x_lb) | (x > x_ub)).any():                                     # env.py:117-124
                break
            # env.py:90-98 (utils.py:289-330)
            T_dot = np.clip(np.clip(u[0], 1000, 19000) - x[12], -10000, 10000)
            dh_dot = np.clip(20.2 * (np.clip(u[1], -25, 25) - x[13]), -60, 60)
            da_dot = np.clip(20.2 * (np.clip(u[2], -21.5, 21.5) - x[14]), -80, 80)
            dr_dot = np.clip(20.2 * (np.clip(u[3], -30, 30) - x[15]), -120, 120)
            handle.atmos(cd(x[2]), cd(x[6]), vp(coeff.ctypes.data))                  # utils.py:291
            alpha_deg = x[7] * 180 / np.pi
            LF_err = alpha_deg - (x[17] + 2 * alpha_deg)
            lef_cmd = np.clip((x[17] + 2 * alpha_deg) * 1.38 + 1.45 - coeff[1] / coeff[2] * 9.05, 0, 25)
            lf2_dot = np.clip((1 / 0.136) * (lef_cmd - x[16]), -25, 25)
            handle.Nlplant(vp(x.ctypes.data), vp(xdot.ctypes.data), ci(0))          # env.py:100, fi_flag = 0
            xdot[12:18] = (T_dot, dh_dot, da_dot, dr_dot, lf2_dot, LF_err * 7.25)   # env.py:102
            x += xdot * 0.001                                                        # env.py:126
        return (time.perf_counter() - t0) / nsteps * 1e6, x

    cpu_us, xc = loop(h_cpu, steps)
    gpu_us, xg = loop(h_gpu, steps)
    return {"steps": steps, "fidelity": "lofi", "us_per_step_cpu_restatement_handle": cpu_us,
            "us_per_step_dropin_gpu_handle": gpu_us, "unit": "us/step",
            "final_state_max_rel_diff": float(np.max(np.abs(xc - xg) / np.maximum(1.0, np.abs(xc)))),
            "reference_us_per_step_measured_in_build_container": "169-188 (BASELINE.md, hifi)"}


if __name__ == "__main__":
    main()
