"""Multi-GPU layer: one process per GPU, aircraft sharded contiguously, ONE collective.

The reference is single-process; aircraft never interact, so the batch is partitioned
contiguously -- rank g owns aircraft [g*B/W, (g+1)*B/W) -- tables and constants are replicated, and there is
no exchange inside the time loop.  The only data-path collective is the all-gather that collates the per-rank
trajectory shards [T,18,B/W] (RCCL `ncclAllGather` over xGMI through torch.distributed's "nccl" backend; "gloo"
on CPU tensors for the host-logic tests).  Scalars (max-over-ranks time, OR of status words) use all-reduce.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(total, world, rank):
    """Contiguous shard [lo, hi) of `total` aircraft for `rank` of `world`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  World size 1 needs no process group; F16_DIST_FORCE_GROUP=1 forms one anyway
    (MASTER_ADDR / MASTER_PORT default to 127.0.0.1 / a free port): every collective of this module then really runs
    through the backend -- the way to execute the RCCL path on a one-GPU box (tests/test_gpu_dist_rccl.py,
    `bench.py --force-group`)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    force = os.environ.get("F16_DIST_FORCE_GROUP") == "1"
    if world == 1 and force and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool
        # F16_DIST_BACKEND=gloo: rehearsal of the multi-rank paths on a box with fewer GPUs than ranks
        backend = backend or os.environ.get("F16_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # bounded collectives: a rank that never arrives makes the others fail after F16_DIST_TIMEOUT seconds (default 300)
        # instead of holding the run until an external limit (torch's defaults are 10 / 30 minutes)
        import datetime
        kw = {"timeout": datetime.timedelta(seconds=float(os.environ.get("F16_DIST_TIMEOUT", "300")))}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def group_active():
    """A process group exists (world size 1 included, F16_DIST_FORCE_GROUP): the collectives go through the backend."""
    return dist.is_available() and dist.is_initialized()


def all_gather_trajectories(traj_local, total=None, layout="flat", chunk_bytes=1 << 28, algo="collective"):
    """Collate the per-rank trajectory shards traj_local [T,18,Bl] (equal Bl on every rank; global aircraft
    g = rank*Bl + b) on every rank.

    layout="ranks": ONE all-gather of the contiguous shard into a [W,T,18,Bl] receive buffer, returned as the strided
        VIEW [T,18,W,Bl] (`buf.permute(1,2,0,3)`) -- no second copy (at config 5: 9.4 GB received per GPU, nothing else).
    layout="flat" (default): the reference-shaped [T,18,W*Bl] array, aircraft in global order.  The final layout
        interleaves the ranks at a granularity of Bl doubles, which no all-gather can write directly, so the shard is
        gathered in chunks of time samples (<= chunk_bytes received per chunk) and each chunk is scattered straight
        into its place in the result: peak extra memory is one chunk, not a second copy of the whole trajectory.
    `total` trims padded aircraft off the end (flat layout only).
    algo="direct" (layout="ranks" only): every rank sends its shard straight to every peer and receives theirs
        (W - 1 sends + W - 1 receives posted as ONE batch of point-to-point operations) -- the one-peer-per-link pattern
        SURVEY.md 8(e) describes for the xGMI mesh (7 links per GPU), beside the library's all-gather ("collective").  Same
        result; `bench.py` times both on a multi-GPU run."""
    W = world_size()
    if not group_active():
        return traj_local if layout == "flat" else traj_local.unsqueeze(2)
    T, K, Bl = traj_local.shape
    src = traj_local.contiguous()
    if algo not in ("collective", "direct") or (algo == "direct" and layout != "ranks"):
        raise ValueError("algo must be 'collective' or 'direct' (direct: layout='ranks' only)")
    if layout == "ranks":
        recv = torch.empty((W, T, K, Bl), dtype=src.dtype, device=src.device)
        if algo == "direct":
            me = dist.get_rank()
            recv[me].copy_(src)
            ops = []
            for d in range(1, W):                                         # peer order staggered by rank: no hot receiver
                to, frm = (me + d) % W, (me - d) % W
                ops.append(dist.P2POp(dist.isend, src, to))
                ops.append(dist.P2POp(dist.irecv, recv[frm], frm))
            for r in (dist.batch_isend_irecv(ops) if ops else []):
                r.wait()
            return recv.permute(1, 2, 0, 3)
        dist.all_gather_into_tensor(recv.view(-1), src.view(-1))          # flat: valid for RCCL and gloo
        return recv.permute(1, 2, 0, 3)
    if layout != "flat":
        raise ValueError("layout must be 'flat' or 'ranks'")
    out = torch.empty((T, K, W * Bl), dtype=src.dtype, device=src.device)
    out4 = out.view(T, K, W, Bl)
    tc = max(1, min(T, int(chunk_bytes // max(1, W * K * Bl * src.element_size()))))
    recv = torch.empty((W, tc, K, Bl), dtype=src.dtype, device=src.device)
    for t0 in range(0, T, tc):
        n = min(tc, T - t0)
        r = recv[:, :n] if n == tc else torch.empty((W, n, K, Bl), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(r.view(-1), src[t0:t0 + n].reshape(-1))
        out4[t0:t0 + n].copy_(r.permute(1, 2, 0, 3))
    return out if total is None else out[..., :total]


def max_over_ranks(value, device=None):
    """max of a Python float over all ranks (the bench's timing rule)."""
    if not group_active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not group_active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def or_status(status):
    """Bitwise OR of the status words of every aircraft of every rank (int): all 26 bits -- the eight condition bits and the
    F16_ST_ENV_STATE(k) which-state bits 8..25.  The OR over the shard runs where the tensor lives; across ranks the word goes as 32
    0 / 1 flags through an all-reduce(MAX) (RCCL has no bitwise-or reduction)."""
    s = status.detach().reshape(-1).to(torch.int32)
    bits = torch.arange(32, dtype=torch.int32, device=s.device)
    flags = ((s.unsqueeze(1) >> bits) & 1).amax(0).to(torch.int32) if s.numel() else torch.zeros(32, dtype=torch.int32, device=s.device)
    if group_active():
        flags = flags.to("cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    return int(sum(int(b) << k for k, b in enumerate(flags.tolist())))


def closed_loop_mpc_rollout(env, steps, hzn, p_dem=0.0, q_dem=0.0, r_dem=0.0, traj_every=1, gather=True, use_plan=True, stats=None,
                            fused=None, hold_command=False, one_lane=False):
    """BASELINE config 5 / test_env.py:480-495 pattern on this rank's shard, then one all-gather:
    per step  cmd = calc_MPC_action(p,q,r,hzn); u.values[1:] = cmd; step(u.values).
    fused=True: the whole loop of the shard as ONE launch (F16Batch.rollout_MPC / C-ABI f16_rollout_mpc: (step, aircraft) pairs from
    a work queue, no join per step); fused=False: the host loop below, its checker (bit-identical with one_lane=True).  Default (None):
    the one launch wherever it applies -- a prepared plan (use_plan) with equilibrated solves (OSQP's defaults), hzn <= 30 -- else the host loop.
    use_plan: the model is frozen (env.py:49-60), so the model-only part of the QP is prepared once
    (F16Batch.prepare_MPC) -- same commands bit for bit.  With the reference's solver settings (OSQP defaults) a plan saves
    the QP build only: the equilibration looks at q, i.e. at the state of the call, so it and the factorisation are redone
    per solve (bench.py reports both legs at equal length); with the opt-in rule the factorisation is cached as well.
    (A HIP-graph replay of the step was measured and is SLOWER than the six eager launches on ROCm 7.2: 0.50 vs
    0.19 ms per step at B = 256, 3.36 vs 3.15 ms at B = 8192 -- the step is kept capture-safe but launched eagerly.)
    hold_command: a step whose QP is infeasible (NaN command, as OSQP returns it) keeps the previous command instead of writing the
    NaN into u.values (lib.F16_FLAG_HOLD_COMMAND; default: the reference's behaviour -- NaN command -> NaN actuator states).
    one_lane: step with the one-lane-per-aircraft rollout kernel whatever the shard size (lib.F16_FLAG_ONE_LANE: the fused kernel's step).
    stats (dict, optional): receives "iters_mean" = mean ADMM iterations per solve over the whole loop, "iters_max_mean" = the mean over
    the steps of the LONGEST solve of the step (device-side reductions, read once at the end; the host loop cannot end a step before
    its longest solve) and "flagged_per_step" = per step the number of aircraft whose solve raised F16_ST_QP_INFEASIBLE / QP_MAXITER /
    NONFINITE (host loop only).
    Returns the collated trajectory [steps//traj_every, 18, B_total] (or the local shard if gather=False)."""
    from . import lib as _lib
    T = steps // traj_every
    if fused is None:
        plan_ok = getattr(env, "_plan", None) is None or (env._plan_hzn == int(hzn) and getattr(env, "_plan_default_settings", False))
        fused = bool(use_plan) and int(hzn) <= 30 and plan_ok and not one_lane
    if fused:
        traj, info = env.rollout_MPC(steps, p_dem, q_dem, r_dem, hzn, traj_every=traj_every, return_info=True, hold_command=hold_command)
        if stats is not None:
            its = info["iters"].to(torch.float64)
            stats["iters_mean"] = float(its.mean())
            stats["iters_max_mean"] = float(its.max(1).values.mean())
        return all_gather_trajectories(traj) if gather else traj
    traj = torch.empty((T, 18, env.B), dtype=torch.float64, device=env.device)
    dem = torch.empty((3, env.B), dtype=torch.float64, device=env.device)      # demands on the device once
    for k, v in enumerate((p_dem, q_dem, r_dem)):
        dem[k] = torch.as_tensor(v, dtype=torch.float64, device=env.device)
    it_sum = torch.zeros((), dtype=torch.float64, device=env.device) if stats is not None else None
    it_max = torch.zeros((), dtype=torch.float64, device=env.device) if stats is not None else None
    flagged = torch.zeros((steps, 3), dtype=torch.int64, device=env.device) if stats is not None else None
    flags0 = env.flags
    if one_lane:
        env.flags = env.flags | _lib.F16_FLAG_ONE_LANE
    try:
        for k in range(steps):
            cmd = env._calc_MPC_action(dem, None, None, hzn, use_plan=use_plan)
            if it_sum is not None:
                it_sum += env.last_iters.sum()
                it_max += env.last_iters.max()
                st = env.last_status
                flagged[k, 0] = ((st & _lib.F16_ST["QP_INFEASIBLE"]) != 0).sum()
                flagged[k, 1] = ((st & _lib.F16_ST["QP_MAXITER"]) != 0).sum()
                flagged[k, 2] = ((st & _lib.F16_ST["NONFINITE"]) != 0).sum()
            c = cmd.t()
            env._u[1:4] = torch.where(torch.isnan(c), env._u[1:4], c) if hold_command else c
            env.rollout(1)
            if (k + 1) % traj_every == 0:
                traj[(k + 1) // traj_every - 1] = env._x
    finally:
        env.flags = flags0
    if stats is not None:
        stats["iters_mean"] = float(it_sum) / max(1, steps * env.B)
        stats["iters_max_mean"] = float(it_max) / max(1, steps)
        stats["flagged_per_step"] = flagged.cpu().numpy()
    return all_gather_trajectories(traj) if gather else traj
