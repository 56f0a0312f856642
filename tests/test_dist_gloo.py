"""N>1 host logic on CPU: world_size-2 gloo processes exercise the sharding + the single all-gather + the
max-over-ranks timing rule exactly as bench.py / dist.py use them on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from f16_mpc_oop_py_amd import dist as fdist


def test_shard_bounds_partition_the_batch():
    for total in (0, 1, 7, 4096, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [fdist.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    try:
        _worker_body(q)
    except Exception as e:      # surface the failure instead of a queue timeout
        q.put((rank, False, repr(e), 0.0, 0))
        raise


def _worker_body(q):
    r, w, _ = fdist.init_from_env("gloo")
    T, Bt = 5, 12
    full = torch.arange(T * 18 * Bt, dtype=torch.float64).reshape(T, 18, Bt)     # global trajectory, aircraft-major last
    lo, hi = fdist.shard_bounds(Bt, w, r)
    mine = full[:, :, lo:hi].contiguous()
    got = fdist.all_gather_trajectories(mine)
    ok = bool(torch.equal(got, full))
    # chunked collation (two time samples per chunk + a ragged last chunk) and the copy-free [T,18,W,Bl] view
    ok = ok and bool(torch.equal(fdist.all_gather_trajectories(mine, chunk_bytes=2 * 18 * Bt * 8), full))
    view = fdist.all_gather_trajectories(mine, layout="ranks")
    ok = ok and tuple(view.shape) == (T, 18, w, Bt // w) and not view.is_contiguous()
    ok = ok and bool(torch.equal(view.reshape(T, 18, Bt), full))
    direct = fdist.all_gather_trajectories(mine, layout="ranks", algo="direct")      # W - 1 sends / receives per rank
    ok = ok and tuple(direct.shape) == (T, 18, w, Bt // w) and bool(torch.equal(direct.reshape(T, 18, Bt), full))
    ok = ok and bool(torch.equal(fdist.all_gather_trajectories(mine, total=Bt - 1), full[..., :Bt - 1]))
    mx = fdist.max_over_ranks(1.0 + r)
    sm = fdist.sum_over_ranks(10.0 * (r + 1))
    st = torch.zeros(4, dtype=torch.int32)
    st[r] = (16 | (1 << (8 + 13))) if r == 0 else (128 | (1 << (8 + 6)))      # an F16_ST_ENV_STATE(k) bit on each rank
    orv = fdist.or_status(st)
    q.put((r, ok, mx, sm, orv))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_collates_shards_in_global_order_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for r, ok, mx, sm, orv in res:
        assert ok and mx == 2.0 and sm == 30.0 and orv == (16 | 128 | (1 << 21) | (1 << 14))


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher (RANK unset): the parent spawns two fresh rank processes, relays rank
    0's JSON line and its exit status.  --dry-run keeps the kernels out so the launch path runs on a box without GPUs;
    the same command line without --dry-run is what the driver's SCALE run uses."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F16_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["requested_gpus"] == 2 and rec["allgather_ok"] and rec["backend"] == "gloo"
    assert rec["max_over_ranks"] == 2.0


def test_one_rank_with_a_forced_group_goes_through_the_backend():
    """F16_DIST_FORCE_GROUP=1 (`bench.py --force-group`): one rank forms a real process group, so the all-gather and the
    reductions execute through the backend instead of short-circuiting (gloo here; RCCL in tests/test_gpu_dist_rccl.py)."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F16_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--dry-run", "--force-group"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["allgather_ok"] and rec["backend"] == "gloo" and rec["max_over_ranks"] == 1.0


def test_bench_parent_propagates_a_rank_failure():
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F16_DIST_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_parent_kills_a_stalled_run_inside_its_deadline():
    """A rank that never enters the first collective (here: it sleeps) must not hold the caller: the parent kills the
    children it started when --rank-timeout expires, prints the ranks' stderr tails and exits non-zero."""
    import subprocess
    import sys
    import time
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F16_DIST_BACKEND"] = "gloo"
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--dry-run-stall-rank", "1",
                        "--rank-timeout", "20"], env=env, capture_output=True, text=True, timeout=240)
    dt = time.monotonic() - t0
    assert r.returncode != 0 and dt < 120, (r.returncode, dt)
    assert "rank-timeout" in r.stderr and "rank 1 stderr tail" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_process_paths_need_no_group():
    t = torch.zeros(3, 18, 4, dtype=torch.float64)
    assert fdist.all_gather_trajectories(t) is t
    assert tuple(fdist.all_gather_trajectories(t, layout="ranks").shape) == (3, 18, 1, 4)
    assert fdist.max_over_ranks(3.5) == 3.5 and fdist.world_size() == 1
