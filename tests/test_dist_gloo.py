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
    got = fdist.all_gather_trajectories(full[:, :, lo:hi].contiguous())
    ok = bool(torch.equal(got, full))
    mx = fdist.max_over_ranks(1.0 + r)
    sm = fdist.sum_over_ranks(10.0 * (r + 1))
    st = torch.zeros(4, dtype=torch.int32)
    st[r] = 16 if r == 0 else 128
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
        assert ok and mx == 2.0 and sm == 30.0 and orv == (16 | 128)


def test_single_process_paths_need_no_group():
    t = torch.zeros(3, 18, 4, dtype=torch.float64)
    assert fdist.all_gather_trajectories(t) is t
    assert fdist.max_over_ranks(3.5) == 3.5 and fdist.world_size() == 1
