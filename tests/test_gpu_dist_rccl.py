"""The RCCL path on the one GPU the builder's box has: a REAL `nccl` process group of world size 1 (F16_DIST_FORCE_GROUP=1; by
default one rank forms no group at all), in a fresh child process, so that the first RCCL execution of dist.py is not the
driver's 8-GPU run: the all-gather of trajectory shards in both layouts and as point-to-point batch, the scalar reductions on
device tensors, the bounded timeout, and `bench.py --gpus 1 --force-group` printing its `allgather` block and the RCCL version."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(F16_DIST_FORCE_GROUP="1", F16_DIST_BACKEND="nccl", F16_DIST_TIMEOUT="120", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=REPO + os.pathsep + env.get("PYTHONPATH", ""))
    env.update(extra)
    return env


def test_world_size_1_nccl_group_runs_every_collective_of_dist_py():
    code = textwrap.dedent("""
        import datetime, json, torch, torch.distributed as dist
        from f16_mpc_oop_py_amd import dist as fdist
        r, w, l = fdist.init_from_env()
        assert dist.is_initialized() and dist.get_backend() == "nccl" and fdist.world_size() == 1 and fdist.group_active()
        dev = torch.device("cuda", l)
        T, B = 7, 96
        full = torch.arange(T * 18 * B, dtype=torch.float64, device=dev).reshape(T, 18, B)
        flat = fdist.all_gather_trajectories(full)
        chunked = fdist.all_gather_trajectories(full, chunk_bytes=3 * 18 * B * 8)          # ragged last chunk
        ranks = fdist.all_gather_trajectories(full, layout="ranks")
        direct = fdist.all_gather_trajectories(full, layout="ranks", algo="direct")
        trimmed = fdist.all_gather_trajectories(full, total=B - 5)
        ok = bool(torch.equal(flat, full)) and flat.data_ptr() != full.data_ptr() and bool(torch.equal(chunked, full))
        ok = ok and tuple(ranks.shape) == (T, 18, 1, B) and bool(torch.equal(ranks.reshape(T, 18, B), full))
        ok = ok and bool(torch.equal(direct.reshape(T, 18, B), full)) and bool(torch.equal(trimmed, full[..., :B - 5]))
        mx = fdist.max_over_ranks(2.5, dev)
        sm = fdist.sum_over_ranks(4.0, dev)
        st = torch.zeros(B, dtype=torch.int32, device=dev)
        st[3], st[40] = 16 | (1 << (8 + 17)), 128
        orv = fdist.or_status(st)
        # a product rollout collated through the same call (shards of a real trajectory)
        from f16_mpc_oop_py_amd import F16Batch
        from f16_mpc_oop_py_amd.workload import config2_states
        x0, u0 = config2_states(256)
        env = F16Batch(x0, u0, device=dev)
        tr = env.rollout(50, traj_every=10)
        got = fdist.all_gather_trajectories(tr)
        ok = ok and bool(torch.equal(got, tr)) and got.data_ptr() != tr.data_ptr()
        torch.cuda.synchronize()
        try:
            tmo = dist.distributed_c10d._get_default_group()._get_backend(dev).options._timeout.total_seconds()
        except Exception:
            tmo = None
        print(json.dumps({"ok": ok, "max": mx, "sum": sm, "or": orv, "rccl": ".".join(map(str, torch.cuda.nccl.version())),
                          "timeout_s": tmo}))
        dist.barrier()
        dist.destroy_process_group()
    """)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["ok"] and rec["max"] == 2.5 and rec["sum"] == 4.0 and rec["or"] == (16 | 128 | (1 << 25))
    assert rec["rccl"] and rec["timeout_s"] in (None, 120.0)  # F16_DIST_TIMEOUT honoured (where torch exposes the option)


def test_bench_gpus_1_with_a_forced_group_prints_the_allgather_block():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--force-group",
                        "--no-cpu", "--no-mpc", "--no-large", "--no-config5"], env=_env(), capture_output=True, text=True,
                       timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["backend"] == "nccl" and rec["versions"]["rccl"]
    ag = rec["allgather"]
    assert ag["bytes_received_per_gpu"] == 1000 * 18 * 4096 * 8 and ag["ms"] > 0
