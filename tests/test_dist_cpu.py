"""N>1 path on CPU: world_size-2 gloo processes exercise the shard / broadcast / gather code
that the GPU run uses with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mtts import dist as mdist


def test_shard_indices_balanced_and_complete():
    lengths = [5, 100, 7, 50, 60, 1, 99, 3]
    for world in (1, 2, 3, 8):
        shards = mdist.shard_indices(lengths, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lengths)))
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1
    two = mdist.shard_indices(lengths, 2)
    loads = [sum(lengths[i] for i in s) for s in two]
    assert abs(loads[0] - loads[1]) <= 20
    assert mdist.shard_indices([], 4) == [[], [], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weights: only rank 0 holds them
        sd = {"a.weight": torch.arange(12, dtype=torch.float32).reshape(3, 4),
              "b.weight": torch.ones(5, dtype=torch.bfloat16) * 3} if rank == 0 else {}
        got = dict(mdist.broadcast_state_dict(sd, "cpu"))
        assert got["a.weight"].shape == (3, 4) and float(got["a.weight"].sum()) == 66.0
        assert got["b.weight"].dtype == torch.bfloat16 and float(got["b.weight"].float().sum()) == 15.0
        # dialogues: item i "generates" a ramp of length 10*(i+1); item 3 fails
        items = [{"text": "x" * (3 * i + 1)} for i in range(7)]

        def run_local(its, idxs):
            return [None if i == 3 else torch.full((10 * (i + 1),), float(i)) for i in idxs]

        res = mdist.process_batch_sharded(items, run_local)
        if rank == 0:
            assert len(res) == 7 and res[3] is None
            for i, w in enumerate(res):
                if i != 3:
                    assert w.shape == (10 * (i + 1),) and float(w[0]) == float(i)
        else:
            assert res is None
        # metric reduction as bench.py does it
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_broadcast_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def _bench(*args, env=None):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), env=e, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts 2 ranks itself (here: --dry-run = the launcher,
    the rendezvous and the MAX/SUM reductions over gloo, no engine); rank 0 prints ONE line whose n_gpus is 2 and
    whose value is the sum over ranks divided by the slowest rank's time."""
    rc, out, err = _bench("--gpus", "2", "--dry-run", "--steps", "10", "--batch", "32")
    assert rc == 0, err[-2000:]
    assert len(out) == 1
    o = out[0]
    assert o["n_gpus"] == 2 and o["ranks"] == 2 and o["steps"] == 10
    assert abs(o["value"] - 2 * 32 * 10 * 8 / (10 * 4e-3 * 1.01)) < 1e-6 * o["value"]
    rc, out, err = _bench("--dry-run", "--steps", "10")                     # N=1: no launcher, no process group
    assert rc == 0 and out[0]["n_gpus"] == 1 and out[0]["ranks"] == 1


def test_bench_refuses_a_world_that_differs_from_gpus():
    rc, out, err = _bench("--gpus", "4", "--dry-run", env=dict(WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1",
                                                              MASTER_PORT=str(_free_port())))
    assert rc != 0 and not out and "--gpus 4" in err
