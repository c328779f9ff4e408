"""CPU, world_size 2, gloo: the N>1 launch plumbing bench.py uses (env rendezvous on
127.0.0.1, barrier-bracketed timing, MAX-over-ranks reduction, per-rank shards)."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    import torch
    import torch.distributed as dist
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import dist as hdist
    r, lr, w = hdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    calls = []

    def step(i):
        calls.append(i)
        time.sleep(0.02 * (rank + 1))  # rank 1 is the slow one

    dt = hdist.timed_steps(step, steps=5, warmup=2)
    # per-rank shard seeds differ; images processed sum over ranks
    t = torch.tensor([float(len(calls) - 2) * 4])
    dist.all_reduce(t)
    q.put((rank, dt, len(calls), hdist.shard_seed(1234, rank), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_timed_steps_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, dt0, n0, s0, tot0), (r1, dt1, n1, s1, tot1) = res
    assert n0 == n1 == 7                      # 2 warm-up + exactly 5 timed
    assert abs(dt0 - dt1) < 1e-9              # both ranks report the MAX over ranks
    assert dt0 >= 5 * 0.04 * 0.9              # ... which is the slow rank's time
    assert (s0, s1) == (1234, 1235)
    assert tot0 == tot1 == 40.0               # 2 ranks x 5 steps x batch 4
