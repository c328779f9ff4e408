"""One-process-per-GPU launch plumbing (torch.distributed; backend "nccl" is RCCL
on ROCm, "gloo" for the CPU tests).

The inference hot path shards independent images across ranks with no
data-path collective ("weak" scaling); the only collectives are the barrier and
the max-over-ranks reduction of the elapsed time that bench.py reports.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the process group described by RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # HRV_DIST_BACKEND=gloo: two ranks sharing one GPU in the DP tests (RCCL refuses duplicate devices)
            backend = os.environ.get("HRV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier_sync(device_sync: Optional[Callable[[], None]] = None):
    if device_sync is not None:
        device_sync()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device_sync is not None:
        device_sync()


def timed_steps(step: Callable[[int], None], steps: int, warmup: int,
                device_sync: Optional[Callable[[], None]] = None) -> float:
    """Run ``warmup`` untimed + exactly ``steps`` timed calls of ``step(i)``, bracketed by
    barrier + device sync on both sides; returns the MAX over ranks of the elapsed seconds."""
    for i in range(warmup):
        step(i)
    barrier_sync(device_sync)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    barrier_sync(device_sync)
    dt = time.perf_counter() - t0
    if dist.is_available() and dist.is_initialized():
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def shard_seed(base: int, rank: int) -> int:
    """Per-rank synthetic-data seed (seed = 1234 + rank)."""
    return base + rank
