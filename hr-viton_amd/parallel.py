"""Data-parallel gradient synchronisation: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), replicas hold identical weights, gradients are summed with bucketed
all-reduces that are launched from inside the hand-written backward plan as soon as a bucket's
last gradient has been produced, so the collectives overlap the remaining backward kernels.
Replaces the reference's single-process nn.DataParallel wrapper
(sync_batchnorm/replicate.py:50-67; train_generator.py:171-178).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are large (default 64 MiB) so each
ring/direct all-reduce is bandwidth- rather than latency-bound; the 402 MB of generator gradients
go out as ~7 buckets in reverse-forward order (conv_img / up_4 first, head_0 last)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, params, bucket_mb: float = 64.0, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.enabled = True
        cap = int(bucket_mb * (1 << 20) / 4)
        # backward produces gradients roughly in reverse registration order
        order = list(reversed(self.params))
        self.buckets: List[dict] = []
        cur, cur_n = [], 0
        for p in order:
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(dict(params=cur, n=cur_n))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(dict(params=cur, n=cur_n))
        self.where: Dict[torch.nn.Parameter, tuple] = {}
        for bi, b in enumerate(self.buckets):
            dev = b["params"][0].device
            b["flat"] = torch.zeros(b["n"], dtype=torch.float32, device=dev)
            off = 0
            for p in b["params"]:
                self.where[p] = (bi, off, p.numel())
                off += p.numel()
            b["pending"] = set(b["params"])
            b["handle"] = None
        self.begin()

    def begin(self):
        """Start of a backward pass: all buckets empty."""
        for b in self.buckets:
            b["pending"] = set(b["params"])
            b["handle"] = None
            b["seen"] = set()

    def grad_of(self, p) -> Optional[torch.Tensor]:
        bi, off, k = self.where[p]
        b = self.buckets[bi]
        if p not in b["seen"]:
            return None
        return b["flat"][off:off + k].view_as(p)

    def on_grad(self, p, g: torch.Tensor):
        """Called by the backward plan when the gradient of ``p`` is final."""
        if not self.enabled or p not in self.where:
            return
        bi, off, k = self.where[p]
        b = self.buckets[bi]
        b["flat"][off:off + k].copy_(g.reshape(-1))
        b["seen"].add(p)
        b["pending"].discard(p)
        if not b["pending"] and b["handle"] is None:
            self._fire(b)

    def _fire(self, b):
        if self.world > 1:
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            b["handle"] = True

    def wait(self):
        """End of backward: flush buckets whose parameters got no gradient (unused parameters, e.g.
        conv_7 in 'more' mode: zeros keep the replicas in lock-step), then wait for the collectives."""
        if not self.enabled:
            return
        for b in self.buckets:
            if b["handle"] is None:
                for p in list(b["pending"]):
                    bi, off, k = self.where[p]
                    b["flat"][off:off + k].zero_()
                b["pending"] = set()
                self._fire(b)
        for b in self.buckets:
            h = b["handle"]
            if h is not None and h is not True:
                h.wait()


def broadcast_module(module: torch.nn.Module, src: int = 0, process_group=None):
    """Initial replica synchronisation (parameters and buffers, incl. the spectral-norm u, v)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
