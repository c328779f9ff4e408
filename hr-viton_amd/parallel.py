"""Data-parallel gradient synchronisation: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), replicas hold identical weights, gradients are summed with bucketed
all-reduces that are launched from inside the hand-written backward plan as soon as a bucket's
last gradient has been produced, so the collectives overlap the remaining backward kernels.
Replaces the reference's single-process nn.DataParallel wrapper
(sync_batchnorm/replicate.py:50-67; train_generator.py:171-178).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are large (default 64 MiB) so each
ring/direct all-reduce is bandwidth- rather than latency-bound; the 402 MB of generator gradients
go out in reverse-forward order (conv_img / up_4 first, head_0 last).  Two refinements for the END of
the backward, where a collective can no longer hide under compute: a parameter of >= ``big_mb`` (the
37.7 MB 3x3 weights of head_0 / G_middle / up_0) is a bucket of its own, fired the moment its weight
gradient exists instead of waiting for its neighbours; and the parameters whose gradients are produced
LAST (the front of the flat buffer) are cut into a small tail bucket (``tail_mb``), so that what is
exposed after the last backward kernel is a few MB, not a 64 MiB bucket."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


class _FakeCollective:
    """Single-GPU stand-in for a bucket's all-reduce (HRV_FAKE_ALLREDUCE=<repeats>): ``repeats`` device-to-device copies
    of the bucket on a side stream, started when the bucket's last gradient exists on the compute stream and joined
    by ``wait()`` -- the stream / event choreography of the RCCL path, so a 1-GPU kernel trace shows how much of the
    collective's duration the remaining backward kernels hide (tools/dp_overlap.sh, profiles/r02_dp_overlap.txt)."""
    stream: Optional[torch.cuda.Stream] = None

    def __init__(self, flat: torch.Tensor, repeats: int):
        if _FakeCollective.stream is None:
            _FakeCollective.stream = torch.cuda.Stream()
        ready = torch.cuda.Event()
        ready.record()                                   # the bucket is final on the compute stream
        self.done = torch.cuda.Event()
        with torch.cuda.stream(_FakeCollective.stream):
            _FakeCollective.stream.wait_event(ready)
            scratch = torch.empty_like(flat)
            for _ in range(repeats):
                scratch.copy_(flat, non_blocking=True)
            self.done.record()
        self.scratch = scratch

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)


class GradSync:
    def __init__(self, params, bucket_mb: float = 64.0, process_group=None, flat: Optional[torch.Tensor] = None,
                 spans=None, big_mb: float = 16.0, tail_mb: float = 8.0):
        """``flat`` / ``spans`` ([(param, offset, numel)] in buffer order): reduce IN PLACE on the fused optimizer's
        flat gradient buffer (optim.Adam.make_grad_sync) -- the backward plans already produce every gradient in
        its slot there, so a bucket is a contiguous slice of that buffer and nothing is copied on either side of
        the collective.  Without them the buckets own their storage and gradients are copied in (stand-alone use)."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.enabled = True
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets: List[dict] = []
        self.where: Dict[torch.nn.Parameter, tuple] = {}
        if flat is not None:
            self.params = [p for p, _, _ in spans]
            # contiguous ranges of the flat buffer, cut back to front: backward finishes the LAST parameters first,
            # so the highest range completes (and is all-reduced) first
            ranges = self.cut_ranges([sp[2] for sp in spans], cap, int(big_mb * (1 << 20) / 4), int(tail_mb * (1 << 20) / 4))
            for lo, hi in ranges:
                a = spans[lo][1]
                bnd = spans[hi - 1][1] + spans[hi - 1][2]
                b = dict(params=[sp[0] for sp in spans[lo:hi]], n=bnd - a, flat=flat[a:bnd])
                for p, off, k in spans[lo:hi]:
                    self.where[p] = (len(self.buckets), off - a, k)
                self.buckets.append(b)
        else:
            self.params = [p for p in params if p.requires_grad]
            # backward produces gradients roughly in reverse registration order
            order = list(reversed(self.params))
            cur, cur_n = [], 0
            for p in order:
                if cur and cur_n + p.numel() > cap:
                    self.buckets.append(dict(params=cur, n=cur_n))
                    cur, cur_n = [], 0
                cur.append(p)
                cur_n += p.numel()
            if cur:
                self.buckets.append(dict(params=cur, n=cur_n))
            for bi, b in enumerate(self.buckets):
                dev = b["params"][0].device
                b["flat"] = torch.zeros(b["n"], dtype=torch.float32, device=dev)
                off = 0
                for p in b["params"]:
                    self.where[p] = (bi, off, p.numel())
                    off += p.numel()
        for b in self.buckets:
            b["pending"] = set(b["params"])
            b["handle"] = None
        self.begin()

    @staticmethod
    def cut_ranges(sizes: List[int], cap: int, big: int, tail: int) -> List[tuple]:
        """[lo, hi) index ranges over the parameters in flat-buffer order, listed back to front (= firing order): <= ``cap``
        elements each; a parameter of >= ``big`` elements alone; the front of the buffer (gradients produced last) closed off
        as a bucket of <= ``tail`` elements."""
        ranges, hi, n = [], len(sizes), 0
        for i in range(len(sizes) - 1, -1, -1):
            k = sizes[i]
            if k >= big:                          # its own bucket, fired as soon as its gradient exists
                if hi > i + 1:
                    ranges.append((i + 1, hi))
                ranges.append((i, i + 1))
                hi, n = i, 0
                continue
            n += k
            if n > cap and hi - i > 1:
                ranges.append((i + 1, hi))
                hi, n = i + 1, k
        if hi > 0:
            # the last range [0, hi): split off the parameters produced last into a small tail bucket
            acc, cut = 0, 0
            for i in range(hi):
                if acc + sizes[i] > tail:
                    break
                acc += sizes[i]
                cut = i + 1
            if 0 < cut < hi:
                ranges.append((cut, hi))
                ranges.append((0, cut))
            else:
                ranges.append((0, hi))
        return ranges

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
        slot = b["flat"][off:off + k]
        if g.data_ptr() != slot.data_ptr():     # in-place mode: the plan already wrote the gradient here
            slot.copy_(g.reshape(-1))
        b["seen"].add(p)
        b["pending"].discard(p)
        if not b["pending"] and b["handle"] is None:
            self._fire(b)

    def _fire(self, b):
        if b["flat"].is_cuda:
            from . import train_ops as _T
            _T.wgrad_sync_for_collective()      # slots of this bucket may have been written on the weight-gradient side stream
        if self.world > 1:
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        elif int(os.environ.get("HRV_FAKE_ALLREDUCE", "0") or 0) > 0 and b["flat"].is_cuda:
            b["handle"] = _FakeCollective(b["flat"], int(os.environ["HRV_FAKE_ALLREDUCE"]))
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


class GraphGradSync(GradSync):
    """The gradient synchronisation of a hipGraph-captured data-parallel iteration (graph.GraphedIteration): a collective is a
    HOST call (RCCL's own stream and launch protocol; gloo stages through the host), so it cannot live inside the capture.
    The iteration is captured as SEGMENTS instead -- the backward plans hand their gradients over exactly as they do to
    GradSync (in place in the optimizer's flat buffer), nothing is reduced bucket by bucket, and ``wait()`` -- the first thing the
    fused optimizer step does -- calls ``cut(self)``: during the capture that ends the running segment and opens the next one;
    on every replay the wrapper all-reduces the optimizer's ENTIRE flat gradient buffer (back-to-back collectives over the bucket
    ranges) between the two segments.  1 / world rides in the fused Adam launch as with GradSync."""

    def __init__(self, flat: torch.Tensor, spans, process_group=None, bucket_mb: float = 64.0):
        super().__init__(None, bucket_mb, process_group, flat=flat, spans=spans)
        self.whole = flat
        self.cut = None               # set by graph.GraphedIteration for the capture

    def _fire(self, b):
        b["handle"] = True

    def reduce(self):
        """what a replay does where the capture was cut: the whole flat gradient buffer, as back-to-back collectives over GradSync's
        contiguous ranges (all in flight together; one 400 MB tensor takes a slow path in gloo)"""
        if self.world > 1:
            hs = [dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.pg, async_op=True) for b in self.buckets]
            for h in hs:
                h.wait()

    def wait(self):
        if not self.enabled:
            return
        super().wait()                # zero-fills parameters without a gradient; no collective was started
        if self.cut is not None and self.whole.is_cuda and torch.cuda.is_current_stream_capturing():
            self.cut(self)
        else:
            self.reduce()             # eager (the warm-up iterations): same result as the replayed segments


def broadcast_module(module: torch.nn.Module, src: int = 0, process_group=None):
    """Initial replica synchronisation (parameters and buffers, incl. the spectral-norm u, v)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
    from . import ops
    ops.LOAD_EPOCH[0] += 1          # (the broadcast writes through .data: the parameters' _version does not move)
    ops.WEIGHTS_EPOCH[0] += 1
