"""hipGraph capture of an inference step (config #5 of SURVEY 8d: "end-to-end inference, hipGraph").

The try-on step at serving sizes is a few hundred launches; at the tocg resolution (256x192) most of them
run for a few microseconds, so the step is bounded by the host's launch rate, not by the GPU.  The whole step
-- every launch of the C-ABI library goes to torch's CURRENT stream, and the library itself never allocates or
synchronises -- is recorded once into a hipGraph (``torch.cuda.CUDAGraph`` IS hipGraph on ROCm) and replayed
with one host call per step.  Inputs live in static device buffers that ``__call__`` refreshes; outputs are the
static tensors the captured step produced (valid until the next replay).

``GraphedTrainStep`` captures a whole train_generator.py iteration the same way (data parallel: a chain of three graphs with
the two gradient all-reduces as host calls between them -- GraphedIteration).  The iteration is GPU-bound (kernel time == wall time), so the replay is
no faster than eager -- it exists so that the launch path is not on the critical path when a future kernel set is faster."""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import _lib
from .ops import HrvError


class CaptureGuard:
    """What a captured region baked into its hipGraph besides the tensors torch's graph pool owns: raw addresses of
    Python-owned buffers -- the per-device split-K / weight-gradient workspace (ops._WS), the pack buffers and record tables
    of the plans' PackBatches, the cached packed streams of frozen weights (train_ops._FROZEN_PACKS), persistent per-plan buffers (S2DConv._w2, the spectral-norm sigma buffer; those are never
    re-allocated while their module lives).  ``before()`` is taken ahead of the capture, ``after()`` behind it:

    * every such tensor alive at the end of the capture is KEPT alive by the guard, so a later eager call that grows the
      workspace (ops._workspace replaces the tensor) or a PackBatch.reset() cannot hand the graph's memory to someone else --
      replays keep reading and writing buffers only they reference;
    * a PackBatch the region used that was reset() afterwards (a weight moved: optimizer re-creation, .to(), load into a new
      module) makes the graph STALE -- its pack launches would read the weights' old addresses: ``check()`` raises HrvError
      instead of replaying."""

    def __init__(self, modules=()):
        from . import ops
        self._runs = {id(o): o.runs for o in list(ops.GRAPH_WATCH)}
        self.keep, self.watch = [], []
        # inference modules whose cached plan (packed weight streams, per-layer constants: ``module._plan``) the captured region
        # reads by address: the plan object alive at the end of the capture is kept (its buffers stay the graph's), and a module
        # that has REBUILT its plan since (a parameter changed: eval between training steps, load_state_dict) makes the graph stale
        self._modules = list(modules)
        self.plans = []

    def after(self):
        import weakref
        from . import ops
        self.keep = list(ops._WS.values())
        # packed streams of frozen weights (train_ops' cache: VGG19, the serving plan's conv_p2 layers) that existed before the capture
        # were baked in by address too; the cache drops everything when it outgrows 256 entries
        from . import train_ops as T
        self.keep.extend(v[0] for v in T._FROZEN_PACKS.values())
        for o in list(ops.GRAPH_WATCH):
            if o.runs != self._runs.get(id(o), 0):           # launched inside the captured region
                self.keep.extend(o.graph_keep())
                self.watch.append((weakref.ref(o), o.generation))
        self.plans = [(weakref.ref(m), getattr(m, "_plan", None)) for m in self._modules]
        return self

    def check(self, who: str):
        for ref, gen in self.watch:
            o = ref()
            if o is not None and o.generation != gen:
                raise HrvError(f"{who}: a weight-pack batch of the captured plan was reset after the capture (a weight moved to a new "
                               "address); the graph holds the old addresses -- capture again")
        for ref, plan in self.plans:
            m = ref()
            key = getattr(m, "_plan_key", None) if m is not None else None
            moved = (key is not None and len(key) > 1 and
                     key[1] != tuple(t._version for t in list(m.parameters()) + list(m.buffers())))      # changed, not yet re-planned
            if m is not None and plan is not None and (getattr(m, "_plan", None) is not plan or moved):
                raise HrvError(f"{who}: {type(m).__name__} rebuilt its inference plan (or changed its weights) after the capture; the graph "
                               "still reads the packed weights of the old plan -- capture again")


class GraphedStep:
    def __init__(self, fn: Callable[[Dict[str, torch.Tensor]], object], example_inputs: Dict[str, torch.Tensor],
                 warmup: int = 2, modules=()):
        """``fn(inputs) -> tensors`` (any nesting of dict / list / tuple).  ``example_inputs`` fix shapes and dtypes.
        ``warmup`` eager calls run first on a side stream so that every lazily built plan, packed weight and
        cached table exists before the capture (nothing may be created on the host during it)."""
        _lib.load()                                  # fail loudly without the HIP extension
        for k, v in example_inputs.items():
            if not v.is_cuda:
                raise HrvError(f"GraphedStep: input '{k}' is on {v.device}; hipGraph capture needs device-resident inputs")
        self.static_in = {k: v.clone() for k, v in example_inputs.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                fn(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        guard = CaptureGuard(modules)       # (``modules``: the networks whose cached inference plans ``fn`` runs)
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = fn(self.static_in)
        self.guard = guard.after()
        self.replays = 0

    def __call__(self, inputs: Dict[str, torch.Tensor]):
        for k, dst in self.static_in.items():
            src = inputs[k]
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise HrvError(f"GraphedStep: input '{k}' is {tuple(src.shape)}/{src.dtype}, captured with "
                               f"{tuple(dst.shape)}/{dst.dtype} (re-capture for a new shape)")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.guard.check("GraphedStep")
        self.graph.replay()
        self.replays += 1
        return self.static_out


def graphed_tryon(opt, tocg, generator, example_inputs: Dict[str, torch.Tensor], warmup: int = 2) -> GraphedStep:
    """The body of test_generator.py:118-219 (pipeline.tryon_step) as one hipGraph."""
    from .pipeline import tryon_step
    keys = ("cloth", "cloth_mask", "parse_agnostic", "densepose", "agnostic")
    return GraphedStep(lambda b: tryon_step(opt, tocg, generator, b), {k: example_inputs[k] for k in keys}, warmup,
                       modules=(tocg, generator))


def graphed_condition(opt, tocg, input1: torch.Tensor, input2: torch.Tensor, warmup: int = 2) -> GraphedStep:
    """ConditionGenerator.forward (networks.py:98-159) as one hipGraph; call with {'input1':…, 'input2':…}."""
    return GraphedStep(lambda b: tocg(opt, b["input1"], b["input2"]), {"input1": input1, "input2": input2}, warmup, modules=(tocg,))


class GraphedIteration:
    """``fn()`` -- a whole training iteration over STATIC device buffers, optimizer steps included -- as one hipGraph, or, under
    data-parallel training, as a CHAIN of hipGraphs with the gradient collectives between them.
    ``optimizers``: hr_viton_amd.optim.Adam instances built with ``device_step=True`` (their learning rates are pushed to
    the device before every replay).  Data parallel: give every optimizer ``make_grad_sync(graph=True)``
    (parallel.GraphGradSync) -- the capture is cut where an optimizer waits for its gradients (segment | all-reduce of that
    optimizer's whole flat gradient buffer | segment ...); all segments share one graph memory pool and replay in capture order.
    The bucketed, backward-overlapped GradSync is host-driven and cannot be captured: refused."""

    def __init__(self, fn: Callable[[], object], optimizers, warmup: int = 3):
        import gc
        from .parallel import GraphGradSync
        _lib.load()
        syncs = []
        for o in optimizers:
            if not getattr(o, "device_step", False):
                raise HrvError("GraphedIteration: build the optimizers with hr_viton_amd.optim.Adam(..., device_step=True)")
            gs = getattr(o, "grad_sync", None)
            if gs is not None and not isinstance(gs, GraphGradSync):
                raise HrvError("GraphedIteration: the bucketed GradSync is host-driven; under data-parallel training build the "
                               "synchronisation with optimizer.make_grad_sync(graph=True)")
            if gs is not None:
                syncs.append(gs)
        self.optimizers = list(optimizers)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):          # eager iterations: plans, pack records, flat buffers, device scalars
                fn()                                 # (a GraphGradSync reduces eagerly here: the replicas stay in lock-step)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.warmup_iterations = max(1, warmup)
        self.graphs = [torch.cuda.CUDAGraph()]
        self.cuts = []                               # cuts[i]: the GraphGradSync reduced between graphs[i] and graphs[i + 1]
        guard = CaptureGuard()
        if not syncs:
            with torch.cuda.graph(self.graphs[0]):
                self.static_out = fn()
        else:
            pool = torch.cuda.graph_pool_handle()

            def cut(sync):
                self.graphs[-1].capture_end()
                self.cuts.append(sync)
                self.graphs.append(torch.cuda.CUDAGraph())
                self.graphs[-1].capture_begin(pool=pool)
            for gs in syncs:
                gs.cut = cut
            gc.collect()
            torch.cuda.empty_cache()
            cap = torch.cuda.Stream()
            cap.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(cap):
                    self.graphs[0].capture_begin(pool=pool)
                    try:
                        self.static_out = fn()
                    except BaseException:
                        # fn() raised inside a capture: close the open segment so the stream does not stay in capture mode (every later
                        # launch on it would fail with "operation not permitted when stream is capturing"), then let the error out
                        try:
                            self.graphs[-1].capture_end()
                        except Exception:
                            pass
                        raise
                    self.graphs[-1].capture_end()
            finally:
                for gs in syncs:
                    gs.cut = None
            torch.cuda.current_stream().wait_stream(cap)
            torch.cuda.synchronize()
        self.graph = self.graphs[0]
        self.guard = guard.after()
        self.replays = 0

    def __call__(self):
        self.guard.check("GraphedIteration")
        for o in self.optimizers:
            o.push_lr()
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.cuts):
                self.cuts[i].reduce()
        self.replays += 1
        return self.static_out


class GraphedTrainStep:
    """One train_generator.py iteration (pipeline.generator_train_step: G step + D step, both Adam updates) as one hipGraph.

    Requirements, checked here: both optimizers were built with ``device_step=True`` (step count and learning rate on the
    device -- a replay must not freeze them); data parallel: ``make_grad_sync(graph=True)`` on both (three graph segments with
    the two gradient all-reduces between them, GraphedIteration).  ``inputs``: {'x': [N,9,H,W], 'parse7': the
    label-map activation tensor [N,H,W,8], 'im': [N,3,H,W]} -- static buffers refreshed by ``__call__``.  SPADE noise is
    drawn inside the captured region (torch's graph-safe Philox) unless ``noise`` / ``noise_d`` plane dicts are given
    (then they are static inputs too).  Losses come back as the static tensors of the capture."""

    def __init__(self, opt, generator, discriminator, crit_gan, crit_feat, crit_vgg, opt_g, opt_d, inputs: Dict[str, torch.Tensor],
                 noise=None, noise_d=None, warmup: int = 3):
        from .ops import Act
        from .pipeline import generator_train_step
        _lib.load()
        for o in (opt_g, opt_d):
            if not getattr(o, "device_step", False):
                raise HrvError("GraphedTrainStep: build the optimizers with hr_viton_amd.optim.Adam(..., device_step=True)")
        self.opt_g, self.opt_d = opt_g, opt_d
        self.static_in = {k: v.clone() for k, v in inputs.items()}
        self.noise = None if noise is None else {k: [z.clone() for z in v] for k, v in noise.items()}
        self.noise_d = None if noise_d is None else {k: [z.clone() for z in v] for k, v in noise_d.items()}

        def body():
            b = self.static_in
            return generator_train_step(opt, generator, discriminator, crit_gan, crit_feat, crit_vgg, opt_g, opt_d, b["x"],
                                        Act(b["parse7"], 7), b["im"], getattr(opt_g, "grad_sync", None),
                                        getattr(opt_d, "grad_sync", None), noise=self.noise, noise_d=self.noise_d)
        self._it = GraphedIteration(body, (opt_g, opt_d), warmup)
        self.graph, self.static_out = self._it.graph, self._it.static_out
        self.replays = 0

    def __call__(self, inputs: Dict[str, torch.Tensor] = None):
        if inputs is not None:
            for k, dst in self.static_in.items():
                src = inputs[k]
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise HrvError(f"GraphedTrainStep: input '{k}' is {tuple(src.shape)}/{src.dtype}, captured with "
                                   f"{tuple(dst.shape)}/{dst.dtype}")
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src, non_blocking=True)
        out = self._it()
        self.replays += 1
        return out
