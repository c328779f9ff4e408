"""hipGraph capture of an inference step (config #5 of SURVEY 8d: "end-to-end inference, hipGraph").

The try-on step at serving sizes is a few hundred launches; at the tocg resolution (256x192) most of them
run for a few microseconds, so the step is bounded by the host's launch rate, not by the GPU.  The whole step
-- every launch of the C-ABI library goes to torch's CURRENT stream, and the library itself never allocates or
synchronises -- is recorded once into a hipGraph (``torch.cuda.CUDAGraph`` IS hipGraph on ROCm) and replayed
with one host call per step.  Inputs live in static device buffers that ``__call__`` refreshes; outputs are the
static tensors the captured step produced (valid until the next replay).

Only inference steps are captured: the training steps are GPU-bound (kernel time == wall time) and
their optimizer/grad-sync control flow is host-driven."""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import _lib
from .ops import HrvError


class GraphedStep:
    def __init__(self, fn: Callable[[Dict[str, torch.Tensor]], object], example_inputs: Dict[str, torch.Tensor],
                 warmup: int = 2):
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
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = fn(self.static_in)
        self.replays = 0

    def __call__(self, inputs: Dict[str, torch.Tensor]):
        for k, dst in self.static_in.items():
            src = inputs[k]
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise HrvError(f"GraphedStep: input '{k}' is {tuple(src.shape)}/{src.dtype}, captured with "
                               f"{tuple(dst.shape)}/{dst.dtype} (re-capture for a new shape)")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.static_out


def graphed_tryon(opt, tocg, generator, example_inputs: Dict[str, torch.Tensor], warmup: int = 2) -> GraphedStep:
    """The body of test_generator.py:118-219 (pipeline.tryon_step) as one hipGraph."""
    from .pipeline import tryon_step
    keys = ("cloth", "cloth_mask", "parse_agnostic", "densepose", "agnostic")
    return GraphedStep(lambda b: tryon_step(opt, tocg, generator, b), {k: example_inputs[k] for k in keys}, warmup)


def graphed_condition(opt, tocg, input1: torch.Tensor, input2: torch.Tensor, warmup: int = 2) -> GraphedStep:
    """ConditionGenerator.forward (networks.py:98-159) as one hipGraph; call with {'input1':…, 'input2':…}."""
    return GraphedStep(lambda b: tocg(opt, b["input1"], b["input2"]), {"input1": input1, "input2": input2}, warmup)
