"""Tensor-level host wrappers over the C ABI (include/hrviton_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every byte of
compute goes through libhrviton_hip.so.  Activations are NHWC fp32 ``Act``
views (tensor [N,H,W,Cs], a channel offset and a logical channel count) so
that channel concatenation / slicing never copies.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple, Union

import weakref

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, HrvError  # noqa: F401


# bumped by hr_viton_amd.optim.Adam.step(): the fused Adam kernel writes parameters through raw
# pointers (no torch version-counter bump), so cached inference plans key on this as well
WEIGHTS_EPOCH = [0]     # global step counter of the fused optimizers (per-iteration caches key on it)
LOAD_EPOCH = [0]        # bumped by checkpoint loads and replica broadcasts (writes torch's _version may not show)


def weights_epoch(tensors) -> int:
    """Sum of the per-parameter update counters the fused Adam maintains (``_hrv_epoch``): the fused step writes
    parameters through raw pointers, which torch's ``_version`` does not see.  A cached plan of a module that no
    optimizer touches (the frozen tocg inside train_generator.py) therefore stays valid across steps."""
    return sum(getattr(t, "_hrv_epoch", 0) for t in tensors)


def _ceil4(c: int) -> int:
    return (c + 3) // 4 * 4


def _cpad(c: int, bf16: bool) -> int:
    """Channel padding of an NHWC tensor: one 16-byte gather group (4 fp32 / 8 bf16 channels)."""
    return (c + 7) // 8 * 8 if bf16 else (c + 3) // 4 * 4


def patch_tile(bf16_sources: bool, KH: int, KW: int, stride: int, pad: int, nsrc: int, up: int, C: int, cols: int,
               N: int, H: int, W: int, wide: bool = False) -> int:
    """Patch-mode tile of the conv engine (conv_f32.hip, VAR bit 6) for this layer, or 0.  Patch mode: 3x3 stride-1
    'same' convolution over ONE bf16-stored source with C % 128 == 0; the 8x16-pixel tile keeps its 10x18 halo
    patch resident in LDS and only the weight tiles stream (the implicit-GEMM gather re-reads every activation
    pixel from L2 once per tap).  tile_cfg 17: 128 columns, 18: 64 columns (column counts that are odd multiples
    of 64 -- the SPADE gamma|beta convs of the 80/144/272-channel blocks).  Needs enough tiles to fill the chip
    (else the gather tiles with split-K win).  HRV_CONV_PATCH=0 disables it, =16 selects the 16x16-pixel tile."""
    env = os.environ.get("HRV_CONV_PATCH", "1")
    if not (bf16_sources and KH == 3 and KW == 3 and stride == 1 and pad == 1 and nsrc == 1 and up == 0 and
            C % 128 == 0 and env != "0"):
        return 0
    # tile_cfg 19 (conv_patchw.hip): 16x16-pixel tiles x up to 192 columns per block, one block per CU, weights
    # streamed once per 256 pixels through 3 LDS stages.  Opt-in (HRV_CONV_PATCHW=1): measured in the training step
    # it is 5-10 % SLOWER than the 8x16 tiles below (up_4 gamma|beta 2.42 vs 2.25 ms) -- with one block per CU nothing
    # overlaps the SPADE epilogue's 246 KB of loads/stores per tile, which the two resident blocks of cfg 17/18 hide.
    # (``wide``: the caller's epilogue is one conv_patchw.hip implements -- the SPADE modulate sites)
    if wide and os.environ.get("HRV_CONV_PATCHW", "0") != "0" and N * ((H + 15) // 16) * ((W + 15) // 16) >= 256:
        return 19
    c64 = (cols + 63) // 64
    if c64 * 64 - cols > 32:
        return 0
    wide = c64 % 2 == 0
    if N * ((H + 7) // 8) * ((W + 15) // 16) * (c64 // 2 if wide else c64) < 512:
        return 0
    if env == "16" and wide:
        return 16
    return 17 if wide else 18


def patch_tile_ok(*a) -> bool:
    return patch_tile(*a) != 0


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def act_bytes(a, C: Optional[int] = None) -> float:
    """Algorithmic HBM bytes of one pass over the logical channels of an NHWC view (padding not counted)."""
    if a is None:
        return 0.0
    return float(a.N * a.H * a.W * (a.C if C is None else C) * (2 if a.bf16 else 4))


def require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise HrvError(f"{what}: tensor is on {t.device}; the MI355X path has no CPU fallback "
                       "(move inputs to cuda, reference semantics of opt.cuda=True)")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise HrvError(f"{what}: expected float32 (or bfloat16 inside the bf16 path), got {t.dtype}")


@dataclass
class Act:
    """NHWC activation view: channels [coff, coff+C) of ``t`` ([N,H,W,Cs], contiguous)."""
    t: torch.Tensor
    C: int
    coff: int = 0

    @property
    def N(self): return self.t.shape[0]
    @property
    def H(self): return self.t.shape[1]
    @property
    def W(self): return self.t.shape[2]
    @property
    def cstride(self): return self.t.shape[3]
    @property
    def bf16(self):
        return self.t.dtype == torch.bfloat16

    @property
    def Cp(self):
        """Channels the conv engine consumes (padded to one 16-byte group; pad channels are zero)."""
        return _cpad(self.C, self.bf16)

    def slice(self, c0: int, c: int) -> "Act":
        return Act(self.t, c, self.coff + c0)


class ActUp:
    """x = cat(nearest_up2(lo), hi) along channels, NEVER MATERIALISED (SPADEGenerator.up + the concatenation of the resized input,
    network_generator.py:203,226-242): ``lo`` is the previous block's output [N, H/2, W/2, Clo] (fp32), ``hi`` the stem's 16
    channels [N, H, W, Chi] (fp32).  Its readers -- the statistics pass, the fused SPADE forward, the normalisation backward
    (csrc: x_up_channels) -- address channel c < Clo at (y >> 1, x >> 1) of ``lo``: 4 x fewer HBM bytes for those channels
    than the 4-fold fp32 copy, which is never written."""

    def __init__(self, lo: Act, hi: Act):
        assert not lo.bf16 and not hi.bf16 and hi.H == 2 * lo.H and hi.W == 2 * lo.W and lo.N == hi.N and lo.C % 16 == 0
        self.lo, self.hi = lo, hi
        self.t = hi.t                      # (device / shape holder)

    N = property(lambda self: self.hi.N)
    H = property(lambda self: self.hi.H)
    W = property(lambda self: self.hi.W)
    C = property(lambda self: self.lo.C + self.hi.C)
    Cp = property(lambda self: self.lo.C + self.hi.C)
    cstride = property(lambda self: self.lo.C + self.hi.C)
    coff = 0
    bf16 = False


def alloc(N: int, H: int, W: int, C: int, device, bf16: bool = False) -> Act:
    cs = _cpad(C, bf16)
    dt = torch.bfloat16 if bf16 else torch.float32
    if cs == C:
        t = torch.empty((N, H, W, cs), dtype=dt, device=device)
    else:  # pad channels must read as zero
        t = torch.zeros((N, H, W, cs), dtype=dt, device=device)
    return Act(t, C, 0)


def alloc_written(N: int, H: int, W: int, C: int, device, bf16: bool = False) -> Act:
    """``alloc`` without the zero fill, for a producer that is handed the PADDED channel count and writes every one of them (the pad
    channels then hold what the kernel computes from its source's zero pads: zeros) -- the resizes, the 3x3 average pool."""
    return Act(torch.empty((N, H, W, _cpad(C, bf16)), dtype=torch.bfloat16 if bf16 else torch.float32, device=device), C, 0)


def to_nhwc(x: torch.Tensor, out: Optional[Act] = None, bf16: bool = False, zero_tail: int = 0) -> Act:
    """fp32 NCHW (reference layout) -> NHWC Act (fp32, or bf16 for the bf16 engine).  ``zero_tail`` (with ``out``): that many channels
    behind the converted ones are written as zeros too -- the pad channels of a tensor the caller owns (no separate fill)."""
    require_cuda(x, "to_nhwc")
    x = x.contiguous()
    N, Cc, H, W = x.shape
    # (a caller's ``out`` may be one slice of a concatenation buffer: nothing outside it is touched unless the caller says so)
    assert out is not None or zero_tail == 0
    if out is None:
        # (the converter writes the pad channels of a tensor allocated HERE as zeros itself: no fill of the whole tensor)
        out = Act(torch.empty((N, H, W, _cpad(Cc, bf16)), dtype=torch.bfloat16 if bf16 else torch.float32, device=x.device), Cc, 0)
        zero_tail = out.cstride - Cc
    lib = _lib.load()
    with _Timed("layout", "nchw_to_nhwc", 0.0, 4.0 * x.numel() + act_bytes(out, Cc)):
        if out.bf16:
            _lib.check(lib.hrv_nchw_f32_to_nhwc_bf16(x.data_ptr(), N, Cc, H, W, out.t.data_ptr(), out.cstride, out.coff,
                                                     zero_tail, _stream()), "hrv_nchw_f32_to_nhwc_bf16")
        else:
            _lib.check(lib.hrv_nchw_to_nhwc_f32(x.data_ptr(), N, Cc, H, W, out.t.data_ptr(), out.cstride, out.coff,
                                                zero_tail, _stream()), "hrv_nchw_to_nhwc_f32")
    return out


def to_nchw(a: Act, c0: int = 0, c: Optional[int] = None) -> torch.Tensor:
    """NHWC Act (channel range) -> contiguous fp32 NCHW tensor."""
    c = a.C - c0 if c is None else c
    out = torch.empty((a.N, c, a.H, a.W), dtype=torch.float32, device=a.t.device)
    lib = _lib.load()
    with _Timed("layout", "nhwc_to_nchw", 0.0, 4.0 * out.numel() + act_bytes(a, c)):
        if a.bf16:
            _lib.check(lib.hrv_nhwc_bf16_to_nchw_f32(a.t.data_ptr(), a.cstride, a.coff + c0, a.N, c, a.H, a.W,
                                                     out.data_ptr(), _stream()), "hrv_nhwc_bf16_to_nchw_f32")
        else:
            _lib.check(lib.hrv_nhwc_to_nchw_f32(a.t.data_ptr(), a.cstride, a.coff + c0, a.N, c, a.H, a.W, out.data_ptr(),
                                                _stream()), "hrv_nhwc_to_nchw_f32")
    return out


# --------------------------------------------------------------------------
# profiling hook (bench.py): per-launch HIP events on the launch stream
# --------------------------------------------------------------------------
class _Prof:
    enabled = False
    records: list = []


def profile_begin():
    _Prof.enabled = True
    _Prof.records = []


def profile_end(kernels: bool = False):
    """Returns [(kind, name, flops, bytes, ms)] for every launch since profile_begin(); ``kernels``: 6-tuples with the
    name of the device kernel family that served the launch (what a rocprofv3 kernel trace groups by) at the end."""
    _Prof.enabled = False
    torch.cuda.synchronize()
    out = [((k, n, fl, by, s.elapsed_time(e), kn) if kernels else (k, n, fl, by, s.elapsed_time(e)))
           for (k, n, fl, by, kn, s, e) in _Prof.records]
    _Prof.records = []
    return out


class _Timed:
    def __init__(self, kind, name, flops, nbytes, kernel=None):
        # ``kernel``: the hrv:: kernel (family) behind the C entry point; the launch kind stands in where one kind = one kernel
        self.args = (kind, name, flops, nbytes, kernel or kind)

    def __enter__(self):
        if _Prof.enabled:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if _Prof.enabled:
            self.e.record()
            _Prof.records.append((*self.args, self.s, self.e))
        return False


_WS = {}
# objects that own device buffers whose RAW ADDRESSES end up inside captured hipGraphs (train_ops.PackBatch: pack buffers and
# record tables).  Each has ``generation`` (bumped when it drops its buffers), ``runs`` (bumped when it launches) and
# ``graph_keep()`` (the tensors a graph that used it must keep alive) -- see graph.CaptureGuard.
GRAPH_WATCH = weakref.WeakSet()


WS_PER_STREAM = [False]      # set by train_ops.wgrad_side while its body runs on the side stream


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Per-device split-K scratch (stream-ordered reuse: every conv launch on the stream finishes
    reading it before the next one writes).  A request beyond the current size REPLACES the tensor; a captured hipGraph
    that recorded the old address keeps the old tensor alive itself (graph.CaptureGuard), so its replays stay on memory
    nobody else owns."""
    # One scratch per device -- except inside a ``train_ops.wgrad_side`` body (opt-in, HRV_WGRAD_SIDE=1), where the weight gradients
    # run on a second stream NEXT to the data-gradient chain and get that stream's own scratch.  (Keyed by every current stream, the
    # warm-up and capture streams of graph.GraphedStep each grew a full-size scratch that nothing released -- ADVICE r5.)
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if WS_PER_STREAM[0] else 0)
    t = _WS.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
        _WS[key] = t
    return t


SrcSpec = Union[Act, Tuple[Act, int, int]]  # Act or (Act, up_shift, pre_act)


class ConvLayer:
    """One convolution of the path with its fused epilogue (hrv_conv2d_nhwc_f32).

    ``weight``: torch OIHW fp32 (any device); ``src_real``: how the Cin axis splits
    over the (channel-concatenated) sources.  Packed weights are built lazily per
    tile configuration with the HOST packer hrv_conv2d_pack_weight_f32.
    """

    def __init__(self, weight: torch.Tensor, src_real: Sequence[int], device, scale: Optional[torch.Tensor] = None,
                 shift: Optional[torch.Tensor] = None, stride: int = 1, pad: int = 1, act: int = ACT_NONE,
                 slope: float = 0.2, name: str = "conv", bf16: bool = False, out_f32: bool = False,
                 mma_bf16: bool = False):
        # bf16: sources + weights are bf16 (fp32 accumulate); out_f32: in bf16 mode the output is written
        # as fp32 (tensors that feed an InstanceNorm stay fp32)
        # mma_bf16: every tensor stays fp32, only the matrix-core operands are rounded to bf16 while staged
        # (hrv_conv2d_t.mixed_flags bit 3) -- the inference counterpart of train_ops.MMA_BF16
        assert not (bf16 and mma_bf16)
        self.bf16 = bf16
        self.mixed = mma_bf16
        self.out_f32 = out_f32 or not bf16
        w = weight.detach().to("cpu", torch.float32).contiguous()
        self.Cout, cin, self.KH, self.KW = w.shape
        assert sum(src_real) == cin, (name, src_real, cin)
        assert 1 <= len(src_real) <= _lib.HRV_MAX_SRC
        self.w_cpu = w
        self.src_real = list(src_real)
        self.src_pad = [_cpad(c, bf16) for c in src_real]
        self.device = device
        self.stride, self.pad, self.act, self.slope, self.name = stride, pad, act, slope, name
        self.scale = None if scale is None else scale.detach().to(device, torch.float32).contiguous()
        self.shift = None if shift is None else shift.detach().to(device, torch.float32).contiguous()
        self._packed = {}
        self._w_dev = None

    def _get_packed(self, cfg: int) -> torch.Tensor:
        if cfg not in self._packed:
            lib = _lib.load()
            n = len(self.src_pad)
            srcC = (C.c_int32 * n)(*self.src_pad)
            srcR = (C.c_int32 * n)(*self.src_real)
            if self.bf16 or self.mixed:
                elems = lib.hrv_conv2d_packed_elems_bf16(self.Cout, self.KH, self.KW, n, srcC, cfg)
                buf = torch.empty(max(elems, 1), dtype=torch.int16)
                _lib.check(lib.hrv_conv2d_pack_weight_bf16(self.w_cpu.data_ptr(), self.Cout, self.KH, self.KW, n, srcC,
                                                           srcR, cfg, buf.data_ptr()), "hrv_conv2d_pack_weight_bf16")
            else:
                elems = lib.hrv_conv2d_packed_elems(self.Cout, self.KH, self.KW, n, srcC, cfg)
                buf = torch.empty(max(elems, 1), dtype=torch.float32)
                _lib.check(lib.hrv_conv2d_pack_weight_f32(self.w_cpu.data_ptr(), self.Cout, self.KH, self.KW, n, srcC,
                                                          srcR, cfg, buf.data_ptr()), "hrv_conv2d_pack_weight_f32")
            if elems <= 0:
                raise HrvError(f"{self.name}: hrv_conv2d_packed_elems failed")
            self._packed[cfg] = buf.to(self.device)
        return self._packed[cfg]

    def out_hw(self, H: int, W: int) -> Tuple[int, int]:
        return ((H + 2 * self.pad - self.KH) // self.stride + 1, (W + 2 * self.pad - self.KW) // self.stride + 1)

    def flops(self, N, Ho, Wo) -> float:
        return 2.0 * N * Ho * Wo * getattr(self, "flops_cout", self.Cout) * sum(self.src_real) * self.KH * self.KW

    def __call__(self, srcs: Sequence[SrcSpec], out: Optional[Act] = None, residual: Optional[Act] = None,
                 H: Optional[int] = None, W: Optional[int] = None, out_up: int = 0, spade=None,
                 out_channels: Optional[int] = None, cfg: Optional[int] = None) -> Act:
        """``out_up=1``: ``out`` is (2Ho x 2Wo), results replicated 2x2 (fused nearest upsample).
        ``spade``: hrv_spade_epi_t for the fused gamma||beta + modulate epilogue (then
        ``out_channels`` = C of the normalised tensor)."""
        lib = _lib.load()
        specs = [(s, 0, ACT_NONE) if isinstance(s, Act) else s for s in srcs]
        assert len(specs) == len(self.src_real), (self.name, len(specs), self.src_real)
        a0, up0, _ = specs[0]
        N = a0.N
        if H is None:
            H, W = (a0.H << up0, a0.W << up0) if up0 >= 0 else (a0.H >> -up0, a0.W >> -up0)
        Ho, Wo = self.out_hw(H, W)
        oc = self.Cout if out_channels is None else out_channels
        if out is None:
            out = alloc(N, Ho << out_up, Wo << out_up, oc, a0.t.device, self.bf16 and not self.out_f32)
        assert all(a.bf16 == self.bf16 for a, _, _ in specs), (self.name, "conv sources must match the engine dtype")
        assert self.bf16 or not out.bf16, (self.name, "fp32 engine writes fp32")
        d = _lib.hrv_conv2d_t()
        mixed = (0 if out.bf16 else 1) | (0 if (residual is None or residual.bf16) else 2)
        d.N, d.H, d.W, d.Ho, d.Wo = N, H, W, Ho, Wo
        d.KH, d.KW, d.stride, d.pad = self.KH, self.KW, self.stride, self.pad
        d.nsrc = len(specs)
        for i, (a, up, pre) in enumerate(specs):
            assert a.C == self.src_real[i], (self.name, i, a.C, self.src_real[i])
            assert ((a.H << up, a.W << up) if up >= 0 else (a.H >> -up, a.W >> -up)) == (H, W) and a.N == N, \
                (self.name, i, a.t.shape, up, H, W)
            s = d.src[i]
            s.ptr, s.C, s.cstride, s.coff = a.t.data_ptr(), self.src_pad[i], a.cstride, a.coff
            s.up_shift, s.pre_act, s.C_real = up, pre, self.src_real[i]
        naive = os.environ.get("HRV_CONV_IMPL", "mfma") == "naive"
        if cfg is None:
            cfg = lib.hrv_conv2d_pick_tile(N * Ho * Wo, self.Cout)
            if self.mixed:      # 128-byte-row tile with the least column padding (cfg 8: 128 columns, 9: 64)
                cfg = 8 if (self.Cout + 127) // 128 * 128 <= (self.Cout + 63) // 64 * 64 else 9
            if self.bf16 and cfg in (0, 6):
                # 128-byte K-tile rows: 128x128 tile +15-20 % (profiles/r01_conv_bench_bf16_rb.txt); the 128x64 tile
                # additionally stages its operands by LDS-DMA (profiles/r01_conv_bench_bf16_glds.txt)
                cfg = 8 if cfg == 0 else 9
            if self.bf16 and self.KH == 1 and self.KW == 1 and sum(self.src_pad) <= 128 and self.Cout % 64 == 0:
                # one or two K-tiles (conv_shared as a 1x1 over the 72 expanded taps): the block is all prologue and
                # epilogue, so the small 128x64 tile with 64-byte rows wins on resident blocks (0.39 vs 0.48 ms)
                cfg = 6
            pt = patch_tile(self.bf16, self.KH, self.KW, self.stride, self.pad, len(specs), up0, self.src_pad[0],
                            self.Cout, N, H, W)
            cfg = pt or cfg
        forced = os.environ.get("HRV_CONV_TILE") if (spade is None and not self.mixed) else None
        if forced is not None:
            cfg = int(forced)
        d.Cout, d.tile_cfg = self.Cout, cfg
        if naive:
            if self._w_dev is None:
                self._w_dev = self.w_cpu.to(self.device)
            d.w_oihw = self._w_dev.data_ptr()
        else:
            d.w_packed = self._get_packed(cfg).data_ptr()
        d.scale = None if self.scale is None else self.scale.data_ptr()
        d.shift = None if self.shift is None else self.shift.data_ptr()
        if residual is not None:
            assert (residual.N, residual.H, residual.W) == (N, Ho, Wo) and residual.C == self.Cout
            d.residual, d.res_cstride, d.res_coff = residual.t.data_ptr(), residual.cstride, residual.coff
        d.act, d.act_slope = self.act, self.slope
        assert (out.N, out.H, out.W) == (N, Ho << out_up, Wo << out_up) and out.C == oc, (self.name, out.t.shape, out.C)
        d.out, d.out_cstride, d.out_coff = out.t.data_ptr(), out.cstride, out.coff
        d.out_up_shift = out_up
        if spade is not None:
            d.spade = C.pointer(spade)
            mixed |= 4 if getattr(spade, "_x_f32", True) else 0
        d.mixed_flags = 15 if self.mixed else (mixed if self.bf16 else 0)
        if not naive and spade is None:
            need = lib.hrv_conv2d_workspace_bytes(C.byref(d))
            if need > 0:
                ws = _workspace(a0.t.device, need)
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        engine_bf16 = self.bf16 or self.mixed
        fn = lib.hrv_conv2d_naive_nhwc_f32 if naive else (lib.hrv_conv2d_nhwc_bf16 if engine_bf16 else lib.hrv_conv2d_nhwc_f32)
        assert not (naive and engine_bf16), "the naive cross-check is fp32"
        # algorithmic bytes: every source once, the output once (x4 for the fused 2x upsample store), the residual /
        # SPADE x once, the weights once
        nbytes = (sum(act_bytes(a) for a, _, _ in specs) + act_bytes(out) + act_bytes(residual) +
                  self.Cout * sum(self.src_real) * self.KH * self.KW * (2 if engine_bf16 else 4))
        if spade is not None:
            nbytes += N * Ho * Wo * oc * (4 if getattr(spade, "_x_f32", True) else 2)
        with _Timed("conv", self.name, self.flops(N, Ho, Wo), nbytes, f"conv_mfma_kernel[tile {d.tile_cfg}]"):
            _lib.check(fn(C.byref(d), _stream()), f"hrv_conv2d_nhwc_f32[{self.name}]")
        return out


class TapConvLayer:
    """KxK stride-1 'same' convolution with a tiny Cout (flow_conv: 768 -> 2), executed as
    a 1x1 convolution with KH*KW*Cout tap-channels on the MFMA engine + hrv_tapsum_nhwc_f32."""

    def __init__(self, weight: torch.Tensor, src_real: Sequence[int], device, bias: Optional[torch.Tensor] = None,
                 name: str = "tapconv", mma_bf16: bool = False):
        w = weight.detach().to("cpu", torch.float32)
        self.Cout, cin, self.KH, self.KW = w.shape
        assert self.KH == self.KW and self.KH % 2 == 1
        self.pad = self.KH // 2
        # [co][c][kh][kw] -> [(kh*KW+kw)*Cout + co][c][1][1]
        w1 = w.permute(2, 3, 0, 1).reshape(self.KH * self.KW * self.Cout, cin, 1, 1).contiguous()
        self.inner = ConvLayer(w1, src_real, device, stride=1, pad=0, name=name + "[taps-as-channels 1x1]",
                               mma_bf16=mma_bf16)
        self.bias = None if bias is None else bias.detach().to(device, torch.float32).contiguous()
        self.name = name

    def __call__(self, srcs, out: Act, residual: Optional[Act] = None) -> Act:
        lib = _lib.load()
        y = self.inner(srcs)
        res_ptr, rcs = (None, 0) if residual is None else (residual.t.data_ptr(), residual.cstride)
        if residual is not None:
            assert residual.coff == 0 and residual.C == self.Cout
        assert out.coff == 0 and out.C == self.Cout
        nbytes = 4.0 * y.N * y.H * y.W * (y.cstride + 2 * self.Cout)
        with _Timed("tapsum", self.name, 0.0, nbytes):
            _lib.check(lib.hrv_tapsum_nhwc_f32(y.t.data_ptr(), y.N, y.H, y.W, self.KH, self.KW, self.pad, self.Cout,
                                               y.cstride, None if self.bias is None else self.bias.data_ptr(),
                                               res_ptr, rcs, out.t.data_ptr(), out.cstride, _stream()),
                       "hrv_tapsum_nhwc_f32")
        return out


def instnorm_stats(a: Act, z: Optional[torch.Tensor] = None, noise_scale: Optional[torch.Tensor] = None,
                   eps: float = 1e-5):
    """hrv_instnorm_stats_nhwc_f32 -> (mean, rstd) float32 [N, Cp] for v = a + z[n,w,h]*noise_scale[c]."""
    lib = _lib.load()
    Cp = a.Cp
    dev = a.t.device
    ws = torch.empty(lib.hrv_instnorm_workspace_elems(a.N, a.H, a.W, Cp), dtype=torch.float32, device=dev)
    mean = torch.empty((a.N, Cp), dtype=torch.float32, device=dev)
    rstd = torch.empty((a.N, Cp), dtype=torch.float32, device=dev)
    if z is not None:
        assert z.is_contiguous() and z.numel() == a.N * a.W * a.H and noise_scale.numel() == Cp
    fn = lib.hrv_instnorm_stats_nhwc_bf16 if a.bf16 else lib.hrv_instnorm_stats_nhwc_f32
    with _Timed("stats", "instnorm_stats", 0.0, (2.0 if a.bf16 else 4.0) * a.N * a.H * a.W * Cp):
        _lib.check(fn(a.t.data_ptr(), a.N, a.H, a.W, Cp, a.cstride, a.coff,
                                                   None if z is None else z.data_ptr(),
                                                   None if z is None else noise_scale.data_ptr(), eps,
                                                   ws.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream()),
                   "hrv_instnorm_stats_nhwc_f32")
    return mean, rstd


def instnorm_stats2(a: Act, z_a: torch.Tensor, ns_a: torch.Tensor, z_b: torch.Tensor, ns_b: torch.Tensor, eps: float = 1e-5):
    """hrv_instnorm_stats2_nhwc_f32 -> ((mean, rstd), (mean, rstd)) for v = a + z*noise_scale of two norms over the same fp32
    ``a``: one pass over it (bit-identical to two instnorm_stats calls)."""
    lib = _lib.load()
    Cp = a.Cp
    dev = a.t.device
    assert not a.bf16 and z_a.is_contiguous() and z_b.is_contiguous() and ns_a.numel() == Cp and ns_b.numel() == Cp
    ws = torch.empty(2 * lib.hrv_instnorm_workspace_elems(a.N, a.H, a.W, Cp), dtype=torch.float32, device=dev)
    st = torch.empty((4, a.N, Cp), dtype=torch.float32, device=dev)
    if isinstance(a, ActUp):
        lo, hi = a.lo, a.hi
        with _Timed("stats", "instnorm_stats x2 [up]", 0.0, 4.0 * a.N * a.H * a.W * (lo.C / 4 + hi.C)):
            _lib.check(lib.hrv_instnorm_stats2_up_nhwc_f32(lo.t.data_ptr(), lo.cstride, lo.coff, lo.C, hi.t.data_ptr(), hi.cstride, hi.coff,
                                                           a.N, a.H, a.W, Cp, z_a.data_ptr(), ns_a.data_ptr(), z_b.data_ptr(), ns_b.data_ptr(),
                                                           eps, ws.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
                                                           st[3].data_ptr(), _stream()), "hrv_instnorm_stats2_up_nhwc_f32")
        return (st[0], st[1]), (st[2], st[3])
    with _Timed("stats", "instnorm_stats x2", 0.0, 4.0 * a.N * a.H * a.W * Cp):
        _lib.check(lib.hrv_instnorm_stats2_nhwc_f32(a.t.data_ptr(), a.N, a.H, a.W, Cp, a.cstride, a.coff, z_a.data_ptr(), ns_a.data_ptr(),
                                                    z_b.data_ptr(), ns_b.data_ptr(), eps, ws.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                                                    st[2].data_ptr(), st[3].data_ptr(), _stream()), "hrv_instnorm_stats2_nhwc_f32")
    return (st[0], st[1]), (st[2], st[3])


def instnorm_apply(a: Act, mean: torch.Tensor, rstd: torch.Tensor, act: int = ACT_NONE, slope: float = 0.2,
                   out: Optional[Act] = None) -> Act:
    lib = _lib.load()
    if out is None:
        out = alloc(a.N, a.H, a.W, a.C, a.t.device)
    with _Timed("apply", "instnorm_apply", 0.0, 8.0 * a.N * a.H * a.W * a.Cp):
        _lib.check(lib.hrv_instnorm_apply_nhwc_f32(a.t.data_ptr(), a.N, a.H, a.W, a.Cp, a.cstride, a.coff,
                                                   mean.data_ptr(), rstd.data_ptr(), act, slope, out.t.data_ptr(),
                                                   out.cstride, out.coff, _stream()), "hrv_instnorm_apply_nhwc_f32")
    return out


def avgpool3x3s2(a: Act) -> Act:
    lib = _lib.load()
    Ho, Wo = (a.H + 2 - 3) // 2 + 1, (a.W + 2 - 3) // 2 + 1
    out = alloc_written(a.N, Ho, Wo, a.C, a.t.device)
    with _Timed("pool", "avgpool3x3s2", 0.0, 4.0 * a.N * a.H * a.W * a.Cp * 1.25):
        _lib.check(lib.hrv_avgpool3x3s2_nhwc_f32(a.t.data_ptr(), a.N, a.H, a.W, a.Cp, a.cstride, a.coff,
                                                 out.t.data_ptr(), out.cstride, out.coff, _stream()),
                   "hrv_avgpool3x3s2_nhwc_f32")
    return out


class SpadeModulate:
    """Fused conv_gamma || conv_beta (128 -> 2C, 3x3) whose epilogue applies
    IN(x + noise) * (1 + gamma) + beta (+ LeakyReLU): network_generator.py:101-122,170-171."""

    def __init__(self, w_gamma, b_gamma, w_beta, b_beta, noise_scale, device, act: int, name: str, bf16: bool = False):
        wg = w_gamma.detach().to("cpu", torch.float32)
        wb = w_beta.detach().to("cpu", torch.float32)
        self.bf16 = bf16
        self.Creal = wg.shape[0]
        self.Cp = _cpad(self.Creal, bf16)
        G = (self.Creal + 31) // 32
        hid, k = wg.shape[1], wg.shape[2]
        w = torch.zeros(G * 64, hid, k, k)
        b = torch.zeros(G * 64)
        for g in range(G):
            n = min(32, self.Creal - g * 32)
            w[g * 64: g * 64 + n] = wg[g * 32: g * 32 + n]
            w[g * 64 + 32: g * 64 + 32 + n] = wb[g * 32: g * 32 + n]
            b[g * 64: g * 64 + n] = b_gamma.detach().cpu().float()[g * 32: g * 32 + n]
            b[g * 64 + 32: g * 64 + 32 + n] = b_beta.detach().cpu().float()[g * 32: g * 32 + n]
        self._w_raw = (wg, wb)                  # the dedicated kernel packs the pair itself (lazily, on first use)
        self._b_raw = (b_gamma.detach().cpu().float().contiguous(), b_beta.detach().cpu().float().contiguous())
        self._gb = None
        self.conv = ConvLayer(w, [hid], device, shift=b, stride=1, pad=k // 2, act=act, name=name, bf16=bf16)
        self.conv.flops_cout = 2 * self.Creal   # algorithmic (unpadded) gamma+beta columns
        # 128x128 tile when the pair count is even (bf16: the 128-byte-row variant), else 128x64
        self.cfg = (8 if bf16 else 0) if (G % 2 == 0) else (9 if bf16 else 6)
        ns = torch.zeros(self.Cp)
        ns[: self.Creal] = noise_scale.detach().cpu().float()
        self.ns = ns.to(device)
        self.has_noise = bool((ns != 0).any().item())

    def flops(self, N, H, W):
        return 2.0 * N * H * W * 2 * self.Creal * self.conv.w_cpu.shape[1] * self.conv.KH * self.conv.KW

    def __call__(self, actv: Act, x: Act, mean, rstd, z: Optional[torch.Tensor], out: Optional[Act] = None) -> Act:
        assert x.C == self.Creal, (self.conv.name, x.C, self.Creal)
        e = _lib.hrv_spade_epi_t()
        e._x_f32 = not x.bf16
        e.x, e.x_cstride, e.x_coff, e.C = x.t.data_ptr(), x.cstride, x.coff, self.Cp
        e.mean, e.rstd = mean.data_ptr(), rstd.data_ptr()
        use_noise = z is not None and self.has_noise
        e.noise_z = z.data_ptr() if use_noise else None
        e.noise_scale = self.ns.data_ptr() if use_noise else None
        if out is None:
            out = alloc(x.N, x.H, x.W, self.Creal, x.t.device, self.bf16)
        if self.bf16 and actv.bf16 and out.bf16 and self.Cp == self.Creal:
            from . import train_ops as T       # (train_ops imports this module)
            if T.spade_gb_ok(0, self.Creal, self.Cp, actv.C, x.N, x.H, x.W):
                # the dedicated gamma|beta kernel (csrc/spade_gb.hip): 16x16-pixel tiles x all columns, no padded columns
                if self._gb is None:
                    dev = x.t.device
                    wg, wb = self._w_raw
                    self._gb = (T.spade_gb_pack(0, wg.to(dev).contiguous(), wb.to(dev).contiguous()),
                                self._b_raw[0].to(dev), self._b_raw[1].to(dev))
                pk, bg, bb = self._gb
                nb = act_bytes(actv) + act_bytes(x) + act_bytes(out) + 2.0 * self.Creal * actv.C * 9 * 2
                T.spade_gb_forward(actv, x, mean, rstd, z if use_noise else None, self.ns if use_noise else None, pk, bg, bb,
                                   self.conv.act, self.conv.slope, out, None, self.conv.name, self.flops(x.N, x.H, x.W), nb)
                return out
        cfg = self.cfg
        cfg = patch_tile(self.bf16, 3, 3, 1, 1, 1, 0, actv.Cp, self.conv.Cout, x.N, x.H, x.W, wide=True) or cfg
        return self.conv([actv], out=out, spade=e, out_channels=self.Creal, cfg=cfg)


def tap_expand(a: Act, down: int, k: int = 3) -> Act:
    """hrv_tap_expand_nhwc: [N,Hs,Ws,C] (Hs = H << down) -> [N,H,W,k*k*Cp], tap-major (t*Cp + c), zero borders."""
    lib = _lib.load()
    es = 2 if a.bf16 else 4
    Cp = a.Cp
    assert (Cp * es) % 16 == 0 and (a.cstride * es) % 16 == 0 and (a.coff * es) % 16 == 0
    H, W = a.H >> down, a.W >> down
    out = torch.empty((a.N, H, W, k * k * Cp), dtype=a.t.dtype, device=a.t.device)
    with _Timed("glue", "tap_expand", 0.0, float(out.numel() * es) * 1.2):
        _lib.check(lib.hrv_tap_expand_nhwc(a.t.data_ptr(), a.N, H, W, Cp * es // 16, a.cstride * es // 16,
                                           a.coff * es // 16, down, k, k, k // 2, out.data_ptr(), _stream()),
                   "hrv_tap_expand_nhwc")
    return Act(out, k * k * Cp)


def resize_bilinear(a: Act, Ho: int, Wo: int, rh: float, rw: float, addend: Optional[Act] = None,
                    out: Optional[Act] = None) -> Act:
    """hrv_resize_bilinear_nhwc_f32: out = bilinear(a) (+ addend)."""
    lib = _lib.load()
    if out is None:
        out = alloc_written(a.N, Ho, Wo, a.C, a.t.device)      # (the kernel runs over a.Cp channels)
    add_ptr, acs, aco = (None, 0, 0) if addend is None else (addend.t.data_ptr(), addend.cstride, addend.coff)
    nbytes = 4.0 * a.N * Ho * Wo * a.Cp * (2 if addend is None else 3)
    with _Timed("resize", "bilinear", 0.0, nbytes):
        _lib.check(lib.hrv_resize_bilinear_nhwc_f32(a.t.data_ptr(), a.N, a.H, a.W, a.Cp, a.cstride, a.coff, Ho, Wo,
                                                    rh, rw, add_ptr, acs, aco, out.t.data_ptr(), out.cstride,
                                                    out.coff, _stream()), "hrv_resize_bilinear_nhwc_f32")
    return out


def resize_nearest(a: Act, Ho: int, Wo: int, addend: Optional[Act] = None, out: Optional[Act] = None) -> Act:
    """hrv_resize_nearest_nhwc_f32: out = F.interpolate(a, mode='nearest') (+ addend) -- networks.py:130-131 with upsample='nearest'."""
    lib = _lib.load()
    if out is None:
        out = alloc_written(a.N, Ho, Wo, a.C, a.t.device)
    add_ptr, acs, aco = (None, 0, 0) if addend is None else (addend.t.data_ptr(), addend.cstride, addend.coff)
    with _Timed("resize", "nearest", 0.0, 4.0 * a.N * Ho * Wo * a.Cp * (2 if addend is None else 3)):
        _lib.check(lib.hrv_resize_nearest_nhwc_f32(a.t.data_ptr(), a.N, a.H, a.W, a.Cp, a.cstride, a.coff, Ho, Wo, add_ptr, acs, aco,
                                                   out.t.data_ptr(), out.cstride, out.coff, _stream()), "hrv_resize_nearest_nhwc_f32")
    return out


def resize_nearest_dense(t: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """the same on a dense [N,h,w,C] tensor of any C (the 2-channel flows, networks.py:133,150)"""
    lib = _lib.load()
    N, H, W, Cc = t.shape
    out = torch.empty((N, Ho, Wo, Cc), dtype=torch.float32, device=t.device)
    _lib.check(lib.hrv_resize_nearest_nhwc_f32(t.data_ptr(), N, H, W, Cc, Cc, 0, Ho, Wo, None, 0, 0, out.data_ptr(), Cc, 0, _stream()),
               "hrv_resize_nearest_nhwc_f32")
    return out


def flow_warp(src: Act, flow: torch.Tensor, Ho: int, Wo: int, rh: float, rw: float, norm_x: float, norm_y: float,
              out: Optional[Act] = None, want_flow_up: bool = True):
    """hrv_flow_warp_nhwc_f32.  ``flow`` is [N,fh,fw,2] (reference layout).
    Returns (warped Act [N,Ho,Wo,C], flow_up tensor [N,Ho,Wo,2] or None)."""
    lib = _lib.load()
    require_cuda(flow, "flow_warp")
    assert flow.is_contiguous() and flow.shape[-1] == 2 and flow.shape[0] == src.N
    if out is None:
        out = alloc(src.N, Ho, Wo, src.C, src.t.device)
    fup = torch.empty((src.N, Ho, Wo, 2), dtype=torch.float32, device=flow.device) if want_flow_up else None
    d = _lib.hrv_flow_warp_t()
    d.src, d.N, d.H, d.W, d.C = src.t.data_ptr(), src.N, src.H, src.W, src.Cp
    d.src_cstride, d.src_coff = src.cstride, src.coff
    d.flow, d.fh, d.fw, d.Ho, d.Wo = flow.data_ptr(), flow.shape[1], flow.shape[2], Ho, Wo
    d.rh, d.rw, d.norm_x, d.norm_y = rh, rw, norm_x, norm_y
    d.out, d.out_cstride, d.out_coff = out.t.data_ptr(), out.cstride, out.coff
    d.flow_up = None if fup is None else fup.data_ptr()
    nbytes = 4.0 * src.N * Ho * Wo * src.Cp * 2
    with _Timed("warp", "flow_warp", 0.0, nbytes):
        _lib.check(lib.hrv_flow_warp_nhwc_f32(C.byref(d), _stream()), "hrv_flow_warp_nhwc_f32")
    return out, fup
