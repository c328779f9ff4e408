"""HIP replacements of the torch.nn.functional calls the reference's training scripts make
between the networks (train_condition.py:164-252, train_generator.py:235-238):

    grid_sample(input, grid, padding_mode='border')     F.grid_sample       (bilinear, align_corners=False)
    interpolate(x, size=/scale_factor=, mode='bilinear')F.interpolate
    softmax(x, dim=1)                                   F.softmax / torch.softmax
    cross_entropy2d(input, target)                      utils.cross_entropy2d (utils.py:29-42)
    tv_loss(flow)                                       the |d/dy| + |d/dx| means of train_condition.py:190-199

Same argument meaning as the torch functions, NCHW fp32 CUDA tensors in and out, each one a
``torch.autograd.Function`` whose forward and backward are single launches of the kernels in
csrc/cond_train.hip / csrc/glue.hip through the C ABI.  No CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import torch

from . import _lib, ops
from .ops import _stream


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    ops.require_cuda(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: float32 expected, got {t.dtype}")
    return t.contiguous()


class _GridSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, grid):
        lib = _lib.load()
        inp, grid = _f32c(inp, "grid_sample(input)"), _f32c(grid, "grid_sample(grid)")
        N, Cc, H, W = inp.shape
        Ho, Wo = grid.shape[1], grid.shape[2]
        assert grid.shape[0] == N and grid.shape[3] == 2, "grid must be [N,Ho,Wo,2]"
        out = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=inp.device)
        _lib.check(lib.hrv_grid_sample_nchw_f32(inp.data_ptr(), N, Cc, H, W, grid.data_ptr(), Ho, Wo, out.data_ptr(),
                                                _stream()), "hrv_grid_sample_nchw_f32")
        ctx.save_for_backward(inp, grid)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        inp, grid = ctx.saved_tensors
        N, Cc, H, W = inp.shape
        Ho, Wo = grid.shape[1], grid.shape[2]
        dout = dout.contiguous()
        din = torch.zeros_like(inp) if ctx.needs_input_grad[0] else None
        dgrid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        if din is None and dgrid is None:
            return None, None
        _lib.check(lib.hrv_grid_sample_nchw_bwd_f32(inp.data_ptr(), N, Cc, H, W, grid.data_ptr(), Ho, Wo, dout.data_ptr(),
                                                    None if din is None else din.data_ptr(),
                                                    None if dgrid is None else dgrid.data_ptr(), _stream()),
                   "hrv_grid_sample_nchw_bwd_f32")
        return din, dgrid


def grid_sample(input: torch.Tensor, grid: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
                align_corners: Optional[bool] = None) -> torch.Tensor:
    """F.grid_sample for the configuration every call site of the reference uses (padding_mode='border',
    bilinear, align_corners False/None): train_condition.py:244-245, train_generator.py:237-238."""
    if mode != "bilinear" or padding_mode != "border" or align_corners:
        raise NotImplementedError("hr-viton_amd grid_sample: bilinear / padding_mode='border' / align_corners=False only")
    return _GridSampleFn.apply(input, grid)


class _InterpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, rh, rw):
        lib = _lib.load()
        x = _f32c(x, "interpolate")
        N, Cc, H, W = x.shape
        out = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
        # N*C planes; the kernel's source ratio is in/out, which equals 1/scale_factor for the integer
        # factors of the path (asserted by interpolate())
        _lib.check(lib.hrv_resize_nchw_f32(x.data_ptr(), N * Cc, H, W, Ho, Wo, 0, out.data_ptr(), _stream()),
                   "hrv_resize_nchw_f32")
        ctx.geom = (N, Cc, H, W, Ho, Wo, rh, rw)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        N, Cc, H, W, Ho, Wo, rh, rw = ctx.geom
        dout = dout.contiguous()
        dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dout.device)
        _lib.check(lib.hrv_resize_bilinear_bwd_nhwc_f32(dout.data_ptr(), N * Cc, Ho, Wo, 1, 1, 0, rh, rw, dx.data_ptr(), H,
                                                        W, 1, 0, 0, _stream()), "hrv_resize_bilinear_bwd_nhwc_f32")
        return dx, None, None, None, None


def interpolate(input: torch.Tensor, size: Optional[Union[int, Sequence[int]]] = None,
                scale_factor: Optional[float] = None, mode: str = "bilinear",
                align_corners: Optional[bool] = None) -> torch.Tensor:
    """F.interpolate(mode='bilinear', align_corners=False) -- SURVEY App. A.2: with scale_factor the given
    factor is the source ratio, with size= it is in/out."""
    if mode != "bilinear" or align_corners:
        raise NotImplementedError("hr-viton_amd interpolate: mode='bilinear', align_corners=False only")
    H, W = input.shape[2], input.shape[3]
    if size is not None:
        Ho, Wo = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        rh, rw = H / Ho, W / Wo
    else:
        Ho, Wo = int(H * scale_factor), int(W * scale_factor)
        if Ho != H * scale_factor or Wo != W * scale_factor:
            raise NotImplementedError("hr-viton_amd interpolate: scale_factor must give integer output sizes")
        rh = rw = 1.0 / scale_factor
    return _InterpFn.apply(input, Ho, Wo, rh, rw)


class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _f32c(x, "softmax")
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        y = torch.empty_like(x)
        _lib.check(lib.hrv_softmax_nchw_f32(x.data_ptr(), N, Cc, HW, y.data_ptr(), _stream()), "hrv_softmax_nchw_f32")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        N, Cc = y.shape[0], y.shape[1]
        HW = y.numel() // (N * Cc)
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.check(lib.hrv_softmax_nchw_bwd_f32(y.data_ptr(), dy.data_ptr(), N, Cc, HW, dx.data_ptr(), _stream()),
                   "hrv_softmax_nchw_bwd_f32")
        return dx


def softmax(input: torch.Tensor, dim: int = 1) -> torch.Tensor:
    if dim != 1 or input.dim() != 4:
        raise NotImplementedError("hr-viton_amd softmax: channel dim of an NCHW tensor only")
    return _SoftmaxFn.apply(input)


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target):
        lib = _lib.load()
        x = _f32c(x, "cross_entropy2d(input)")
        if not target.is_cuda:
            raise _lib.HrvError("cross_entropy2d(target): tensor is on the CPU; the MI355X path has no CPU fallback")
        target = target.contiguous().to(torch.int64)
        N, Cc, H, W = x.shape
        assert tuple(target.shape) == (N, H, W), "target must be [N,H,W] (sizes equal: utils.py:34-35 is not on the path)"
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = torch.empty(1024, dtype=torch.float32, device=x.device)
        need = ctx.needs_input_grad[0]
        grad = torch.empty_like(x) if need else None
        # grad holds (softmax - onehot); 1/count is applied in backward (count stays on the device)
        _lib.check(lib.hrv_cross_entropy_nchw_f32(x.data_ptr(), target.data_ptr(), N, Cc, H * W, 1.0,
                                                  None if grad is None else grad.data_ptr(), ws.data_ptr(),
                                                  out.data_ptr(), _stream()), "hrv_cross_entropy_nchw_f32")
        ctx.grad, ctx.count = grad, out[1:2]
        return out[0]

    @staticmethod
    def backward(ctx, g_out):
        g = ctx.grad
        ctx.grad = None
        if g is None:
            return None, None
        lib = _lib.load()
        s = (g_out.reshape(1).to(torch.float32) / ctx.count.clamp_min(1.0)).contiguous()   # 1-element device scalar
        _lib.check(lib.hrv_scale_f32(g.data_ptr(), g.numel(), 1.0, s.data_ptr(), _stream()), "hrv_scale_f32")
        return g, None


def cross_entropy2d(input: torch.Tensor, target: torch.Tensor, weight=None, size_average: bool = True) -> torch.Tensor:
    """utils.cross_entropy2d (utils.py:29-42): mean softmax cross entropy over the pixels, ignore_index=250."""
    if weight is not None or not size_average:
        raise NotImplementedError("hr-viton_amd cross_entropy2d: weight=None, size_average=True (the reference's call)")
    return _CEFn.apply(input, target)


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow):
        lib = _lib.load()
        flow = _f32c(flow, "tv_loss")
        N, H, W, two = flow.shape
        assert two == 2, "flow must be [N,h,w,2]"
        out = torch.empty(1, dtype=torch.float32, device=flow.device)
        ws = torch.empty(1024, dtype=torch.float32, device=flow.device)
        grad = torch.empty_like(flow) if ctx.needs_input_grad[0] else None
        _lib.check(lib.hrv_tv_loss_f32(flow.data_ptr(), N, H, W, None if grad is None else grad.data_ptr(), ws.data_ptr(),
                                       out.data_ptr(), _stream()), "hrv_tv_loss_f32")
        ctx.grad = grad
        return out[0]

    @staticmethod
    def backward(ctx, g_out):
        g = ctx.grad
        ctx.grad = None
        if g is None:
            return None
        lib = _lib.load()
        s = g_out.reshape(1).to(torch.float32).contiguous()
        _lib.check(lib.hrv_scale_f32(g.data_ptr(), g.numel(), 1.0, s.data_ptr(), _stream()), "hrv_scale_f32")
        return g


def tv_loss(flow: torch.Tensor) -> torch.Tensor:
    """mean|flow[:,1:]-flow[:,:-1]| + mean|flow[:,:,1:]-flow[:,:,:-1]| (train_condition.py:192-199)."""
    return _TVFn.apply(flow)
