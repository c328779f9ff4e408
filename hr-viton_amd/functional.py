"""HIP replacements of the torch.nn.functional calls the reference's training scripts make
between the networks (train_condition.py:164-252, train_generator.py:235-238):

    grid_sample(input, grid, padding_mode='border')     F.grid_sample       (bilinear, align_corners=False)
    interpolate(x, size=/scale_factor=, mode='bilinear' | 'nearest')   F.interpolate
    softmax(x, dim=1)                                   F.softmax / torch.softmax
    cross_entropy2d(input, target)                      utils.cross_entropy2d (utils.py:29-42)
    tv_loss(flow)                                       the |d/dy| + |d/dx| means of train_condition.py:190-199

Same argument meaning as the torch functions, NCHW fp32 CUDA tensors in and out.  Every function is an operator
registered with ``torch.library`` in the ``hrviton`` namespace (``torch.ops.hrviton.grid_sample`` ...; CUDA dispatch key
only -- there is no CPU kernel to fall back to), with its backward registered as the autograd formula; forward and
backward are single launches of the kernels in csrc/cond_train.hip / csrc/glue.hip, reached through the C ABI
(ctypes is the transport, the dispatcher is the boundary).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch

from . import _lib, ops
from .ops import _stream


def _f32c(t: torch.Tensor, what: str) -> torch.Tensor:
    ops.require_cuda(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: float32 expected, got {t.dtype}")
    return t.contiguous()


def _op(name):
    return torch.library.custom_op(f"hrviton::{name}", mutates_args=(), device_types="cuda")


# ---------------------------------------------------------------- grid_sample
@_op("grid_sample")
def _grid_sample_op(inp: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    inp, grid = _f32c(inp, "grid_sample(input)"), _f32c(grid, "grid_sample(grid)")
    N, Cc, H, W = inp.shape
    Ho, Wo = grid.shape[1], grid.shape[2]
    assert grid.shape[0] == N and grid.shape[3] == 2, "grid must be [N,Ho,Wo,2]"
    out = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=inp.device)
    _lib.check(lib.hrv_grid_sample_nchw_f32(inp.data_ptr(), N, Cc, H, W, grid.data_ptr(), Ho, Wo, out.data_ptr(),
                                            _stream()), "hrv_grid_sample_nchw_f32")
    return out


@_grid_sample_op.register_fake
def _(inp, grid):
    return inp.new_empty((inp.shape[0], inp.shape[1], grid.shape[1], grid.shape[2]))


@_op("grid_sample_backward")
def _grid_sample_bwd_op(inp: torch.Tensor, grid: torch.Tensor, dout: torch.Tensor, need_input: bool,
                        need_grid: bool) -> List[torch.Tensor]:
    lib = _lib.load()
    inp, grid, dout = inp.contiguous(), grid.contiguous(), dout.contiguous()
    N, Cc, H, W = inp.shape
    Ho, Wo = grid.shape[1], grid.shape[2]
    din = torch.zeros_like(inp) if need_input else inp.new_empty(0)
    dgrid = torch.empty_like(grid) if need_grid else grid.new_empty(0)
    if need_input or need_grid:
        _lib.check(lib.hrv_grid_sample_nchw_bwd_f32(inp.data_ptr(), N, Cc, H, W, grid.data_ptr(), Ho, Wo, dout.data_ptr(),
                                                    din.data_ptr() if need_input else None,
                                                    dgrid.data_ptr() if need_grid else None, _stream()),
                   "hrv_grid_sample_nchw_bwd_f32")
    return [din, dgrid]


@_grid_sample_bwd_op.register_fake
def _(inp, grid, dout, need_input, need_grid):
    return [torch.empty_like(inp) if need_input else inp.new_empty(0),
            torch.empty_like(grid) if need_grid else grid.new_empty(0)]


def _gs_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _gs_backward(ctx, dout):
    inp, grid = ctx.saved_tensors
    ni, ng = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    din, dgrid = _grid_sample_bwd_op(inp, grid, dout, ni, ng)
    return (din if ni else None), (dgrid if ng else None)


_grid_sample_op.register_autograd(_gs_backward, setup_context=_gs_setup)


def grid_sample(input: torch.Tensor, grid: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
                align_corners: Optional[bool] = None) -> torch.Tensor:
    """F.grid_sample for the configuration every call site of the reference uses (padding_mode='border',
    bilinear, align_corners False/None): train_condition.py:244-245, train_generator.py:237-238."""
    if mode != "bilinear" or padding_mode != "border" or align_corners:
        raise NotImplementedError("hr-viton_amd grid_sample: bilinear / padding_mode='border' / align_corners=False only")
    _f32c(input, "grid_sample(input)"), _f32c(grid, "grid_sample(grid)")
    return _grid_sample_op(input, grid)


# ---------------------------------------------------------------- interpolate
@_op("interpolate_bilinear")
def _interp_op(x: torch.Tensor, Ho: int, Wo: int, rh: float, rw: float) -> torch.Tensor:
    lib = _lib.load()
    x = _f32c(x, "interpolate")
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    # N*C planes; the kernel's source ratio is in/out, which equals 1/scale_factor for the integer
    # factors of the path (asserted by interpolate())
    _lib.check(lib.hrv_resize_nchw_f32(x.data_ptr(), N * Cc, H, W, Ho, Wo, 0, out.data_ptr(), _stream()),
               "hrv_resize_nchw_f32")
    return out


@_interp_op.register_fake
def _(x, Ho, Wo, rh, rw):
    return x.new_empty((x.shape[0], x.shape[1], Ho, Wo))


@_op("interpolate_bilinear_backward")
def _interp_bwd_op(dout: torch.Tensor, H: int, W: int, rh: float, rw: float) -> torch.Tensor:
    lib = _lib.load()
    dout = dout.contiguous()
    N, Cc, Ho, Wo = dout.shape
    dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dout.device)
    _lib.check(lib.hrv_resize_bilinear_bwd_nhwc_f32(dout.data_ptr(), N * Cc, Ho, Wo, 1, 1, 0, rh, rw, dx.data_ptr(), H,
                                                    W, 1, 0, 0, _stream()), "hrv_resize_bilinear_bwd_nhwc_f32")
    return dx


@_interp_bwd_op.register_fake
def _(dout, H, W, rh, rw):
    return dout.new_empty((dout.shape[0], dout.shape[1], H, W))


def _interp_setup(ctx, inputs, output):
    x, _Ho, _Wo, rh, rw = inputs
    ctx.geom = (x.shape[2], x.shape[3], rh, rw)


def _interp_backward(ctx, dout):
    H, W, rh, rw = ctx.geom
    return _interp_bwd_op(dout, H, W, rh, rw), None, None, None, None


_interp_op.register_autograd(_interp_backward, setup_context=_interp_setup)


@_op("interpolate_nearest")
def _interp_nearest_op(x: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    lib = _lib.load()
    x = _f32c(x, "interpolate")
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    _lib.check(lib.hrv_resize_nearest_nchw_f32(x.data_ptr(), N * Cc, H, W, Ho, Wo, out.data_ptr(), _stream()), "hrv_resize_nearest_nchw_f32")
    return out


@_interp_nearest_op.register_fake
def _(x, Ho, Wo):
    return x.new_empty((x.shape[0], x.shape[1], Ho, Wo))


@_op("interpolate_nearest_backward")
def _interp_nearest_bwd_op(dout: torch.Tensor, H: int, W: int) -> torch.Tensor:
    lib = _lib.load()
    dout = dout.contiguous()
    N, Cc, Ho, Wo = dout.shape
    dx = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dout.device)
    _lib.check(lib.hrv_resize_nearest_nchw_bwd_f32(dout.data_ptr(), N * Cc, Ho, Wo, H, W, dx.data_ptr(), _stream()),
               "hrv_resize_nearest_nchw_bwd_f32")
    return dx


@_interp_nearest_bwd_op.register_fake
def _(dout, H, W):
    return dout.new_empty((dout.shape[0], dout.shape[1], H, W))


def _interp_nearest_setup(ctx, inputs, output):
    ctx.geom = (inputs[0].shape[2], inputs[0].shape[3])


def _interp_nearest_backward(ctx, dout):
    return _interp_nearest_bwd_op(dout, *ctx.geom), None, None


_interp_nearest_op.register_autograd(_interp_nearest_backward, setup_context=_interp_nearest_setup)


def interpolate(input: torch.Tensor, size: Optional[Union[int, Sequence[int]]] = None,
                scale_factor: Optional[float] = None, mode: str = "bilinear",
                align_corners: Optional[bool] = None) -> torch.Tensor:
    """F.interpolate(mode='bilinear', align_corners=False) -- SURVEY App. A.2: with scale_factor the given
    factor is the source ratio, with size= it is in/out -- and mode='nearest' with size= (train_condition.py:242 under
    --upsample nearest)."""
    if mode == "nearest" and not align_corners:
        if size is None:
            raise NotImplementedError("hr-viton_amd interpolate(mode='nearest'): size= form only (train_condition.py:242)")
        Ho, Wo = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        _f32c(input, "interpolate")
        return _interp_nearest_op(input, Ho, Wo)
    if mode != "bilinear" or align_corners:
        raise NotImplementedError("hr-viton_amd interpolate: mode='bilinear' | 'nearest', align_corners=False only")
    H, W = input.shape[2], input.shape[3]
    if size is not None:
        Ho, Wo = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        rh, rw = H / Ho, W / Wo
    else:
        Ho, Wo = int(H * scale_factor), int(W * scale_factor)
        if Ho != H * scale_factor or Wo != W * scale_factor:
            raise NotImplementedError("hr-viton_amd interpolate: scale_factor must give integer output sizes")
        rh = rw = 1.0 / scale_factor
    _f32c(input, "interpolate")
    return _interp_op(input, Ho, Wo, float(rh), float(rw))


# ---------------------------------------------------------------- softmax over channels
@_op("softmax2d")
def _softmax_op(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = _f32c(x, "softmax")
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    y = torch.empty_like(x)
    _lib.check(lib.hrv_softmax_nchw_f32(x.data_ptr(), N, Cc, HW, y.data_ptr(), _stream()), "hrv_softmax_nchw_f32")
    return y


@_softmax_op.register_fake
def _(x):
    return torch.empty_like(x)


@_op("softmax2d_backward")
def _softmax_bwd_op(y: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    y, dy = y.contiguous(), dy.contiguous()
    N, Cc = y.shape[0], y.shape[1]
    HW = y.numel() // (N * Cc)
    dx = torch.empty_like(y)
    _lib.check(lib.hrv_softmax_nchw_bwd_f32(y.data_ptr(), dy.data_ptr(), N, Cc, HW, dx.data_ptr(), _stream()),
               "hrv_softmax_nchw_bwd_f32")
    return dx


@_softmax_bwd_op.register_fake
def _(y, dy):
    return torch.empty_like(y)


def _softmax_setup(ctx, inputs, output):
    ctx.save_for_backward(output)


def _softmax_backward(ctx, dy):
    (y,) = ctx.saved_tensors
    return _softmax_bwd_op(y, dy)


_softmax_op.register_autograd(_softmax_backward, setup_context=_softmax_setup)


def softmax(input: torch.Tensor, dim: int = 1) -> torch.Tensor:
    if dim != 1 or input.dim() != 4:
        raise NotImplementedError("hr-viton_amd softmax: channel dim of an NCHW tensor only")
    _f32c(input, "softmax")
    return _softmax_op(input)


# ---------------------------------------------------------------- losses (value + gradient in ONE launch)
def _scale_saved(g: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """g *= s[0] in place (g is the op's own saved (softmax - onehot) / sign buffer, consumed once)."""
    lib = _lib.load()
    _lib.check(lib.hrv_scale_f32(g.data_ptr(), g.numel(), 1.0, s.data_ptr(), _stream()), "hrv_scale_f32")
    return g


@_op("cross_entropy2d")
def _ce_op(x: torch.Tensor, target: torch.Tensor, with_grad: bool) -> List[torch.Tensor]:
    """-> [loss (0-d), count (1), grad (x's shape, or empty)]; grad holds (softmax - onehot), 1/count is applied in
    backward (count stays on the device)."""
    lib = _lib.load()
    x = _f32c(x, "cross_entropy2d(input)")
    target = target.contiguous().to(torch.int64)
    N, Cc, H, W = x.shape
    assert tuple(target.shape) == (N, H, W), "target must be [N,H,W] (sizes equal: utils.py:34-35 is not on the path)"
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    ws = torch.empty(1024, dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x) if with_grad else x.new_empty(0)
    _lib.check(lib.hrv_cross_entropy_nchw_f32(x.data_ptr(), target.data_ptr(), N, Cc, H * W, 1.0,
                                              grad.data_ptr() if with_grad else None, ws.data_ptr(),
                                              out.data_ptr(), _stream()), "hrv_cross_entropy_nchw_f32")
    return [out[0].clone(), out[1:2].clone(), grad]


@_ce_op.register_fake
def _(x, target, with_grad):
    return [x.new_empty(()), x.new_empty(1), torch.empty_like(x) if with_grad else x.new_empty(0)]


def _ce_setup(ctx, inputs, output):
    ctx.with_grad = inputs[2]
    ctx.count, ctx.grad = output[1].detach(), output[2].detach()     # detached: no node <-> output cycle


def _ce_backward(ctx, grads):
    g, ctx.grad = ctx.grad, None
    if not ctx.with_grad or g is None:
        return None, None, None
    s = (grads[0].reshape(1).to(torch.float32) / ctx.count.clamp_min(1.0)).contiguous()   # 1-element device scalar
    return _scale_saved(g, s), None, None


_ce_op.register_autograd(_ce_backward, setup_context=_ce_setup)


def cross_entropy2d(input: torch.Tensor, target: torch.Tensor, weight=None, size_average: bool = True) -> torch.Tensor:
    """utils.cross_entropy2d (utils.py:29-42): mean softmax cross entropy over the pixels, ignore_index=250."""
    if weight is not None or not size_average:
        raise NotImplementedError("hr-viton_amd cross_entropy2d: weight=None, size_average=True (the reference's call)")
    _f32c(input, "cross_entropy2d(input)")
    if not target.is_cuda:
        raise _lib.HrvError("cross_entropy2d(target): tensor is on the CPU; the MI355X path has no CPU fallback")
    return _ce_op(input, target, bool(input.requires_grad and torch.is_grad_enabled()))[0]


@_op("tv_loss")
def _tv_op(flow: torch.Tensor, with_grad: bool) -> List[torch.Tensor]:
    lib = _lib.load()
    flow = _f32c(flow, "tv_loss")
    N, H, W, two = flow.shape
    assert two == 2, "flow must be [N,h,w,2]"
    out = torch.empty(1, dtype=torch.float32, device=flow.device)
    ws = torch.empty(1024, dtype=torch.float32, device=flow.device)
    grad = torch.empty_like(flow) if with_grad else flow.new_empty(0)
    _lib.check(lib.hrv_tv_loss_f32(flow.data_ptr(), N, H, W, grad.data_ptr() if with_grad else None, ws.data_ptr(),
                                   out.data_ptr(), _stream()), "hrv_tv_loss_f32")
    return [out.reshape(()), grad]


@_tv_op.register_fake
def _(flow, with_grad):
    return [flow.new_empty(()), torch.empty_like(flow) if with_grad else flow.new_empty(0)]


def _tv_setup(ctx, inputs, output):
    ctx.with_grad = inputs[1]
    ctx.grad = output[1].detach()


def _tv_backward(ctx, grads):
    g, ctx.grad = ctx.grad, None
    if not ctx.with_grad or g is None:
        return None, None
    return _scale_saved(g, grads[0].reshape(1).to(torch.float32).contiguous()), None


_tv_op.register_autograd(_tv_backward, setup_context=_tv_setup)


def tv_loss(flow: torch.Tensor) -> torch.Tensor:
    """mean|flow[:,1:]-flow[:,:-1]| + mean|flow[:,:,1:]-flow[:,:,:-1]| (train_condition.py:192-199)."""
    _f32c(flow, "tv_loss")
    return _tv_op(flow, bool(flow.requires_grad and torch.is_grad_enabled()))[0]


REGISTERED_OPS = ("grid_sample", "grid_sample_backward", "interpolate_bilinear", "interpolate_bilinear_backward",
                  "softmax2d", "softmax2d_backward", "cross_entropy2d", "tv_loss")
