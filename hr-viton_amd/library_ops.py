"""The tensor-level operators of the HIP path as ``torch.library`` custom ops (namespace ``hrviton``, CUDA dispatch key
only) -- the operator list SURVEY.md 8(b) asks of a C-ABI replacement under the reference's classes:

    conv2d_nhwc_fwd / conv2d_nhwc_dgrad / conv2d_nhwc_wgrad     nn.Conv2d forward, data and weight (+ bias) gradient
    instnorm_stats                                               InstanceNorm2d statistics (network_generator.py:104-110)
    spade_gamma_beta_fused_fwd / _bwd                            conv_gamma || conv_beta + modulate (:117-121) / its data gradient
    spade_modulate_bwd                                           SPADE / InstanceNorm backward (dx, [dgamma | dbeta], d noise_scale)
    spectral_sigma                                               torch SpectralNorm.compute_weight's power iteration + sigma
    loss_reduce                                                  L1 / hinge / -mean / MSE value + gradient in one pass
    fused_adam                                                   torch.optim.Adam's update over a flat buffer

(The script-level functional ops -- grid_sample, interpolate, softmax, cross_entropy2d, tv_loss -- are registered by
functional.py.)  Tensors are dense NHWC ([N, H, W, C], C % 4 == 0 for fp32, % 8 for bf16) like everything under the
module boundary; the module classes (networks.py / network_generator.py of this package) drive the same kernels through
their hand-written forward / backward plans, these ops expose them to the dispatcher: schema + fake-tensor checks,
``torch.ops.hrviton.*`` call sites, ``torch.library.opcheck``.  ctypes is the transport, no compute happens in torch.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from . import train_ops as T
from .ops import Act


def _op(name, mutates=()):
    return torch.library.custom_op(f"hrviton::{name}", mutates_args=mutates, device_types="cuda")


def _act(t: torch.Tensor, what: str) -> Act:
    ops.require_cuda(t, what)
    if t.dim() != 4 or not t.is_contiguous():
        raise ValueError(f"{what}: a dense NHWC tensor [N, H, W, C] is expected, got shape {tuple(t.shape)}")
    return Act(t, t.shape[3])


# ------------------------------------------------------------------------------------------------ convolution
@_op("conv2d_nhwc_fwd")
def conv2d_nhwc_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, pad: int, act: int,
                    slope: float) -> torch.Tensor:
    """y = act(conv2d(x, weight, bias)); x NHWC, weight OIHW fp32 (packed on the device per call)."""
    a = _act(x, "conv2d_nhwc_fwd(x)")
    return T.conv_forward_dev(weight.contiguous(), [(a, 0)], stride, pad, shift=bias, act=act, slope=slope).t


@conv2d_nhwc_fwd.register_fake
def _(x, weight, bias, stride, pad, act, slope):
    N, H, W, _ = x.shape
    Co, _, KH, KW = weight.shape
    return x.new_empty((N, (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1, ops._ceil4(Co)), dtype=torch.float32)


@_op("conv2d_nhwc_dgrad")
def conv2d_nhwc_dgrad(dy: torch.Tensor, weight: torch.Tensor, H: int, W: int, stride: int, pad: int) -> torch.Tensor:
    """dX [N, H, W, Cin] of y = conv2d(x, weight) (stride 1 or 2)."""
    return T.conv_dgrad(_act(dy, "conv2d_nhwc_dgrad(dy)"), weight.contiguous(), H, W, stride, pad).t


@conv2d_nhwc_dgrad.register_fake
def _(dy, weight, H, W, stride, pad):
    return dy.new_empty((dy.shape[0], H, W, ops._ceil4(weight.shape[1])), dtype=torch.float32)


@_op("conv2d_nhwc_wgrad")
def conv2d_nhwc_wgrad(dy: torch.Tensor, x: torch.Tensor, KH: int, KW: int, stride: int, pad: int) -> List[torch.Tensor]:
    """[dW (OIHW fp32), dbias] of y = conv2d(x, w) + b; the bias gradient rides along as a ones-column of the same reduction."""
    d, a = _act(dy, "conv2d_nhwc_wgrad(dy)"), _act(x, "conv2d_nhwc_wgrad(x)")
    dw = torch.empty((d.C, a.C, KH, KW), dtype=torch.float32, device=x.device)
    db = torch.empty(d.C, dtype=torch.float32, device=x.device)
    T.conv_wgrad(d, a, 0, 0, a.C, KH, KW, stride, pad, dw, dbias=db)
    return [dw, db]


@conv2d_nhwc_wgrad.register_fake
def _(dy, x, KH, KW, stride, pad):
    return [dy.new_empty((dy.shape[3], x.shape[3], KH, KW), dtype=torch.float32), dy.new_empty((dy.shape[3],), dtype=torch.float32)]


# ------------------------------------------------------------------------------------------------ normalisation
@_op("instnorm_stats")
def instnorm_stats(x: torch.Tensor, eps: float) -> List[torch.Tensor]:
    """(mean, rstd) [N, C] of InstanceNorm2d over an NHWC tensor (two-stage, shifted sums, double finalise)."""
    mean, rstd = ops.instnorm_stats(_act(x, "instnorm_stats(x)"), None, None, eps)
    return [mean, rstd]


@instnorm_stats.register_fake
def _(x, eps):
    return [x.new_empty((x.shape[0], x.shape[3]), dtype=torch.float32), x.new_empty((x.shape[0], x.shape[3]), dtype=torch.float32)]


@_op("spade_gamma_beta_fused_fwd")
def spade_gamma_beta_fused_fwd(actv: torch.Tensor, x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                               z: Optional[torch.Tensor], noise_scale: Optional[torch.Tensor], w_gamma: torch.Tensor,
                               w_beta: torch.Tensor, b_gamma: torch.Tensor, b_beta: torch.Tensor, act: int, slope: float,
                               save_g1p: bool) -> List[torch.Tensor]:
    """[out, 1 + gamma] (bf16): act(IN(x + z * noise_scale) * (1 + conv_gamma(actv)) + conv_beta(actv)) on the dedicated
    kernel (csrc/spade_gb.hip).  actv: bf16 NHWC [.., 128]; x: fp32 or bf16 NHWC [.., C]; C % 32 in {0, 16}."""
    a, xa = _act(actv, "spade_gamma_beta_fused_fwd(actv)"), _act(x, "spade_gamma_beta_fused_fwd(x)")
    if actv.dtype != torch.bfloat16:
        raise TypeError("spade_gamma_beta_fused_fwd: actv must be bf16 (matrix-core operand)")
    C_ = xa.C
    out = ops.alloc(xa.N, xa.H, xa.W, C_, x.device, bf16=True)
    g1p = torch.empty((xa.N, xa.H, xa.W, C_), dtype=torch.bfloat16, device=x.device) if save_g1p else None
    pk = T.spade_gb_pack(0, w_gamma.contiguous(), w_beta.contiguous())
    T.spade_gb_forward(a, xa, mean, rstd, z, noise_scale, pk, b_gamma, b_beta, act, slope, out, g1p, "spade_gamma_beta_fused_fwd",
                       2.0 * xa.N * xa.H * xa.W * 2 * C_ * a.C * 9, 0.0)
    return [out.t, g1p if save_g1p else x.new_empty(0, dtype=torch.bfloat16)]


@spade_gamma_beta_fused_fwd.register_fake
def _(actv, x, mean, rstd, z, noise_scale, w_gamma, w_beta, b_gamma, b_beta, act, slope, save_g1p):
    o = x.new_empty(x.shape, dtype=torch.bfloat16)
    return [o, x.new_empty(x.shape if save_g1p else (0,), dtype=torch.bfloat16)]


@_op("spade_gamma_beta_fused_bwd")
def spade_gamma_beta_fused_bwd(dgb: torch.Tensor, w_gamma: torch.Tensor, w_beta: torch.Tensor, actv: torch.Tensor) -> torch.Tensor:
    """d(actv) (bf16 [.., 128]) = conv^T([dgamma | dbeta]) * (actv > 0): data gradient of the fused pair with conv_shared's
    ReLU derivative."""
    d, m = _act(dgb, "spade_gamma_beta_fused_bwd(dgb)"), _act(actv, "spade_gamma_beta_fused_bwd(actv)")
    if dgb.dtype != torch.bfloat16 or actv.dtype != torch.bfloat16:
        raise TypeError("spade_gamma_beta_fused_bwd: dgb and actv must be bf16")
    out = ops.alloc(d.N, d.H, d.W, m.C, dgb.device, bf16=True)
    T.spade_gb_dgrad(d, T.spade_gb_pack(1, w_gamma.contiguous(), w_beta.contiguous()), w_gamma.shape[0], m, 0.0, out,
                     "spade_gamma_beta_fused_bwd")
    return out.t


@spade_gamma_beta_fused_bwd.register_fake
def _(dgb, w_gamma, w_beta, actv):
    return actv.new_empty(actv.shape)


@_op("spade_modulate_bwd")
def spade_modulate_bwd(x: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, dout: torch.Tensor, out: Optional[torch.Tensor],
                       g1p: Optional[torch.Tensor], z: Optional[torch.Tensor], noise_scale: Optional[torch.Tensor], act: int,
                       slope: float) -> List[torch.Tensor]:
    """[dx, [dgamma | dbeta] (bf16 when g1p is), d noise_scale] of out = act(IN(x + z ns) * g1p + beta)."""
    xa, da = _act(x, "spade_modulate_bwd(x)"), _act(dout, "spade_modulate_bwd(dout)")
    dns = torch.zeros(xa.Cp, dtype=torch.float32, device=x.device)
    dx, dgb = T.norm_bwd(xa, mean, rstd, da, act=act, slope=slope, out=None if out is None else _act(out, "spade_modulate_bwd(out)"),
                         g1p=None if g1p is None else _act(g1p, "spade_modulate_bwd(g1p)"), z=z, noise_scale=noise_scale,
                         want_dgb=g1p is not None, dnoise_scale=dns if z is not None else None,
                         dgb_bf16=g1p is not None and g1p.dtype == torch.bfloat16)
    return [dx.t, dgb.t if dgb is not None else x.new_empty(0), dns]


@spade_modulate_bwd.register_fake
def _(x, mean, rstd, dout, out, g1p, z, noise_scale, act, slope):
    N, H, W, C_ = x.shape
    dgb = x.new_empty((N, H, W, 2 * C_), dtype=g1p.dtype) if g1p is not None else x.new_empty(0)
    return [x.new_empty(x.shape, dtype=torch.float32), dgb, x.new_empty((C_,), dtype=torch.float32)]


# ------------------------------------------------------------------------------------------------ spectral norm / losses / Adam
@_op("spectral_sigma", mutates=("u", "v"))
def spectral_sigma(w_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor, power_iterations: int) -> torch.Tensor:
    """sigma [1] after ``power_iterations`` in-place power iterations on (u, v) (torch SpectralNorm.compute_weight)."""
    ops.require_cuda(w_orig, "spectral_sigma(w)")
    return T.spectral_sigma(w_orig.contiguous(), u, v, power_iterations)


@spectral_sigma.register_fake
def _(w_orig, u, v, power_iterations):
    return w_orig.new_empty((1,))


@_op("loss_reduce")
def loss_reduce(a: torch.Tensor, b: Optional[torch.Tensor], mode: int, lscale: float, gscale: float, with_grad: bool) -> List[torch.Tensor]:
    """[lscale * sum l(a, b), gscale * dl/da]; mode 0 L1, 1 hinge-D fake, 2 hinge-D real, 3 -a, 4 MSE (hrv_loss_f32)."""
    ops.require_cuda(a, "loss_reduce(a)")
    lo = torch.zeros(1, dtype=torch.float32, device=a.device)
    g = T.loss(a.contiguous(), None if b is None else b.contiguous(), mode, lscale, gscale, lo, accumulate=False, want_grad=with_grad)
    return [lo, g if g is not None else a.new_empty(0, dtype=torch.float32)]


@loss_reduce.register_fake
def _(a, b, mode, lscale, gscale, with_grad):
    return [a.new_empty((1,), dtype=torch.float32), a.new_empty(a.shape if with_grad else (0,), dtype=torch.float32)]


@_op("fused_adam", mutates=("w", "m", "v"))
def fused_adam(w: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, step: int, grad_scale: float) -> None:
    """One torch.optim.Adam update over flat fp32 buffers (w, m, v updated in place; g * grad_scale is the gradient)."""
    ops.require_cuda(w, "fused_adam(w)")
    T.adam_step(w, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale)


REGISTERED_OPS = ("conv2d_nhwc_fwd", "conv2d_nhwc_dgrad", "conv2d_nhwc_wgrad", "instnorm_stats", "spade_gamma_beta_fused_fwd",
                  "spade_gamma_beta_fused_bwd", "spade_modulate_bwd", "spectral_sigma", "loss_reduce", "fused_adam")
