"""Loss modules of the generator training step on the HIP path, with the reference's
call signatures: GANLoss('hinge') (network_generator.py:318-398), nn.L1Loss for the
feature-matching term (train_generator.py:148,300-309).  Each call is one fused
value+gradient kernel (hrv_loss_f32) wrapped in a torch.autograd.Function."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from . import train_ops as T


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mode):
        ops.require_cuda(a, "loss")
        # elementwise + full reduction: any dense layout works as long as both operands share it (the
        # discriminator hands out channels-last views of its NHWC feature maps -- no NCHW copy)
        cl = torch.channels_last
        if a.dim() == 4 and not a.is_contiguous() and a.is_contiguous(memory_format=cl) and \
                (b is None or (b.shape == a.shape and b.is_contiguous(memory_format=cl))):
            ac, bc = a, (None if b is None else b.detach())
        else:
            ac = a.contiguous()
            bc = None if b is None else b.detach().contiguous()
        n = ac.numel()
        out = torch.empty(1, dtype=torch.float32, device=a.device)      # (accumulate=False below: the kernel's final stage WRITES out[0])
        # (a bf16-stored operand -- the PatchGAN's feature taps in mixed precision -- gets its gradient in bf16 as well: autograd
        #  wants the input's dtype back, and the data gradient that adds it reads half the bytes)
        grad = T.loss(ac, bc, mode, 1.0 / n, 1.0 / n, out, accumulate=False, want_grad=ctx.needs_input_grad[0],
                      grad_bf16=ac.dtype == torch.bfloat16)
        ctx.grad = grad
        return out

    @staticmethod
    def backward(ctx, g_out):
        g = ctx.grad
        ctx.grad = None
        if g is None:
            return None, None, None
        # multiply by the upstream scalar on the device (no host sync)
        T.scale_(g, 1.0, g_out.contiguous())
        return g, None, None


class L1Loss(nn.Module):
    """nn.L1Loss() drop-in (mean reduction); the target receives no gradient."""

    def forward(self, input, target):
        return _LossFn.apply(input, target, T.LOSS_L1).squeeze(0)


class MSELoss(nn.Module):
    def forward(self, input, target):
        return _LossFn.apply(input, target, T.LOSS_MSE).squeeze(0)


class GANLoss(nn.Module):
    """network_generator.py:318-398.  'hinge' (the mode train_generator.py:146-148 uses) and 'w' run on
    the HIP kernels; 'ls'/'original' are not on the hot path."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor):
        super().__init__()
        if gan_mode not in ("ls", "original", "w", "hinge"):
            raise ValueError("Unexpected gan_mode {}".format(gan_mode))
        if gan_mode not in ("hinge", "w"):
            raise NotImplementedError("hr-viton_amd GANLoss implements gan_mode 'hinge' (and 'w')")
        self.gan_mode = gan_mode
        self.real_label, self.fake_label, self.Tensor = target_real_label, target_fake_label, tensor

    def loss(self, input, target_is_real, for_discriminator=True):
        if self.gan_mode == "hinge":
            if for_discriminator:
                mode = T.LOSS_HINGE_D_REAL if target_is_real else T.LOSS_HINGE_D_FAKE
                return _LossFn.apply(input, None, mode)
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            return _LossFn.apply(input, None, T.LOSS_NEG_MEAN)
        out = _LossFn.apply(input, None, T.LOSS_NEG_MEAN)     # wgan: -mean for real, +mean for fake
        return out if target_is_real else -out

    def __call__(self, input, target_is_real, for_discriminator=True):
        if isinstance(input, list):
            loss = 0
            for pred_i in input:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                loss = loss + self.loss(pred_i, target_is_real, for_discriminator)   # shape [1], like the reference
            return loss / len(input)
        return self.loss(input, target_is_real, for_discriminator)
