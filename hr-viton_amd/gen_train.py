"""Training-mode forward/backward of the SPADE generator and its PatchGAN on the HIP
path (train_generator.py:279-360).  Every module call is ONE ``torch.autograd.Function``
whose backward is a hand-written plan over the training kernels (train_ops.py):
torch.autograd only stitches the module-level graph of the reference's training
script (torch.cat of images, loss sums); no per-op autograd, no torch compute.

Forward saves, per SPADENorm: its input, (mean, rstd), the noise draw, the 128-ch
``actv``, (1+gamma) and its activated output; per conv: its sources.  Backward walks the
blocks in reverse: conv wgrad / dgrad (MFMA), SPADE + InstanceNorm backward, the 2x2
down-sum of the fused nearest-upsample store, spectral-norm gradient transform.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from . import train_ops as T
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, Act, _ceil4

Grads = Dict[nn.Parameter, torch.Tensor]


# ``p._hrv_grad_sync``: the GradSync a parameter reports to -- the backward plan announces every finished
# gradient so its bucket's all-reduce can start while the rest of the backward is still running (parallel.py)
def attach_grad_sync(sync):
    for p in sync.params:
        p._hrv_grad_sync = sync


def detach_grad_sync(params):
    for p in params:
        if hasattr(p, "_hrv_grad_sync"):
            del p._hrv_grad_sync


# ``p._hrv_flat_grad``: the parameter's slot in the fused optimizer's flat gradient buffer (optim.Adam sets the
# attribute at its first step).  Backward plans produce a gradient directly in that slot and hand it over as ``p.grad``: no autograd
# accumulation copy, no gather copy in the optimizer (~1000 tiny copy launches per iteration otherwise).
def flat_grad_slot(p: nn.Parameter) -> Optional[torch.Tensor]:
    return getattr(p, "_hrv_flat_grad", None)


def grad_buffer(p: nn.Parameter) -> torch.Tensor:
    """Where the gradient of ``p`` should be written: its flat-buffer slot when it is free this iteration."""
    v = flat_grad_slot(p)
    if v is not None and p.grad is None:
        return v
    return torch.empty_like(p.data)


def _pair_slot(pa: nn.Parameter, pb: nn.Parameter) -> Optional[torch.Tensor]:
    """One tensor [2 * rows, ...] over the flat-buffer slots of two same-shaped parameters when they are free this
    iteration and adjacent (optim._flat_order puts ``_hrv_flat_after`` mates next to each other), else None."""
    va, vb = flat_grad_slot(pa), flat_grad_slot(pb)
    if va is None or vb is None or pa.grad is not None or pb.grad is not None or va.shape != vb.shape:
        return None
    if vb.data_ptr() != va.data_ptr() + 4 * va.numel():
        return None
    shape = (2 * va.shape[0],) + tuple(va.shape[1:])
    stride, k = [], 1
    for d in reversed(shape):
        stride.append(k)
        k *= d
    return torch.as_strided(va, shape, tuple(reversed(stride)))


def _acc(grads: Grads, p: nn.Parameter, g: torch.Tensor):
    v = flat_grad_slot(p)
    if p.grad is not None:
        T.wgrad_join()      # a second contribution: the first may still be in flight on the weight-gradient side stream
    if v is not None:
        if p.grad is None:
            if g.data_ptr() != v.data_ptr():
                v.copy_(g.reshape(v.shape))
            p.grad = v
            g = v
        else:                       # a second contribution before the optimizer step (e.g. D(fake), D(real))
            p.grad.add_(g.reshape(p.grad.shape))
            g = p.grad
    else:
        assert p not in grads, "each parameter is used once per forward on this path"
        grads[p] = g
    s = getattr(p, "_hrv_grad_sync", None)
    if s is not None:
        s.on_grad(p, g)


class TConv:
    """A trainable (optionally spectral-normalised) convolution of the plan."""

    def __init__(self, conv: nn.Module, stride: int, pad: int, name: str):
        self.conv, self.stride, self.pad, self.name = conv, stride, pad, name
        self.spectral = hasattr(conv, "weight_orig")
        self.sigma: Optional[torch.Tensor] = None

    @property
    def wparam(self) -> nn.Parameter:
        return self.conv.weight_orig if self.spectral else self.conv.weight

    @property
    def bparam(self) -> Optional[nn.Parameter]:
        return getattr(self.conv, "bias", None)

    def prepare(self, power_iteration: bool):
        """torch SpectralNorm.compute_weight: one in-place power iteration on (u, v) when the
        module is in training mode, then sigma (kept on the device)."""
        if self.spectral:
            self.sigma = T.spectral_sigma(self.wparam.data, self.conv.weight_u, self.conv.weight_v,
                                          1 if power_iteration else 0)
            # keep the (u, v) that produced sigma: the backward treats them as constants
            self.u, self.v = self.conv.weight_u.clone(), self.conv.weight_v.clone()

    def forward(self, srcs: Sequence[Tuple[Act, int]], act: int = ACT_NONE, residual: Optional[Act] = None,
                out: Optional[Act] = None, out_up: int = 0, slope: float = 0.2, out_bf16: bool = False) -> Act:
        b = self.bparam
        return T.conv_forward_dev(self.wparam.data, srcs, self.stride, self.pad, sigma=self.sigma,
                                  shift=None if b is None else b.data, residual=residual, act=act, slope=slope, out=out,
                                  out_up=out_up, name=self.name, out_bf16=out_bf16, batch=getattr(self, "pack_batch", None))

    def backward(self, dy: Act, srcs: Sequence[Tuple[Act, int]], grads: Grads, need_dx: bool = True,
                 act_mask: Optional[Act] = None, slope: float = 0.2, need_w: bool = True, dx_bf16: bool = False,
                 dy_wgrad: Optional[Act] = None, add: Optional[Act] = None, dx_out: Optional[Act] = None) -> Optional[Act]:
        """``dy_wgrad``: a bf16 copy of ``dy`` for the weight gradient (so a bf16-stored source takes the LDS-DMA kernel).
        ``dx_out``: where the data gradient goes (a channel slice of a wider tensor: the dbeta half of a SPADE norm's
        [dgamma | dbeta], SpadeT.dout_slot)."""
        w = self.wparam.data
        Cout, cin, KH, KW = w.shape
        if need_w:
            dyw = dy if dy_wgrad is None else dy_wgrad
            with T.wgrad_side(dyw.N * dyw.H * dyw.W, dyw, *[a for a, _ in srcs]):      # (a leaf of the backward: second stream)
                G = torch.empty_like(w) if self.spectral else grad_buffer(self.wparam)
                db = grad_buffer(self.bparam) if self.bparam is not None else None
                base = 0
                for a, up in srcs:
                    # the bias gradient rides along with the first source as a ones-column of the same MFMA reduction
                    T.conv_wgrad(dyw, a, up, base, cin, KH, KW, self.stride, self.pad, G,
                                 name=self.name + ".wgrad", dbias=db if base == 0 else None)
                    base += a.C
                if self.spectral:
                    dwo = grad_buffer(self.wparam)
                    T.spectral_grad(G, w, self.u, self.v, self.sigma, dwo)
                    _acc(grads, self.wparam, dwo)
                else:
                    _acc(grads, self.wparam, G)
                if db is not None:
                    _acc(grads, self.bparam, db)
        if not need_dx:
            return None
        a0, up0 = srcs[0]
        H, W = (a0.H << up0, a0.W << up0) if up0 >= 0 else (a0.H >> -up0, a0.W >> -up0)
        return T.conv_dgrad(dy, w, H, W, self.stride, self.pad, sigma=self.sigma, act_mask=act_mask, slope=slope, out=dx_out,
                            name=self.name + ".dgrad", out_bf16=dx_bf16, add=add, batch=getattr(self, "pack_batch", None))


class S2DConv(TConv):
    """A 4x4, stride-2, pad-2 convolution over a handful of channels -- PatchGAN's model0 (network_generator.py
    NLayerDiscriminator: input_nc = 7 + 3 -> ndf) -- computed as a 2x2 stride-1 pad-1 convolution over the space-to-depth
    tensor [N, H/2, W/2, 4*Cp]:  kh = 2*ty + dy, kw = 2*tx + dx, in[2(o-1+ty)+dy] = cell (o-1+ty), sub-pixel dy.
    The implicit-GEMM engine gives every TAP its own 64-k row group: 16 taps x 10 of 64 channels is 5/6 padding (model0
    forward 0.64 ms, its weight gradient 0.64 ms, its four-phase data gradient 0.67 ms at 1024x768, all at ~1 TB/s);
    4 taps x 48 channels fill it.  Same products, same fp32 accumulation -- only the summation order inside a K-tile
    changes.  The weight lives in a buffer that stays (the plan's PackBatch reads it by address); ``refresh`` re-derives
    it from the parameter before the plan's ``prepare_convs``."""

    @staticmethod
    def fits(m: nn.Module) -> bool:
        return (tuple(m.kernel_size) == (4, 4) and tuple(m.stride) == (2, 2) and tuple(m.padding) == (2, 2) and
                m.in_channels <= 16 and m.groups == 1 and tuple(m.dilation) == (1, 1) and
                os.environ.get("HRV_S2D_CONV", "1") != "0")

    def __init__(self, conv: nn.Module, name: str):
        super().__init__(conv, 2, 2, name)
        self.Cq = _ceil4(conv.in_channels)
        self._w2: Optional[torch.Tensor] = None
        self._cache = None

    def refresh(self):
        w = self.wparam.data
        Cout, cin = w.shape[0], w.shape[1]
        if self._w2 is None or self._w2.device != w.device:
            self._w2 = torch.zeros((Cout, 2, 2, self.Cq, 2, 2), dtype=torch.float32, device=w.device)
        # (co, c, ty, dy, tx, dx) -> (co, dy, dx, c, ty, tx); the pad channels stay zero
        self._w2[:, :, :, :cin].copy_(w.view(Cout, cin, 2, 2, 2, 2).permute(0, 3, 5, 1, 2, 4))

    @property
    def w2(self) -> torch.Tensor:
        return self._w2.view(self._w2.shape[0], 4 * self.Cq, 2, 2)

    def forward(self, srcs, act: int = ACT_NONE, residual=None, out=None, out_up: int = 0, slope: float = 0.2,
                out_bf16: bool = False) -> Act:
        a, up = srcs[0]
        if (len(srcs) != 1 or up != 0 or a.bf16 or a.H % 2 or a.W % 2 or a.Cp != self.Cq or residual is not None or
                out is not None or out_up != 0 or self._w2 is None):
            self._cache = None
            return super().forward(srcs, act, residual, out, out_up, slope, out_bf16)
        a2 = T.space_to_depth2(a)
        self._cache = (a.t.data_ptr(), a.coff, a2)
        b = self.bparam
        return T.conv_forward_dev(self.w2, [(a2, 0)], 1, 1, sigma=self.sigma, shift=None if b is None else b.data, act=act,
                                  slope=slope, name=self.name, out_bf16=out_bf16, batch=getattr(self, "pack_batch", None))

    def backward(self, dy: Act, srcs, grads: Grads, need_dx: bool = True, act_mask=None, slope: float = 0.2,
                 need_w: bool = True, dx_bf16: bool = False, dy_wgrad=None, add=None):
        a, _ = srcs[0]
        c = self._cache
        if c is None or c[0] != a.t.data_ptr() or c[1] != a.coff or act_mask is not None or add is not None:
            return super().backward(dy, srcs, grads, need_dx, act_mask, slope, need_w, dx_bf16, dy_wgrad, add)
        a2 = Act(c[2].t[:a.N], c[2].C)             # (the generator step back-propagates the fake half of the batch only)
        w = self.wparam.data
        Cout, cin = w.shape[0], w.shape[1]
        if need_w:
            dw2 = torch.empty_like(self._w2)
            db = grad_buffer(self.bparam) if self.bparam is not None else None
            T.conv_wgrad(dy, a2, 0, 0, 4 * self.Cq, 2, 2, 1, 1, dw2.view(Cout, 4 * self.Cq, 2, 2), name=self.name + ".wgrad",
                         dbias=db)
            G = dw2[:, :, :, :cin].permute(0, 3, 4, 1, 5, 2).reshape(Cout, cin, 4, 4)      # back to (co, c, kh, kw)
            if self.spectral:
                dwo = grad_buffer(self.wparam)
                T.spectral_grad(G, w, self.u, self.v, self.sigma, dwo)
                _acc(grads, self.wparam, dwo)
            else:
                _acc(grads, self.wparam, G)
            if db is not None:
                _acc(grads, self.bparam, db)
        if not need_dx:
            return None
        d2 = T.conv_dgrad(dy, self.w2, a2.H, a2.W, 1, 1, sigma=self.sigma, name=self.name + ".dgrad",
                          batch=getattr(self, "pack_batch", None))
        return T.depth_to_space2(d2, a.C)


class SpadeT:
    """One SPADENorm (+ optional LeakyReLU) in training mode."""

    def __init__(self, norm: nn.Module, act: int, name: str):
        self.norm, self.act, self.name = norm, act, name
        self.shared = TConv(norm.conv_shared[0], 1, 1, name + ".conv_shared")
        self.C = norm.conv_gamma.out_channels
        self.Cp = _ceil4(self.C)
        self.hid = norm.conv_gamma.in_channels
        G = (self.C + 31) // 32
        self.G = G
        self.cfg = 0 if G % 2 == 0 else 6
        dev = norm.conv_gamma.weight.device
        rows_g = torch.tensor([g * 64 + l for g in range(G) for l in range(32) if g * 32 + l < self.C], device=dev)
        self.rows_g, self.rows_b = rows_g, rows_g + 32

    def vecs(self):
        """(bias of the fused gamma|beta conv in its interleaved column order, noise scale padded to ceil4(C)): from the plan's one
        launch for all its norms (GeneratorTrainPlan.forward) when that ran for this forward, else one launch of its own."""
        v = getattr(self, "_vec_pre", None)
        if v is not None:
            return v
        n = self.norm
        return T.spade_vec_prep(n.conv_gamma.bias.data, n.conv_beta.bias.data, n.noise_scale.data)

    def shared_as_1x1(self) -> Tuple[torch.Tensor, torch.Tensor, int]:
        """conv_shared's 3x3 weight as a 1x1 over the tap-expanded label map (ops.tap_expand: tap-major, channel-
        minor, channels padded to one 16-byte group): ([hid, 9*cp, 1, 1], bias, cp)."""
        w = self.shared.wparam.data                       # [hid, label_nc, 3, 3]
        co, c, kh, kw = w.shape
        return w.permute(0, 2, 3, 1), self.shared.bparam.data, c

    def forward(self, x: Act, actv: Optional[Act], z: Optional[torch.Tensor], save: bool = True, fused=None, stats=None):
        """``fused`` = (bf16 label map Act [N, H << shift, W << shift, 8], shift): conv_shared + ReLU are computed inside the
        gamma|beta kernel (csrc/spade_fused.hip); ``actv`` is then the slice the kernel WRITES for the backward (None: no_grad
        forward, actv never reaches HBM).  ``stats`` = (padded noise scale, (mean, rstd)): the statistics were computed together
        with another norm's over the same x (BlockT.forward, ops.instnorm_stats2)."""
        n = self.norm
        dev = x.t.device
        if fused is not None:
            sg, shift = fused
            zz = z
            if stats is not None:
                ns, (mean, rstd) = stats
            else:
                ns = self.vecs()[1] if zz is not None else None
                mean, rstd = ops.instnorm_stats(x, zz, ns if zz is not None else None)
            out = ops.alloc(x.N, x.H, x.W, self.C, dev, bf16=True)
            g1p = torch.empty((x.N, x.H, x.W, self.Cp), dtype=torch.bfloat16, device=dev) if save else None
            pk = T.spade_fused_pack(self.shared.wparam.data, self.shared.bparam.data, n.conv_gamma.weight.data, n.conv_beta.weight.data)
            T.spade_fused_forward(sg, shift, x, mean, rstd, zz, n.noise_scale.data, pk, n.conv_gamma.bias.data, n.conv_beta.bias.data,
                                  self.act, 0.2, out, g1p, actv if save else None, self.name + ".conv_shared+gamma|beta")
            ctx = dict(x=x, z=zz, ns=ns, mean=mean, rstd=rstd, actv=actv, g1p=Act(g1p, self.C) if save else None, out=out)
            return out, ctx
        # bias of the fused conv in its interleaved (gamma32 | beta32) column order + padded noise scale: one launch
        bc, ns = self.vecs()
        zz = z  # the noise term is always applied in training (noise_scale is a learnable parameter)
        if stats is not None:
            mean, rstd = stats[1]
        else:
            mean, rstd = ops.instnorm_stats(x, zz, ns if zz is not None else None)
        mb = T.MMA_BF16[0]     # mixed precision: bf16 matrix cores, fp32 epilogue / statistics / x
        cfg = ((8 if self.G % 2 == 0 else 9) if mb else self.cfg)
        if mb:                 # bf16 actv: each tile's halo patch stays in LDS (ops.patch_tile)
            cfg = ops.patch_tile(actv.bf16, 3, 3, 1, 1, 1, 0, self.hid, self.G * 64, x.N, x.H, x.W, wide=True) or cfg
        out_bf16 = mb and actv.bf16 and self.C % 8 == 0
        if (mb and actv.bf16 and out_bf16 and x.cstride % 4 == 0 and
                T.spade_gb_ok(0, self.C, self.Cp, self.hid, x.N, x.H, x.W)):
            # the dedicated kernel: 16x16-pixel tiles x ALL columns per block, no padded columns (csrc/spade_gb.hip)
            out = ops.alloc(x.N, x.H, x.W, self.C, dev, bf16=True)
            # (1 + gamma) in bf16: the normalisation backward is its only reader (half the bytes of the fp32 form)
            g1p = torch.empty((x.N, x.H, x.W, self.Cp), dtype=torch.bfloat16, device=dev) if save else None
            pk = T.spade_gb_pack(0, n.conv_gamma.weight.data, n.conv_beta.weight.data)
            fl = 2.0 * x.N * x.H * x.W * 2 * self.C * self.hid * 9
            nbytes = (ops.act_bytes(actv) + (1.5 if save else 1) * ops.act_bytes(x) + ops.act_bytes(out) +
                      2.0 * self.C * self.hid * 9 * 2)
            T.spade_gb_forward(actv, x, mean, rstd, zz, n.noise_scale.data, pk, n.conv_gamma.bias.data, n.conv_beta.bias.data,
                               self.act, 0.2, out, g1p, self.name + ".conv_gamma|beta", fl, nbytes)
            ctx = dict(x=x, z=zz, ns=ns, mean=mean, rstd=rstd, actv=actv, g1p=Act(g1p, self.C) if save else None, out=out)
            return out, ctx
        # (conv_gamma, conv_beta) weights packed straight into the combined interleaved matrix (no concatenated copy)
        packed, _, _ = T.pack_weight_pair_dev(n.conv_gamma.weight.data, n.conv_beta.weight.data, 1, [self.hid], [self.hid],
                                              cfg, 0, 1, mb, batch=getattr(self.shared, "pack_batch", None))
        # mixed precision: the modulated activation is read by matrix cores only (conv_0 / conv_1 / conv_s, their weight
        # gradients) and as the sign mask of its own LeakyReLU -- stored in bf16 (same operand bits as rounding while
        # staging, half the bytes); widths that are not a multiple of 4 keep the fp32 weight-gradient kernel and fp32
        out = ops.alloc(x.N, x.H, x.W, self.C, dev, bf16=mb and actv.bf16 and self.C % 8 == 0)
        # (1 + gamma) is only read by the backward: the no_grad forward of the discriminator step (train_generator.py:
        # 327-330) does not write it (252 MB per image and norm at up_4)
        # (``save``: passed down by the caller -- torch.is_grad_enabled() is always False inside autograd.Function.forward)
        g1p = torch.empty((x.N, x.H, x.W, self.Cp), dtype=torch.float32, device=dev) if save else None
        e = ops._lib.hrv_spade_epi_t()
        e.x, e.x_cstride, e.x_coff, e.C = x.t.data_ptr(), x.cstride, x.coff, self.Cp
        e.mean, e.rstd = mean.data_ptr(), rstd.data_ptr()
        if zz is not None:
            e.noise_z, e.noise_scale = zz.data_ptr(), ns.data_ptr()
        if save:
            e.g1p_out = g1p.data_ptr()
        import ctypes as C
        lib = ops._lib.load()
        d = ops._lib.hrv_conv2d_t()
        d.N, d.H, d.W, d.Ho, d.Wo = x.N, x.H, x.W, x.H, x.W
        d.KH, d.KW, d.stride, d.pad = 3, 3, 1, 1
        d.nsrc = 1
        s = d.src[0]
        s.ptr, s.C, s.cstride, s.coff, s.up_shift, s.pre_act, s.C_real = (actv.t.data_ptr(), self.hid, actv.cstride,
                                                                          actv.coff, 0, 0, self.hid)
        d.w_packed, d.Cout, d.tile_cfg = packed.data_ptr(), self.G * 64, cfg
        d.mixed_flags = ((7 if actv.bf16 else 15) & ~(1 if out.bf16 else 0)) if mb else 0
        d.shift = bc.data_ptr()
        d.act, d.act_slope = self.act, 0.2
        d.out, d.out_cstride, d.out_coff = out.t.data_ptr(), out.cstride, out.coff
        d.spade = C.pointer(e)
        fl = 2.0 * x.N * x.H * x.W * 2 * self.C * self.hid * 9
        nbytes = (ops.act_bytes(actv) + (2 if save else 1) * ops.act_bytes(x) + ops.act_bytes(out) +
                  2.0 * self.C * self.hid * 9 * (2 if mb else 4))     # actv, x, out, 1+gamma, weights
        with ops._Timed("conv", self.name + ".conv_gamma|beta", fl, nbytes, f"conv_mfma_kernel[tile {cfg}]"):
            fn = lib.hrv_conv2d_nhwc_bf16 if mb else lib.hrv_conv2d_nhwc_f32
            ops._lib.check(fn(C.byref(d), ops._stream()), "hrv_conv2d_nhwc_%s[spade]" % ("bf16" if mb else "f32"))
        ctx = dict(x=x, z=zz, ns=ns, mean=mean, rstd=rstd, actv=actv, g1p=Act(g1p, self.C) if save else None, out=out)
        return out, ctx

    def norm_args(self, ctx, dout: Act):
        """keyword arguments of T.norm_bwd for this norm (without x / dx) + the noise-scale gradient buffer"""
        n = self.norm
        dns = None
        if ctx["z"] is not None:
            dns = grad_buffer(n.noise_scale) if self.Cp == self.C else torch.empty(self.Cp, device=dout.t.device)
        return dict(mean=ctx["mean"], rstd=ctx["rstd"], dout=dout, act=self.act, slope=0.2,
                    out=ctx["out"] if self.act != ACT_NONE else None, g1p=ctx["g1p"], z=ctx["z"],
                    noise_scale=ctx["ns"] if ctx["z"] is not None else None, want_dgb=True, dnoise_scale=dns,
                    dgb_bf16=ctx["actv"].bf16)   # [dgamma|dbeta] feeds matrix cores only (gb.wgrad, gb.dgrad)

    def dout_slot(self, ctx, want_bf16: bool):
        """(dgb, its dbeta half, the activation mask) when the gradient of this norm's output can be WRITTEN into the dbeta half of
        [dgamma | dbeta] by the data gradient that produces it, LeakyReLU derivative applied in that kernel's epilogue (mixed
        precision, dense channels): the normalisation backward then reads neither the activation output nor stores dbeta again --
        4 of its ~28 bytes per element.  Else None (HRV_DBETA_INPLACE=0: always None)."""
        if (os.environ.get("HRV_DBETA_INPLACE", "1") == "0" or os.environ.get("HRV_NORM_BWD2", "0") != "0" or
                not T.MMA_BF16[0] or not want_bf16 or self.Cp != self.C or
                self.C % 8 != 0 or not ctx["actv"].bf16 or not ctx["out"].bf16):
            return None
        dgb, dbeta = T.norm_bwd_dgb(ctx["x"], True)
        return dgb, dbeta, (ctx["out"] if self.act != ACT_NONE else None)

    def backward(self, ctx, dout: Act, grads: Grads, dx: Optional[Act], dx_accumulate: bool, dact: Act,
                 dx_bf16: bool = False, dgb: Optional[Act] = None) -> Act:
        """``dact``: this norm's slice of the block-wide d(actv) tensor (the block back-propagates its norms'
        conv_shared together, BlockT.backward).  ``dgb``: ``dout`` is its dbeta half (dout_slot), activation derivative applied."""
        a = self.norm_args(ctx, dout)
        if dgb is not None:
            a.update(act=ACT_NONE, out=None, dgb=dgb)
        dx, dgb = T.norm_bwd(ctx["x"], dx=dx, dx_accumulate=dx_accumulate, dx_bf16=dx_bf16 and ctx["actv"].bf16, **a)
        self.after_norm(ctx, dgb, a["dnoise_scale"], grads, dact)
        return dx

    def after_norm(self, ctx, dgb: Act, dns: Optional[torch.Tensor], grads: Grads, dact: Act):
        """what follows the normalisation backward: the noise-scale gradient, the gamma|beta weight and data gradients"""
        n = self.norm
        C_, Cp = self.C, self.Cp
        dev = dgb.t.device
        if dns is not None:
            _acc(grads, n.noise_scale, dns[:C_])
        else:
            _acc(grads, n.noise_scale, torch.zeros(C_, device=dev))
        # gamma/beta convs: one conv with Wcat = [Wgamma ; Wbeta] over dgb = [dgamma | dbeta]
        actv = ctx["actv"]
        if Cp == C_:           # dense halves: the data gradient packs (W_gamma, W_beta) as a pair, no concatenated copy
            wcat = (n.conv_gamma.weight.data, n.conv_beta.weight.data)
        else:
            wcat = torch.zeros((2 * Cp, self.hid, 3, 3), device=dev)
            wcat[:C_] = n.conv_gamma.weight.data
            wcat[Cp:Cp + C_] = n.conv_beta.weight.data
        with T.wgrad_side(dgb.N * dgb.H * dgb.W, dgb, actv):      # (a leaf of the backward: second stream at the coarse levels)
            dwcat, db = _pair_slot(n.conv_gamma.weight, n.conv_beta.weight), _pair_slot(n.conv_gamma.bias, n.conv_beta.bias)
            if Cp != C_ or dwcat is None or db is None:
                dwcat = torch.empty((2 * Cp, self.hid, 3, 3), device=dev)
                db = torch.empty(2 * Cp, device=dev)
            T.conv_wgrad(dgb, actv, 0, 0, self.hid, 3, 3, 1, 1, dwcat, name=self.name + ".gb.wgrad", dbias=db)
            direct = flat_grad_slot(n.conv_gamma.weight) is not None
            keep = (lambda t: t) if direct else (lambda t: t.clone())     # slices are copied into the flat slots by _acc
            _acc(grads, n.conv_gamma.weight, keep(dwcat[:C_]))
            _acc(grads, n.conv_beta.weight, keep(dwcat[Cp:Cp + C_]))
            _acc(grads, n.conv_gamma.bias, keep(db[:C_]))
            _acc(grads, n.conv_beta.bias, keep(db[Cp:Cp + C_]))
        # d actv, with the ReLU derivative of conv_shared fused (slope 0)
        if (Cp == C_ and T.MMA_BF16[0] and dgb.bf16 and actv.bf16 and dact.cstride % 8 == 0 and (2 * C_) % 32 == 0 and
                T.conv_p2_ok(2 * C_, self.hid, actv.N, actv.H, actv.W)):
            # two blocks per CU, 32-channel source chunks double-buffered (csrc/conv_p2.hip)
            pk = T.conv_p2_pack(2, n.conv_gamma.weight.data, n.conv_beta.weight.data, 2 * C_, self.hid)
            T.conv_p2(dgb, pk, self.hid, dact, mask=actv, mask_slope=0.0, name=self.name + ".gb.dgrad",
                      flops=2.0 * actv.N * actv.H * actv.W * 2 * C_ * self.hid * 9, tag=" [spade_gb]")
        elif (Cp == C_ and T.MMA_BF16[0] and dgb.bf16 and actv.bf16 and
                T.spade_gb_ok(1, C_, Cp, self.hid, actv.N, actv.H, actv.W)):
            T.spade_gb_dgrad(dgb, T.spade_gb_pack(1, n.conv_gamma.weight.data, n.conv_beta.weight.data), C_, actv, 0.0, dact,
                             self.name + ".gb.dgrad")
        else:
            # (a padded ``wcat`` is a temporary: a PackBatch record holds the weight's raw address, so only the persistent
            #  parameter pair may be recorded -- train_ops._pack_batched)
            T.conv_dgrad(dgb, wcat, actv.H, actv.W, 1, 1, act_mask=actv, slope=0.0, out=dact, name=self.name + ".gb.dgrad",
                         batch=getattr(self.shared, "pack_batch", None) if Cp == C_ else None)


class BlockT:
    def __init__(self, blk: nn.Module, name: str):
        self.blk, self.name = blk, name
        self.learned = blk.learned_shortcut
        self.n0 = SpadeT(blk.norm_0, ACT_LRELU, name + ".norm_0")
        self.n1 = SpadeT(blk.norm_1, ACT_LRELU, name + ".norm_1")
        self.c0 = TConv(blk.conv_0, 1, 1, name + ".conv_0")
        self.c1 = TConv(blk.conv_1, 1, 1, name + ".conv_1")
        if self.learned:
            self.ns_ = SpadeT(blk.norm_s, ACT_NONE, name + ".norm_s")
            self.cs = TConv(blk.conv_s, 1, 0, name + ".conv_s")

    def convs(self):
        return [self.c0, self.c1] + ([self.cs] if self.learned else [])

    def wants_bf16_dout(self, ctx) -> bool:
        """d(block output) may be stored in bf16: learned shortcut (no fp32 residual add of the gradient) and this level's
        conv inputs are bf16 (so the weight-gradient kernel takes bf16 on both sides)."""
        return bool(self.learned and ctx["h1"].bf16 and ctx["hs"].bf16 and self.c1.conv.out_channels % 8 == 0)

    def norms(self):
        return ([self.ns_] if self.learned else []) + [self.n0, self.n1]

    def _fused_ok(self, N: int, H: int, W: int, x_cstride: int, seg: Act, seg_shift: int) -> bool:
        if not (T.MMA_BF16[0] and seg.coff == 0 and seg.cstride == 8 and (seg.W >> seg_shift) % 4 == 0 and x_cstride % 4 == 0):
            return False
        return all(n_.C % 8 == 0 and n_.Cp == n_.C and n_.norm.conv_shared[0].weight.shape[1] <= 8 and
                   T.spade_fused_ok(n_.C, n_.hid, n_.norm.conv_shared[0].weight.shape[1], N, H, W) for n_ in self.norms())

    def fused_label_map(self, x: Act, seg: Act, seg_shift: int) -> Optional[Act]:
        """The bf16 label map when EVERY norm of this block can run conv_shared inside its gamma|beta kernel (mixed precision,
        csrc/spade_fused.hip: two blocks per CU, actv never read from HBM), else None."""
        if not self._fused_ok(x.N, x.H, x.W, x.cstride, seg, seg_shift):
            return None
        sg = seg if seg.bf16 else getattr(seg, "_as_bf16", None)          # one cast per step, shared by the blocks
        if sg is None:
            sg = seg._as_bf16 = Act(seg.t.to(torch.bfloat16), seg.C)
        return sg

    def reads_upsampled_input(self, N: int, H: int, W: int, C_in: int, seg: Act, seg_shift: int) -> bool:
        """This block can read its input as ops.ActUp (never materialised): every reader of x must understand it -- the one-pass
        statistics of norm_s / norm_0, the fused SPADE forward, the normalisation backward -- i.e. a learned-shortcut block whose
        norms all run on csrc/spade_fused.hip (mixed precision).  HRV_XUP=0 switches it off (A/B runs)."""
        if os.environ.get("HRV_XUP", "1") == "0" or os.environ.get("HRV_STATS2", "1") == "0" or not self.learned:
            return False
        if H % 2 or W % 2 or (C_in - 16) % 32 != 0 or self.n0.C != C_in or self.ns_.C != C_in:
            return False
        return self._fused_ok(N, H, W, 4, seg, seg_shift)

    def shared_forward(self, seg: Act, seg_shift: int):
        """The conv_shared 3x3s (label_nc -> 128, + ReLU) of the block's norms as ONE 1x1 convolution over the
        tap-expanded label map (ops.tap_expand: 9 taps x 8 padded channels = 72 dense inputs): the label map is
        read once per block and no K-tile is 7/8 padding (the 3x3 form spends 9 K-tiles on 8 channels each).
        Mixed precision: the one-hot map is exact in bf16, and actv is read by matrix cores only (the gamma|beta
        conv, its weight gradient, the sign mask of its data gradient) -- both are stored in bf16 there; levels
        whose width is not a multiple of 4 take the fp32 weight-gradient kernel and stay fp32."""
        norms = self.norms()
        mb = T.MMA_BF16[0] and (seg.W >> seg_shift) % 4 == 0
        sg = seg
        if mb and seg.coff == 0 and seg.cstride == 8:
            sg = getattr(seg, "_as_bf16", None)          # one cast per step, shared by the blocks
            if sg is None:
                sg = seg._as_bf16 = Act(seg.t.to(torch.bfloat16), seg.C)
        mb = mb and sg.bf16
        segx = ops.tap_expand(sg, seg_shift, 3)
        cp = sg.Cp
        # the norms' conv_shared weights as one tap-major 1x1 weight + concatenated bias: one launch
        wt_all, b_all = T.shared_taps_prep([n_.shared.wparam.data for n_ in norms], [n_.shared.bparam.data for n_ in norms], cp)
        hid = norms[0].hid
        actv_all = T.conv_forward_dev(wt_all, [(segx, 0)], 1, 0, shift=b_all, act=ACT_RELU,
                                      out_bf16=mb, name=self.name + ".conv_shared[x%d as 1x1 over taps]" % len(norms))
        return segx, [actv_all.slice(hid * i, hid) for i in range(len(norms))]

    def shared_backward(self, segx: Act, dact_all: Act, grads: Grads):
        norms = self.norms()
        hid, cp = norms[0].hid, segx.C // 9
        with T.wgrad_side(dact_all.N * dact_all.H * dact_all.W, dact_all, segx):
            dw = torch.empty((hid * len(norms), segx.C, 1, 1), device=dact_all.t.device)
            db = torch.empty(hid * len(norms), device=dact_all.t.device)
            T.conv_wgrad(dact_all, segx, 0, 0, segx.C, 1, 1, 1, 0, dw, name=self.name + ".conv_shared.wgrad", dbias=db)
            # tap-major 1x1 gradient -> the norms' [hid, c, 3, 3] weight / bias gradients (their flat-buffer slots when free)
            gws = [grad_buffer(n_.shared.wparam) for n_ in norms]
            gbs = [grad_buffer(n_.shared.bparam) for n_ in norms]
            T.shared_taps_grad(dw, db, gws, gbs, cp)
            for n_, gw, gb in zip(norms, gws, gbs):
                _acc(grads, n_.shared.wparam, gw)
                _acc(grads, n_.shared.bparam, gb)

    def forward(self, x: Act, seg: Act, seg_shift: int, zs, out: Optional[Act], out_up: int, out_act: int,
                save: bool = True):
        zi = iter(zs)
        ctx = {"x": x}
        sgf = self.fused_label_map(x, seg, seg_shift)
        fused = None
        if sgf is not None:
            # conv_shared runs inside each norm's gamma|beta kernel.  The training forward keeps what the backward reads: the
            # norms' actv side by side (written by those kernels, bf16) and the tap-expanded label map (conv_shared's weight
            # gradient); the no_grad forward keeps neither
            fused = (sgf, seg_shift)
            norms = self.norms()
            hid = norms[0].hid
            segx, actvs = None, [None] * len(norms)
            if save:
                segx = ops.tap_expand(sgf, seg_shift, 3)
                actv_all = Act(torch.empty((x.N, x.H, x.W, hid * len(norms)), dtype=torch.bfloat16, device=x.t.device), hid * len(norms))
                actvs = [actv_all.slice(hid * i, hid) for i in range(len(norms))]
        else:
            segx, actvs = self.shared_forward(seg, seg_shift)
        ai = iter(actvs)
        ctx["segx"] = segx
        st_s = st_0 = None
        if self.learned:
            z_s, z_0 = next(zi), next(zi)
            if z_s is not None and z_0 is not None and not x.bf16 and os.environ.get("HRV_STATS2", "1") != "0":
                # norm_s and norm_0 normalise the same x (network_generator.py:158-166) with their own noise draws: one pass over it
                ns_s = self.ns_.vecs()[1]
                ns_0 = self.n0.vecs()[1]
                ms, m0 = ops.instnorm_stats2(x, z_s, ns_s, z_0, ns_0)
                st_s, st_0 = (ns_s, ms), (ns_0, m0)
            hs, ctx["ns"] = self.ns_.forward(x, next(ai), z_s, save, fused, st_s)
            # (round 6, measured and not kept: this shortcut output and conv_0's output stored in bf16 -- DESIGN.md 7f)
            x_s = self.cs.forward([(hs, 0)])
            ctx["hs"] = hs
        else:
            x_s = x
            z_0 = next(zi)
        h0, ctx["n0"] = self.n0.forward(x, next(ai), z_0, save, fused, st_0)
        dx = self.c0.forward([(h0, 0)])
        h1, ctx["n1"] = self.n1.forward(dx, next(ai), next(zi), save, fused)
        # the last block's activated output only feeds conv_img (matrix cores + the sign mask of its data gradient)
        o = self.c1.forward([(h1, 0)], residual=x_s, act=out_act, out=out, out_up=out_up,
                            out_bf16=out is None and h1.bf16 and self.c1.conv.out_channels % 8 == 0)
        ctx.update(h0=h0, h1=h1)
        return o, ctx

    def backward(self, ctx, d_out: Act, grads: Grads) -> Act:
        """d_out: gradient w.r.t. the block's PRE-activation output (x_s + conv_1(...)) at block resolution."""
        x = ctx["x"]
        norms = self.norms()
        hid = norms[0].hid
        # d(actv) of the block's norms, side by side: read by conv_shared's weight gradient only (matrix cores) -- bf16 in
        # mixed precision when the tap-expanded label map is (the LDS-DMA weight-gradient kernel then takes both operands)
        dact_all = ops.alloc(x.N, x.H, x.W, hid * len(norms), x.t.device,
                             bf16=bool(T.MMA_BF16[0] and ctx["segx"].bf16 and x.W % 4 == 0))
        k0 = 1 if self.learned else 0          # slice order = norms(): [norm_s,] norm_0, norm_1
        # d(h): the gradient of a bf16-STORED SPADE output, read once by that norm's backward -- stored in bf16 as well
        # (autocast hands the gradient of a half-precision convolution input back in half precision; HRV_DH_BF16=0: fp32)
        dh16 = bool(T.MMA_BF16[0] and os.environ.get("HRV_DH_BF16", "1") != "0")
        def down(conv, d_y, h, norm, nctx):
            """d(h) of h = act(norm(.)) through ``conv``: into the dbeta half of the norm's [dgamma | dbeta] where that is possible"""
            slot = norm.dout_slot(nctx, dh16 and h.bf16)
            if slot is None:
                return conv.backward(d_y, [(h, 0)], grads, dx_bf16=dh16 and h.bf16), None
            dgb, dbeta, mask = slot
            conv.backward(d_y, [(h, 0)], grads, act_mask=mask, slope=0.2, dx_out=dbeta)
            return dbeta, dgb
        d_h1, dgb1 = down(self.c1, d_out, ctx["h1"], self.n1, ctx["n1"])
        # d(conv_0 output) is read by conv_0's weight / data gradient only: bf16 when the mixed-precision plan stores
        # that level's matrix-core tensors in bf16
        d_dx = self.n1.backward(ctx["n1"], d_h1, grads, None, False, dact_all.slice(hid * (k0 + 1), hid), dx_bf16=True, dgb=dgb1)
        d_h0, dgb0 = down(self.c0, d_dx, ctx["h0"], self.n0, ctx["n0"])
        if (self.learned and ctx["n0"]["z"] is not None and ctx["ns"]["z"] is not None and not x.bf16 and
                os.environ.get("HRV_NORM_BWD2", "0") != "0"):
            # norm_0 and norm_s normalise the same x: one pass per stage over it, dx = dx_0 + dx_s written once (opt-in: bit-identical,
            # 29 % fewer bytes -- and measured 10-17 % SLOWER than the two sequential calls at up_2..up_4: 168 / 132 registers leave
            # three waves per SIMD where the single kernels keep four, and capping them at 128 spills)
            d_hs = self.cs.backward(d_out, [(ctx["hs"], 0)], grads, dx_bf16=dh16 and ctx["hs"].bf16)
            a0, as_ = self.n0.norm_args(ctx["n0"], d_h0), self.ns_.norm_args(ctx["ns"], d_hs)
            d_x, dgb0, dgbs = T.norm_bwd2(x, a0, as_)
            self.n0.after_norm(ctx["n0"], dgb0, a0["dnoise_scale"], grads, dact_all.slice(hid * k0, hid))
            self.ns_.after_norm(ctx["ns"], dgbs, as_["dnoise_scale"], grads, dact_all.slice(0, hid))
            self.shared_backward(ctx["segx"], dact_all, grads)
            return d_x
        d_x = self.n0.backward(ctx["n0"], d_h0, grads, None, False, dact_all.slice(hid * k0, hid), dgb=dgb0)
        if self.learned:
            d_hs, dgbs = down(self.cs, d_out, ctx["hs"], self.ns_, ctx["ns"])
            self.ns_.backward(ctx["ns"], d_hs, grads, d_x, True, dact_all.slice(0, hid), dgb=dgbs)
        else:
            T.add_slice(d_out, d_x, True)
        self.shared_backward(ctx["segx"], dact_all, grads)
        return d_x


def noise_elems(gen: nn.Module, N: int) -> int:
    """Length of the flat draw ``noise_planes`` cuts up (every plane starts on a 16-byte boundary)."""
    return sum((3 if getattr(gen, name).learned_shortcut else 2) * ((N * (gen.sw << j) * (gen.sh << j) + 3) // 4 * 4)
               for j, name in enumerate(gen._blocks()))


def noise_planes(gen: nn.Module, N: int, zall: torch.Tensor):
    """{block name: [b x w x h x 1 plane per SPADE layer]} as views of one flat standard-normal draw, in block order."""
    out, off = {}, 0
    for j, name in enumerate(gen._blocks()):
        h, w = gen.sh << j, gen.sw << j
        zn = N * w * h
        planes = []
        for _ in range(3 if getattr(gen, name).learned_shortcut else 2):
            planes.append(zall[off:off + zn].view(N, w, h, 1))
            off += (zn + 3) // 4 * 4
        out[name] = planes
    return out


class GeneratorTrainPlan:
    def __init__(self, gen: nn.Module):
        self.gen = gen
        names = gen._blocks()
        self.names = names
        self.blocks = [BlockT(getattr(gen, n), n) for n in names]
        self.stems = [TConv(getattr(gen, f"conv_{i}"), 1, 1, f"conv_{i}") for i in range(len(names))]
        self.img = TConv(gen.conv_img, 1, 1, "conv_img")

    def forward(self, x: torch.Tensor, seg, noise, power_iteration: bool, save: bool = True):
        """``save``: keep what the backward needs (False: the no_grad forward of the discriminator step)."""
        gen = self.gen
        N, _, H, W = x.shape
        nb = len(self.names)
        top = nb - 1
        dev = x.device
        # one batched power iteration for every spectral-normalised convolution of the generator (four launches)
        # ... and one batched launch for the plan's weight packs (T.PackBatch)
        T.prepare_convs(self, [c for b in self.blocks for c in b.convs()] + list(self.stems) + [self.img] +
                        [n_.shared for b in self.blocks for n_ in b.norms()], power_iteration, backward=save,
                        extra_weights=[w_ for b in self.blocks for n_ in b.norms()
                                       for w_ in (n_.norm.conv_gamma.weight.data, n_.norm.conv_beta.weight.data)])
        # the bias / noise-scale vectors of every SPADE norm of the pass: one launch (58 launches of 4 us per iteration otherwise)
        norms_all = [n_ for b in self.blocks for n_ in b.norms()]
        if os.environ.get("HRV_VEC_PREP_MULTI", "1") != "0":
            pre = T.spade_vec_prep_multi([(n_.norm.conv_gamma.bias.data, n_.norm.conv_beta.bias.data, n_.norm.noise_scale.data) for n_ in norms_all])
            for n_, v in zip(norms_all, pre):
                n_._vec_pre = v
        try:
            return self._forward(x, seg, noise, save)
        finally:
            for n_ in norms_all:
                n_._vec_pre = None

    def _forward(self, x: torch.Tensor, seg, noise, save: bool):
        gen = self.gen
        N, _, H, W = x.shape
        nb = len(self.names)
        top = nb - 1
        dev = x.device
        xin = ops.to_nhwc(x)
        # mixed precision: the full-resolution stem (conv_7: 9 -> 16 channels over every pixel) reads a bf16 copy of the
        # input (matrix-core operand only) so that it runs on the thin-convolution kernel
        xin_top = ops.to_nhwc(x, bf16=True) if (T.MMA_BF16[0] and W % 4 == 0) else xin
        sg = seg if isinstance(seg, Act) else ops.to_nhwc(seg)
        ctxs = []
        stem_in = []
        cur = None
        if noise is None:
            # the noise planes of every SPADE layer of this forward from ONE generator launch (network_generator.py:103
            # draws b x w x h x 1 per layer: 23 launches a pass otherwise)
            noise = noise_planes(gen, N, torch.randn(noise_elems(gen, N), device=dev))
        for j, name in enumerate(self.names):
            blk = self.blocks[j]
            h, w = gen.sh << j, gen.sw << j
            shift = top - j
            cin = getattr(gen, name).input_nc
            xs, xsh = (xin_top, 0) if shift == 0 else (xin, -shift)
            if (0 < shift <= 2 and xin_top.bf16 and N * h * w >= 65536 and w % 4 == 0 and os.environ.get("HRV_STEM_DOWN", "1") != "0"):
                # conv_5 / conv_6 read x nearest-down-sampled by 4 / 2 (network_generator.py:226-238: F.interpolate, mode 'nearest' --
                # source pixel (y << shift, x << shift)): a materialised bf16 copy of those pixels (25 / 6 MB) puts the layer and its
                # weight gradient on the kernels of the full-resolution stem instead of the strided gather of the generic tiles
                f = 1 << shift
                xs, xsh = Act(xin_top.t[:, ::f, ::f, :].contiguous(), xin_top.C), 0
            stem_in.append((xs, xsh))
            if j == 0:
                cur = self.stems[0].forward([(xs, xsh)])
            elif isinstance(cur, ops.ActUp):
                self.stems[j].forward([(xs, xsh)], out=cur.hi)
            else:
                self.stems[j].forward([(xs, xsh)], out=cur.slice(cin - 16, 16))
            k = 3 if blk.learned else 2
            zs = [z.to(dev).contiguous() for z in noise[name]]
            assert len(zs) == k, (name, len(zs), k)
            if j == nb - 1:
                o, c = blk.forward(cur, sg, shift, zs, None, 0, ACT_LRELU, save)
            else:
                nxt_c = getattr(gen, self.names[j + 1]).input_nc
                if self.blocks[j + 1].reads_upsampled_input(N, h * 2, w * 2, nxt_c, sg, shift - 1):
                    # the next block reads cat(up2(this output), its stem) in place (ops.ActUp): the 4-fold fp32 copy of this
                    # block's output is never written and its readers fetch a quarter of the bytes
                    lo = ops.alloc(N, h, w, nxt_c - 16, dev)
                    _, c = blk.forward(cur, sg, shift, zs, lo, 0, ACT_NONE, save)
                    o = ops.ActUp(lo, ops.alloc(N, h * 2, w * 2, 16, dev))
                else:
                    nxt = ops.alloc(N, h * 2, w * 2, nxt_c, dev)
                    o, c = blk.forward(cur, sg, shift, zs, nxt.slice(0, nxt_c - 16), 1, ACT_NONE, save)
                    o = nxt
            c["shift"] = shift
            ctxs.append(c)
            cur = o
        img = self.img.forward([(cur, 0)], act=ACT_TANH)
        return ops.to_nchw(img), dict(blocks=ctxs, last=cur, img=img, stem_in=stem_in)

    def backward(self, ctx, d_img: torch.Tensor) -> Grads:
        grads: Grads = {}
        gen = self.gen
        img: Act = ctx["img"]
        dimg = ops.to_nhwc(d_img.contiguous())
        dpre = Act(T.tanh_bwd(dimg.t, img.t), 3)
        # conv_img reads lrelu(x_last): its dgrad carries that LeakyReLU's derivative
        # the gradient of a learned-shortcut block's output is read by conv_1's and conv_s's backward only (matrix
        # cores): stored in bf16 when that block's activations are (BlockT.wants_bf16_dout)
        dpre8 = None
        if ctx["last"].bf16:
            # conv_img's weight gradient (3 output channels over 3.1 M pixels per image: memory-bound): a bf16 copy of
            # d(pre-tanh), padded to one 16-byte group per pixel, lets it run on the LDS-DMA weight-gradient kernel
            t8 = torch.zeros(dpre.t.shape[:3] + (8,), dtype=torch.bfloat16, device=dpre.t.device)
            t8[..., :4].copy_(dpre.t)
            dpre8 = Act(t8, 3)
        d_cur = self.img.backward(dpre if dpre8 is None else dpre8, [(ctx["last"], 0)], grads, act_mask=ctx["last"],
                                  slope=0.2, dx_bf16=self.blocks[-1].wants_bf16_dout(ctx["blocks"][-1]))
        for j in range(len(self.names) - 1, -1, -1):
            blk, c = self.blocks[j], ctx["blocks"][j]
            d_x = blk.backward(c, d_cur, grads)
            cin = getattr(gen, self.names[j]).input_nc
            xin, xsh = ctx["stem_in"][j]
            if j == 0:
                self.stems[0].backward(d_x, [(xin, xsh)], grads, need_dx=False)
            else:
                d_stem = d_x.slice(cin - 16, 16)
                d_stem_w = None
                if xin.bf16 and xsh == 0 and not d_stem.bf16:
                    # a stem that reads a bf16 copy of its input at its own resolution (9 -> 16 channels over every pixel:
                    # memory-bound): a bf16 copy of its 16 gradient channels puts the weight gradient on the LDS-DMA kernel (both
                    # operands bf16): 0.91 -> ~0.2 ms at full resolution
                    d_stem_w = Act(d_stem.t[..., d_stem.coff:d_stem.coff + 16].to(torch.bfloat16), 16)
                self.stems[j].backward(d_stem, [(xin, xsh)], grads, need_dx=False, dy_wgrad=d_stem_w)
                d_cur = T.downsum2x2(d_x.slice(0, cin - 16), out_bf16=self.blocks[j - 1].wants_bf16_dout(ctx["blocks"][j - 1]))
        return grads


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gen, plan, x, seg, noise, *params):
        out, saved = plan.forward(x, seg, noise, power_iteration=True)
        ctx.plan, ctx.saved, ctx.params = plan, saved, params
        return out

    @staticmethod
    def backward(ctx, d_out):
        grads = ctx.plan.backward(ctx.saved, d_out)
        ctx.saved = None
        T.wgrad_join()
        return (None, None, None, None, None) + tuple(grads.get(p) for p in ctx.params)


def generator_train_forward(gen: nn.Module, x: torch.Tensor, seg, noise=None) -> torch.Tensor:
    ops.require_cuda(x, "SPADEGenerator.forward(x)")
    plan = getattr(gen, "_train_plan", None)
    if plan is None:
        plan = gen._train_plan = GeneratorTrainPlan(gen)
    params = [p for p in gen.parameters()]
    if not torch.is_grad_enabled():
        # train-mode forward under no_grad (train_generator.py:327-330): same math, incl. the
        # spectral-norm power iteration, nothing saved
        out, _ = plan.forward(x, seg, noise, power_iteration=True, save=False)
        return out
    return _GenFn.apply(gen, plan, x, seg, noise, *params)


# ----------------------------------------------------------------------------
# PatchGAN (network_generator.py:250-316)
# ----------------------------------------------------------------------------
# test hook: dropout keep-masks (already scaled by 1/(1-p), NCHW or NHWC) consumed in call order instead of fresh draws
DROP_MASKS: List[torch.Tensor] = []


class _EngineMode:
    """``with _EngineMode(fp32):`` -- the convolutions inside run on the fp32 matrix-core engine although the step is in mixed
    precision (T.MMA_BF16): a layer whose operand rounding the caller wants out of the comparison (HRV_D_F32_LAYERS)."""

    def __init__(self, fp32: bool):
        self.fp32 = fp32

    def __enter__(self):
        self.old = T.MMA_BF16[0]
        if self.fp32:
            T.MMA_BF16[0] = False

    def __exit__(self, *exc):
        T.MMA_BF16[0] = self.old
        return False


def _d_f32(layer: int, part: str, own_step: bool = True, scale: int = 0) -> bool:
    """Mixed precision: does convolution ``layer`` of a PatchGAN scale keep fp32 operands in ``part`` ("fwd": the forward, "bwd": its
    data and weight gradients)?  Default (round 6): the FORWARDS of layers 0, 1, 2 (model0 .. model2: everything in front of the last
    InstanceNorm) of ``discriminator_1`` (the half-resolution scale) in the discriminator's OWN step.  Measured (tools/d_f32_layers.py:
    the D half of the iteration at 2 x 1024x768 against torch autograd over the fp32 oracle; profiles/r06_d_f32_seeds.txt), minimum
    cosine over D's parameter gradients on the seeds 1 / 2 / 3 of the draw:
        every convolution on bf16 operands (amp O1's choice)      0.980 (seed 1; the oracle's own bf16-operand evaluation: 0.983)
        model1 forward (round 5's default)                        0.9915 / 0.9825 / 0.9903   -- seed 2 falls under 0.99
        model1 + model2 forwards                                  0.9938 / 0.9893 / 0.9927
        model0 + model1 forwards                                  0.9920 / 0.9823 / 0.9930
        model0 + model1 + model2 forwards (the default)           0.9975 / 0.9960 / 0.9973   (+ model3: the same)
    i.e. the rounding that matters is that of the WHOLE chain in front of the half-resolution scale's last InstanceNorm (a quarter
    of the pixels average it less than the full-resolution scale does); operand rounding in the gradients is invisible, and the
    full-resolution scale needs nothing.  Cost of the default: +0.25 ms of the 71.6 ms iteration, same box (mask 2: 71.57, mask 7:
    71.81; both scales or both passes through D cost 1.2 - 1.8 ms -- the generator step's gradient cosine is 0.998 without).
    HRV_D_F32_MASK=<bitmask of layer indices> (or HRV_D_F32_LAYERS=<k>: the first k layers), HRV_D_F32_PARTS=all|fwd|bwd,
    HRV_D_F32_SCOPE=dstep|always, HRV_D_F32_SCALES=<bitmask of discriminator_k>; HRV_D_F32_MASK=0: every convolution in bf16 (what
    amp O1 does, train_generator.py:186-190).  The tocg discriminator's plans (cond_train: DiscTrainPlan.from_sequential) carry no
    scale index: they count as scale 0 and keep bf16 operands under the default scale mask."""
    mask = int(os.environ.get("HRV_D_F32_MASK", "7") or 0) | ((1 << int(os.environ.get("HRV_D_F32_LAYERS", "0") or 0)) - 1)
    if not (mask >> layer) & 1:
        return False
    if not (int(os.environ.get("HRV_D_F32_SCALES", "2") or 0) >> scale) & 1:      # bit k = discriminator_k; default: the half-resolution scale
        return False
    if not own_step and os.environ.get("HRV_D_F32_SCOPE", "dstep") != "always":
        return False
    return os.environ.get("HRV_D_F32_PARTS", "fwd") in ("all", part)


class DiscTrainPlan:
    """One NLayerDiscriminator: conv0+lrelu, [SNconv, IN, lrelu] x (n_layers-1), conv_last."""

    def __init__(self, D: nn.Module, name: str):
        self.D = D
        self.layers = []
        for n in range(D.n_models):
            m = getattr(D, "model" + str(n))
            first = m[0]
            if isinstance(first, nn.Sequential):
                conv = first[0]
                self.layers.append(("in", TConv(conv, conv.stride[0], conv.padding[0], f"{name}.model{n}")))
            else:
                tc = (S2DConv(first, f"{name}.model{n}") if (n == 0 and S2DConv.fits(first)) else
                      TConv(first, first.stride[0], first.padding[0], f"{name}.model{n}"))
                self.layers.append(("lrelu" if len(m) > 1 else "plain", tc))

    @classmethod
    def from_sequential(cls, seq: nn.Sequential, name: str) -> "DiscTrainPlan":
        """The tocg discriminator keeps one flattened nn.Sequential per scale (networks.py:389-393):
        [conv, LReLU] + [conv, InstanceNorm2d, LReLU]* + [conv]."""
        self = cls.__new__(cls)
        self.D, self.layers = seq, []
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if not isinstance(m, nn.Conv2d):
                raise NotImplementedError(f"hr-viton_amd tocg discriminator: unsupported layer {type(m).__name__} "
                                          "(Ddropout / use_sigmoid / BatchNorm variants are not on the HIP path)")
            tc = S2DConv(m, f"{name}.{i}") if (not self.layers and S2DConv.fits(m)) else TConv(m, m.stride[0], m.padding[0], f"{name}.{i}")
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, nn.InstanceNorm2d):
                if nxt.affine or not isinstance(mods[i + 2], nn.LeakyReLU):
                    raise NotImplementedError("hr-viton_amd tocg discriminator: InstanceNorm2d(affine=False) + LeakyReLU")
                drop = mods[i + 3] if i + 3 < len(mods) and isinstance(mods[i + 3], nn.Dropout) else None
                if drop is not None:       # --Ddropout (networks.py:363-368)
                    self.layers.append(("in_drop", tc))
                    self.drop_p = drop.p
                    self.drop_mod = drop
                    i += 4
                else:
                    self.layers.append(("in", tc))
                    i += 3
            elif isinstance(nxt, nn.LeakyReLU):
                self.layers.append(("lrelu", tc))
                i += 2
            else:
                self.layers.append(("plain", tc))
                i += 1
        return self

    def refresh_s2d(self):
        for _, conv in self.layers:
            if isinstance(conv, S2DConv):
                conv.refresh()

    # ---- mixed precision with bf16-STORED feature maps (csrc/conv_s2.hip) -----------------------------------------------------
    def _s2_chain(self, a: Act, own: bool, sc: int) -> bool:
        return self._s2_chain_dims(a.N, a.H, a.W, a.Cp, a.bf16, own, sc)

    def _s2_chain_dims(self, aN: int, aH: int, aW: int, aCp: int, abf16: bool, own: bool, sc: int) -> bool:
        """Does this scale run on csrc/conv_s2.hip with its feature maps stored in bf16?  Mixed precision only, the reference's
        NLayerDiscriminator shape ([conv 4x4 s2 + LReLU] [SN conv 4x4 s2 + IN + LReLU]+ [conv 4x4 s1 -> 1]), every convolution of
        the scale on bf16 operands (a layer kept on fp32 operands by _d_f32 -- the half-resolution scale in D's own step -- keeps
        the whole scale on the fp32-stored path), even input extents, shapes the kernel serves.  HRV_D_BF16=0: off."""
        if not T.MMA_BF16[0] or os.environ.get("HRV_D_BF16", "1") == "0" or abf16 or len(self.layers) < 3:
            return False
        kinds = [k for k, _ in self.layers]
        if kinds[0] != "lrelu" or kinds[-1] != "plain" or any(k != "in" for k in kinds[1:-1]):
            return False
        c0 = self.layers[0][1]
        if not isinstance(c0, S2DConv) or c0._w2 is None or aH % 2 or aW % 2 or aCp != c0.Cq:
            return False
        if any(_d_f32(li, "bwd", own, sc) for li in range(len(self.layers))):
            return False
        split = [_d_f32(li, "fwd", own, sc) for li in range(len(self.layers))]
        # a forward kept on fp32 operands runs as three bf16 products over split operands (hi*hi + lo*hi + hi*lo: ~16 mantissa
        # bits; HRV_D_SPLIT3=0: such a scale stays on the fp32-stored path and the fp32 matrix-core engine)
        if any(split) and (os.environ.get("HRV_D_SPLIT3", "1") == "0" or (4 * c0.Cq) % 8 or c0.Cq % 2):
            return False
        H, W, N = aH // 2 + 1, aW // 2 + 1, aN
        cin = c0.conv.out_channels
        if cin % 64 or not T.conv_s2_ok(T.S2_CELLS, 4 * c0.Cq, cin, 0, N, H, W):
            return False
        for _, conv in self.layers[1:-1]:
            m = conv.conv
            if (tuple(m.kernel_size), tuple(m.stride), tuple(m.padding)) != ((4, 4), (2, 2), (2, 2)) or m.in_channels != cin:
                return False
            Ho, Wo = H // 2 + 1, W // 2 + 1
            if not (T.conv_s2_ok(T.S2_FWD, cin, m.out_channels, 0, N, Ho, Wo) and
                    T.conv_s2_ok(T.S2_DGRAD, m.out_channels, 4 * cin, cin, N // 2, H, W)):
                return False
            H, W, cin = Ho, Wo, m.out_channels
        return True

    def s2_fwd_jobs(self, own: bool, sc: int):
        """(layer index, conv_s2_pack_multi job) of the scale's forward: the caller packs them in one launch and leaves the streams in
        ``self._s2_pk`` (MultiscaleDTrainPlan.forward)."""
        split = [_d_f32(li, "fwd", own, sc) for li in range(len(self.layers))]
        jobs = []
        for li, (kind, conv) in enumerate(self.layers[:-1]):
            m = conv.conv
            if li == 0:
                jobs.append((0, (T.S2_CELLS, conv.w2, 4 * conv.Cq * (3 if split[0] else 1), m.out_channels, 0, conv.sigma, split[0])))
            else:
                jobs.append((li, (T.S2_FWD, conv.wparam.data, m.in_channels * (3 if split[li] else 1), m.out_channels, 0, conv.sigma, split[li])))
        return jobs

    def s2_bwd_jobs(self):
        return [(i, (T.S2_DGRAD, conv.wparam.data, conv.conv.out_channels, 4 * conv.conv.in_channels, conv.conv.in_channels, conv.sigma, False))
                for i, (kind, conv) in enumerate(self.layers) if kind == "in"]

    def _forward_s2(self, a: Act, own: bool, sc: int):
        """The scale's forward with f0, f1, ... stored in bf16 (the last InstanceNorm output stays fp32: the one-channel
        convolution behind it is a dot-product kernel over fp32).  A layer _d_f32 keeps on fp32 operands reads its source as
        [hi | lo | hi] (T.split3 of the fp32 feature in front of it) against weights packed [hi | hi | lo]; the backward's bf16
        operand of that feature is the hi third."""
        feats, ctx = [], []
        n_in = len(self.layers) - 2
        split = [_d_f32(li, "fwd", own, sc) for li in range(len(self.layers))]
        sbf = None            # the bf16 operand of the feature in front of the current layer (bf16 Act, or the hi third of its split)
        pre = getattr(self, "_s2_pk", None) or {}      # streams packed ahead, one launch for both scales
        self._s2_pk = {}
        for li, (kind, conv) in enumerate(self.layers):
            b = conv.bparam
            bias = None if b is None else b.data
            nxt_split = li + 1 < len(self.layers) - 1 and split[li + 1]        # the next 4x4 stride-2 layer reads split operands
            if li == 0:
                Cout, K0 = conv.conv.out_channels, 4 * conv.Cq
                # (one cell of zeros below / right of the image: out = cells + 1 makes this a 'same' 2x2 convolution -- the shape the
                #  LDS-DMA weight-gradient kernel serves)
                src2 = T.space_to_depth2_cells(a, a.H // 2 + 1, a.W // 2 + 1, split3=split[0])
                a2 = Act(src2.t, K0, 0)
                f = ops.alloc(a.N, a.H // 2 + 1, a.W // 2 + 1, Cout, a.t.device, bf16=not nxt_split)
                pk = pre.pop(0, None)
                if pk is None:
                    pk = T.conv_s2_pack(T.S2_CELLS, conv.w2, src2.C, Cout, sigma=conv.sigma, split3=split[0])
                T.conv_s2(T.S2_CELLS, src2, pk, Cout, f, bias=bias, act=ACT_LRELU, slope=0.2, name=conv.name,
                          flops=2.0 * f.N * f.H * f.W * Cout * conv.conv.in_channels * 16 * (3 if split[0] else 1))
                ctx.append(dict(src=a, a2=a2, f=f, s2=True))
            elif kind == "in":
                Cout, cin = conv.conv.out_channels, conv.conv.in_channels
                c = ops.alloc(a.N, a.H // 2 + 1, a.W // 2 + 1, Cout, a.t.device)
                pk = pre.pop(li, None)
                if pk is None:
                    pk = T.conv_s2_pack(T.S2_FWD, conv.wparam.data, src.C, Cout, sigma=conv.sigma, split3=split[li])
                T.conv_s2(T.S2_FWD, src, pk, Cout, c, bias=bias, name=conv.name,
                          flops=2.0 * c.N * c.H * c.W * Cout * cin * 16 * (3 if split[li] else 1))
                mean, rstd = ops.instnorm_stats(c)
                if li < n_in and not nxt_split:
                    f = T.instnorm_apply_bf16(c, mean, rstd, ACT_LRELU, 0.2)
                else:
                    f = ops.instnorm_apply(c, mean, rstd, ACT_LRELU, 0.2)
                ctx.append(dict(src=sbf, c=c, mean=mean, rstd=rstd, f=f, s2=True))
            else:
                with _EngineMode(split[li]):
                    f = conv.forward([(a, 0)], act=ACT_NONE)
                ctx.append(dict(src=a, f=f))
            feats.append(f)
            # what the next layer multiplies / the backward reads as this feature's bf16 operand
            if nxt_split:
                src = T.split3(f)
                sbf = Act(src.t, f.C, 0)
            else:
                src = sbf = f
            a = f
        return feats, ctx

    def _backward_s2(self, ctx, dfeats: List[Optional[Act]], grads: Grads, need_dx: bool, rows: Optional[int], need_w: bool) -> Optional[Act]:
        def cut(a):
            if rows is None or a is None:
                return a
            return Act(a.t[:rows], a.C, a.coff) if isinstance(a, Act) else a[:rows]

        def param_grads(conv, G):
            if conv.spectral:
                dwo = grad_buffer(conv.wparam)
                T.spectral_grad(G, conv.wparam.data, conv.u, conv.v, conv.sigma, dwo)
                _acc(grads, conv.wparam, dwo)
            else:
                _acc(grads, conv.wparam, G)

        d_next: Optional[Act] = None
        tap_in_dnext = False
        pre = getattr(self, "_s2_pkd", None) or {}
        self._s2_pkd = {}
        for i in range(len(self.layers) - 1, -1, -1):
            kind, conv = self.layers[i]
            c = {k: cut(v) for k, v in ctx[i].items() if k != "s2"}
            d = dfeats[i]
            if d is None and d_next is None:
                continue
            if d is None or tap_in_dnext:
                d = d_next
            elif d_next is not None:
                T.add_slice(d_next, d, True)
            if i == len(self.layers) - 1:          # the one-channel convolution: as on the fp32-stored path
                tap = dfeats[i - 1]
                tap_in_dnext = tap is not None and tap.t.dtype == torch.float32 and tap.t.shape[:3] == c["src"].t.shape[:3]
                d_next = conv.backward(d, [(c["src"], 0)], grads, need_dx=True, need_w=need_w, add=tap if tap_in_dnext else None)
                continue
            if kind == "in":
                d_c, _ = T.norm_bwd(c["c"], c["mean"], c["rstd"], d, act=ACT_LRELU, slope=0.2, out=c["f"], dx_bf16=True)
                src = c["src"]                      # bf16 feature of the layer in front
                m = conv.conv
                w = conv.wparam.data
                if need_w:
                    G = torch.empty_like(w) if conv.spectral else grad_buffer(conv.wparam)
                    db = grad_buffer(conv.bparam) if conv.bparam is not None else None
                    T.conv_wgrad(d_c, src, 0, 0, m.in_channels, 4, 4, 2, 2, G, name=conv.name + ".wgrad", dbias=db)
                    param_grads(conv, G)
                    if db is not None:
                        _acc(grads, conv.bparam, db)
                tap = dfeats[i - 1]
                tap_ok = tap is not None and tap.t.shape[:3] == src.t.shape[:3] and tap.C == src.C
                dx = ops.alloc(src.N, src.H, src.W, src.C, src.t.device, bf16=True)
                pk = pre.pop(i, None)
                if pk is None:
                    pk = T.conv_s2_pack(T.S2_DGRAD, w, m.out_channels, 4 * m.in_channels, m.in_channels, sigma=conv.sigma)
                # the gradient of the layer's input feature: + its feature-matching tap; the feature behind model0 is LeakyReLU(pre):
                # its derivative rides along as the mask (the features behind an InstanceNorm get theirs in norm_bwd)
                T.conv_s2(T.S2_DGRAD, d_c, pk, 4 * m.in_channels, dx, Cph=m.in_channels, residual=tap if tap_ok else None,
                          mask=src if i == 1 else None, mask_slope=0.2, name=conv.name + ".dgrad",
                          flops=2.0 * d_c.N * d_c.H * d_c.W * m.out_channels * m.in_channels * 16)
                if tap is not None and not tap_ok:
                    raise AssertionError("PatchGAN bf16 path: feature-matching tap of an unexpected shape")
                tap_in_dnext = tap is not None      # (for i == 1 the sum is already multiplied by LeakyReLU'(f0))
                d_next = dx
                continue
            # model0 (its LeakyReLU derivative was applied by model1's data gradient)
            a2 = c["a2"]
            w = conv.wparam.data
            Cout, cin = w.shape[0], w.shape[1]
            if d_next is None or d is not d_next:      # (a tap of f0 was summed, and LeakyReLU'(f0) applied, by model1's data gradient)
                raise AssertionError("PatchGAN bf16 path: model0's output gradient comes from model1's data gradient")
            if need_w:
                dw2 = torch.empty_like(conv._w2)
                db = grad_buffer(conv.bparam) if conv.bparam is not None else None
                T.conv_wgrad(d, a2, 0, 0, 4 * conv.Cq, 2, 2, 1, 1, dw2.view(Cout, 4 * conv.Cq, 2, 2), name=conv.name + ".wgrad", dbias=db)
                G = dw2[:, :, :, :cin].permute(0, 3, 4, 1, 5, 2).reshape(Cout, cin, 4, 4)      # back to (co, c, kh, kw)
                param_grads(conv, G)
                if db is not None:
                    _acc(grads, conv.bparam, db)
            d_next = None
            if need_dx:
                d2 = T.conv_dgrad(d, conv.w2, c["src"].H // 2, c["src"].W // 2, 1, 1, sigma=conv.sigma, name=conv.name + ".dgrad",
                                  batch=getattr(conv, "pack_batch", None))
                d_next = T.depth_to_space2(d2, c["src"].C)
        return d_next

    def forward(self, a: Act, power_iteration: bool, prepared: bool = False):
        feats, ctx = [], []
        if not prepared:
            self.refresh_s2d()
            T.prepare_convs(self, [conv for _, conv in self.layers], power_iteration)
        own = getattr(self, "own_step", True)      # (False: the generator step's pass through D -- its parameter gradients are discarded)
        sc = getattr(self, "scale_index", 0)
        if self._s2_chain(a, own, sc):
            return self._forward_s2(a, own, sc)
        for li, (kind, conv) in enumerate(self.layers):
            if kind in ("in", "in_drop"):
                with _EngineMode(_d_f32(li, "fwd", own, sc)):
                    c = conv.forward([(a, 0)])
                mean, rstd = ops.instnorm_stats(c)
                f = ops.instnorm_apply(c, mean, rstd, ACT_LRELU, 0.2)
                entry = dict(src=a, c=c, mean=mean, rstd=rstd, f=f)
                if kind == "in_drop" and self.drop_mod.training:
                    # nn.Dropout(p): keep mask scaled by 1/(1-p); DROP_MASKS lets a test inject the draws
                    if DROP_MASKS:
                        m = DROP_MASKS.pop(0).to(f.t.device)
                        m = m.permute(0, 2, 3, 1).contiguous() if m.shape != f.t.shape else m
                    else:
                        m = torch.empty_like(f.t).bernoulli_(1.0 - self.drop_p).mul_(1.0 / (1.0 - self.drop_p))
                    f = Act(f.t.clone(), f.C)          # f (pre-dropout) is kept for the LeakyReLU derivative
                    T.mul_(f, m)
                    entry["mask"] = m
                ctx.append(entry)
            else:
                with _EngineMode(_d_f32(li, "fwd", own, sc)):
                    f = conv.forward([(a, 0)], act=ACT_LRELU if kind == "lrelu" else ACT_NONE)
                ctx.append(dict(src=a, f=f))
            feats.append(f)
            a = f
        return feats, ctx

    def backward(self, ctx, dfeats: List[Optional[Act]], grads: Grads, need_dx: bool, rows: Optional[int] = None,
                 need_w: bool = True) -> Optional[Act]:
        """``rows``: only the first ``rows`` samples of the batch carry a gradient (the fake half in the
        generator step: the real half is a detached target) -- every saved tensor is cut to that prefix."""
        def cut(a):
            if rows is None or a is None:
                return a
            return Act(a.t[:rows], a.C, a.coff) if isinstance(a, Act) else a[:rows]
        if ctx and ctx[0].get("s2"):
            return self._backward_s2(ctx, dfeats, grads, need_dx, rows, need_w)
        d_next: Optional[Act] = None   # gradient flowing back into feats[i] from layer i+1
        tap_in_dnext = False           # dfeats[i] already summed into d_next by layer i+1's data-gradient epilogue
        for i in range(len(self.layers) - 1, -1, -1):
            kind, conv = self.layers[i]
            c = {k: cut(v) for k, v in ctx[i].items()}
            d = dfeats[i]
            if d is None and d_next is None:
                continue
            if d is None or tap_in_dnext:
                d = d_next
            elif d_next is not None:
                T.add_slice(d_next, d, True)
            if kind in ("in", "in_drop"):
                if c.get("mask") is not None:
                    T.mul_(d, c["mask"])
                d_c, _ = T.norm_bwd(c["c"], c["mean"], c["rstd"], d, act=ACT_LRELU, slope=0.2, out=c["f"])
            elif kind == "lrelu":
                T.act_bwd_(d, c["f"], ACT_LRELU, 0.2)
                d_c = d
            else:
                d_c = d
            # the feature-matching gradient of this layer's INPUT feature rides along with the data gradient
            tap = dfeats[i - 1] if i > 0 else None
            tap_in_dnext = tap is not None and tap.t.dtype == torch.float32 and tap.t.shape[:3] == c["src"].t.shape[:3]
            with _EngineMode(_d_f32(i, "bwd", need_w, getattr(self, "scale_index", 0))):
                d_next = conv.backward(d_c, [(c["src"], 0)], grads, need_dx=(i > 0 or need_dx), need_w=need_w,
                                       add=tap if tap_in_dnext else None)
        return d_next


class MultiscaleDTrainPlan:
    def __init__(self, msd: nn.Module):
        self.msd = msd
        self.plans = [DiscTrainPlan(D, f"discriminator_{k}") for k, D in enumerate(msd.children())]

    def forward(self, inp: Optional[torch.Tensor], power_iteration: bool, a: Optional[Act] = None):
        """``a``: the input already as an NHWC activation (the [fake ; real] pair assembled by _DiscPairFn)."""
        if a is None:
            a = ops.to_nhwc(inp)
        feats_all, ctxs, inputs = [], [], []
        for p in self.plans:
            p.refresh_s2d()
        T.prepare_convs(self, [conv for p in self.plans for _, conv in p.layers], power_iteration)
        own = not getattr(self.msd, "_hrv_discard_param_grads", False)
        # the weight streams of the scales that run on csrc/conv_s2.hip: one launch for all of them
        if os.environ.get("HRV_S2_PACK_MULTI", "1") != "0":
            dims, jobs, owners = (a.N, a.H, a.W, a.Cp, a.bf16), [], []
            for k, p in enumerate(self.plans):
                p._s2_pk = {}
                if p._s2_chain_dims(*dims, own, k):
                    for li, job in p.s2_fwd_jobs(own, k):
                        jobs.append(job)
                        owners.append((p, li))
                dims = (dims[0], (dims[1] - 1) // 2 + 1, (dims[2] - 1) // 2 + 1, dims[3], False)
            for (p, li), buf in zip(owners, T.conv_s2_pack_multi(jobs)):
                p._s2_pk[li] = buf
        for k, p in enumerate(self.plans):
            inputs.append(a)
            p.own_step, p.scale_index = own, k
            feats, c = p.forward(a, power_iteration, prepared=True)
            feats_all.append(feats)
            ctxs.append(c)
            if k + 1 < len(self.plans):
                a = ops.avgpool3x3s2(a)
        return feats_all, dict(ctxs=ctxs, inputs=inputs)

    def backward(self, ctx, dfeats_all, need_dx: bool, rows: Optional[int] = None, need_w: bool = True):
        grads: Grads = {}
        d_in_next: Optional[Act] = None
        if os.environ.get("HRV_S2_PACK_MULTI", "1") != "0":
            jobs, owners = [], []
            for k, p in enumerate(self.plans):
                p._s2_pkd = {}
                if ctx["ctxs"][k] and ctx["ctxs"][k][0].get("s2"):
                    for i, job in p.s2_bwd_jobs():
                        jobs.append(job)
                        owners.append((p, i))
            for (p, i), buf in zip(owners, T.conv_s2_pack_multi(jobs)):
                p._s2_pkd[i] = buf
        for k in range(len(self.plans) - 1, -1, -1):
            a = ctx["inputs"][k]
            if rows is not None:
                a = Act(a.t[:rows], a.C, a.coff)
            d_a = self.plans[k].backward(ctx["ctxs"][k], dfeats_all[k], grads, need_dx, rows, need_w)
            if need_dx:
                if d_a is None:
                    d_a = Act(torch.zeros_like(a.t), a.C)
                if d_in_next is not None:
                    T.avgpool3x3s2_bwd(d_in_next, a.H, a.W, dx=d_a, accumulate=True)
                d_in_next = d_a
        return grads, d_in_next


def _nchw_view(f: Act) -> torch.Tensor:
    """NHWC activation as an NCHW tensor: a zero-copy channels-last view when no channel padding is involved."""
    if f.coff == 0 and f.C == f.cstride:
        return f.t.permute(0, 3, 1, 2)
    return ops.to_nchw(f)


def _nhwc_act(d: torch.Tensor, C: int) -> Act:
    """Incoming NCHW gradient as an NHWC Act: zero-copy when it is channels-last (as the HIP losses return it)."""
    if d.dim() == 4 and d.shape[1] == _ceil4(C) == C and d.is_contiguous(memory_format=torch.channels_last) \
            and not d.is_contiguous():
        return Act(d.permute(0, 2, 3, 1), C)
    return ops.to_nhwc(d.contiguous())


class _DiscFn(torch.autograd.Function):
    """outputs: per scale, per layer the feature map -- or, with ``split``, its fake (first) half and its real
    (second) half as two outputs, so the training script needs no torch slicing and the backward sees directly
    which half carries a gradient (generator step: the fake half only -> half-batch backward)."""

    @staticmethod
    def forward(ctx, msd, plan, inp, split, *params):
        feats_all, saved = plan.forward(inp, power_iteration=True)
        ctx.plan, ctx.saved, ctx.params, ctx.split = plan, saved, params, split
        # outputs without a gradient (detached real-half features, the feature maps of the D step) arrive as None, not as
        # zero tensors the size of the feature maps: the backward below skips them (and takes the half-batch path)
        ctx.set_materialize_grads(False)
        # generator step: the discriminator's own parameter gradients of loss_G are thrown away by the training script
        # (optimizer_dis.zero_grad() before the D backward, train_generator.py:354) -- the weight / bias gradient
        # kernels and the spectral-norm transform are skipped when the caller says so (pipeline.generator_train_step)
        ctx.need_w = not getattr(msd, "_hrv_discard_param_grads", False)
        ctx.shapes = [[(f.N, f.H, f.W, f.C) for f in fs] for fs in feats_all]
        ctx.in_shape = tuple(inp.shape)
        return _disc_outputs(feats_all, split)

    @staticmethod
    def backward(ctx, *d_outs):
        grads, d_in, rows = _disc_backward(ctx, d_outs, ctx.needs_input_grad[2])
        T.wgrad_join()
        d_inp = None
        if d_in is not None:
            d_inp = ops.to_nchw(d_in)
            if rows is not None:      # the real half of the input gets a zero gradient
                full = torch.empty(ctx.in_shape, dtype=d_inp.dtype, device=d_inp.device)
                full[:rows] = d_inp
                full[rows:].zero_()       # (only the half that needs it is filled)
                d_inp = full
        return (None, None, d_inp, None) + tuple(grads.get(p) for p in ctx.params)


def _disc_backward(ctx, d_outs, need_dx: bool):
    """Shared by _DiscFn / _DiscPairFn: output gradients -> plan backward.  Returns (parameter gradients, d(input) as an NHWC
    Act or None, rows: the leading samples d(input) covers when only the fake half carried a gradient, else None)."""
    it = iter(d_outs)
    dfeats_all, rows = [], None
    if ctx.split:
        pairs = [[(next(it), next(it)) for _ in fs] for fs in ctx.shapes]
        only_fake = all(dr is None for fs in pairs for _, dr in fs)
        half = ctx.shapes[0][0][0] // 2
        rows = half if only_fake else None
        for fs, shp in zip(pairs, ctx.shapes):
            row = []
            for (df, dr), (n, h, w, c) in zip(fs, shp):
                if df is None and dr is None:
                    row.append(None)
                elif only_fake:
                    row.append(_nhwc_act(df, c))
                else:   # both halves (discriminator step: hinge terms on the last, tiny, maps)
                    full = ops.alloc(n, h, w, c, (df if df is not None else dr).device)
                    full.t.zero_()
                    if df is not None:
                        T.add_slice(_nhwc_act(df, c), Act(full.t[:half], c), False)
                    if dr is not None:
                        T.add_slice(_nhwc_act(dr, c), Act(full.t[half:], c), False)
                    row.append(full)
            dfeats_all.append(row)
    else:
        for fs in ctx.shapes:
            row = []
            for (n, h, w, c) in fs:
                d = next(it)
                row.append(None if d is None else _nhwc_act(d, c))
            dfeats_all.append(row)
    grads, d_in = ctx.plan.backward(ctx.saved, dfeats_all, need_dx, rows, ctx.need_w)
    ctx.saved = None
    return grads, (d_in if need_dx else None), rows


def _disc_outputs(feats_all, split: bool):
    outs = []
    for fs in feats_all:
        for f in fs:
            v = _nchw_view(f)
            if split:
                h = v.shape[0] // 2
                outs += [v[:h], v[h:]]
            else:
                outs.append(v)
    return tuple(outs)


class _DiscPairFn(torch.autograd.Function):
    """The [fake ; real] discriminator pass of train_generator.py:283-295 / :327-345 without the three torch.cat's and the
    layout round trip: the PatchGAN input cat((parse, image), 1) of both halves is assembled NHWC by one kernel per half
    (hrv_concat_nhwc_nchw_f32) straight from the label-map activation and the two NCHW images, and the backward returns
    d(fake image) only -- the label map and the real image have no gradient.  Same outputs as _DiscFn with ``split``."""

    @staticmethod
    def forward(ctx, msd, plan, parse7, fake, real, *params):
        N, Cb, H, W = fake.shape
        assert tuple(real.shape) == tuple(fake.shape) and (parse7.N, parse7.H, parse7.W) == (N, H, W) and not parse7.bf16
        Ca = parse7.C
        a = Act(torch.empty((2 * N, H, W, _ceil4(Ca + Cb)), dtype=torch.float32, device=fake.device), Ca + Cb)
        T.concat_nhwc_nchw(parse7, fake.detach(), Act(a.t[:N], Ca + Cb))
        T.concat_nhwc_nchw(parse7, real.detach(), Act(a.t[N:], Ca + Cb))
        feats_all, saved = plan.forward(None, power_iteration=True, a=a)
        ctx.plan, ctx.saved, ctx.params, ctx.split = plan, saved, params, True
        ctx.set_materialize_grads(False)
        ctx.need_w = not getattr(msd, "_hrv_discard_param_grads", False)
        ctx.shapes = [[(f.N, f.H, f.W, f.C) for f in fs] for fs in feats_all]
        ctx.cut = (N, Ca, Cb)
        return _disc_outputs(feats_all, True)

    @staticmethod
    def backward(ctx, *d_outs):
        grads, d_in, rows = _disc_backward(ctx, d_outs, ctx.needs_input_grad[3])
        T.wgrad_join()
        N, Ca, Cb = ctx.cut
        d_fake = None
        if d_in is not None:
            d_fake = ops.to_nchw(Act(d_in.t[:N], Ca + Cb), Ca, Cb)      # channels [Ca, Ca+Cb) of the fake half
        return (None, None, None, d_fake, None) + tuple(grads.get(p) for p in ctx.params)


def discriminator_train_forward_pair(msd: nn.Module, parse7: Act, fake: torch.Tensor, real: torch.Tensor):
    """discriminator(cat((cat((parse, fake), 1), cat((parse, real), 1)), 0)) of train_generator.py:283-295, returning
    (pred_fake, pred_real) like ``discriminator_train_forward(..., split=True)`` -- see _DiscPairFn."""
    ops.require_cuda(fake, "MultiscaleDiscriminator.forward_pair")
    plan = getattr(msd, "_train_plan", None)
    if plan is None:
        plan = msd._train_plan = MultiscaleDTrainPlan(msd)
    params = [p for p in msd.parameters()]
    nD = len(plan.plans)
    flat = list(_DiscPairFn.apply(msd, plan, parse7, fake, real, *params))     # (under no_grad: no graph is recorded)

    def group(seq):
        per = len(seq) // nD
        res = [seq[k * per:(k + 1) * per] for k in range(nD)]
        return [[r[-1]] for r in res] if msd.no_ganFeat_loss else res
    return group(flat[0::2]), group(flat[1::2])


def discriminator_train_forward(msd: nn.Module, inp: torch.Tensor, split: bool = False):
    """``split``: the batch is [fake ; real] (train_generator.py:283-295); returns (pred_fake, pred_real), each a
    list (scales) of lists (layers), as zero-copy halves of the feature maps."""
    ops.require_cuda(inp, "MultiscaleDiscriminator.forward")
    plan = getattr(msd, "_train_plan", None)
    if plan is None:
        plan = msd._train_plan = MultiscaleDTrainPlan(msd)
    params = [p for p in msd.parameters()]
    nD = len(plan.plans)
    if not torch.is_grad_enabled():
        feats_all, _ = plan.forward(inp, power_iteration=True)
        flat = []
        for fs in feats_all:
            for f in fs:
                v = _nchw_view(f)
                flat += [v[: v.shape[0] // 2], v[v.shape[0] // 2:]] if split else [v]
    else:
        flat = list(_DiscFn.apply(msd, plan, inp, split, *params))

    def group(seq):
        per = len(seq) // nD
        res = [seq[k * per:(k + 1) * per] for k in range(nD)]
        return [[r[-1]] for r in res] if msd.no_ganFeat_loss else res

    if split:
        return group(flat[0::2]), group(flat[1::2])
    return group(flat)
