"""Training-mode forward/backward of the condition generator and its discriminator on the HIP
path (train_condition.py:113-286; networks.py:98-198 with ``tocg.train()``).

ConditionGenerator.forward in training mode is ONE ``torch.autograd.Function``.  Inside it the
network runs on NHWC activations through the training kernels and records a *tape* of backward
closures (the tocg graph is a DAG -- encoder features feed the 1x1 laterals, the skip
concatenations and the next stage; T1 feeds the warp and the next level -- so gradients are
accumulated per tensor instead of following a hand-ordered plan as gen_train.py does).
Backward replays the tape in reverse: MFMA weight/data gradients, batch-statistics BatchNorm
backward, the adjoints of the bilinear x2 resize and of the fused flow warp (atomic scatter into
the warped feature's gradient + coordinate gradient into the flow pyramid).

BatchNorm uses per-GPU batch statistics (north star: sync_batchnorm is replaced by per-GPU BN);
running_mean / running_var / num_batches_tracked are updated in place like nn.BatchNorm2d does.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from . import train_ops as T
from .gen_train import DiscTrainPlan, Grads, TConv, _acc, grad_buffer
from .ops import ACT_NONE, ACT_RELU, Act, _stream


class Var:
    """Tape variable: an NHWC activation and its lazily allocated gradient accumulator."""
    __slots__ = ("a", "g", "req")

    def __init__(self, a: Act, req: bool = True):
        self.a, self.g, self.req = a, None, req

    def add_grad(self, d: Act, owned: bool):
        """g += d.  ``owned``: the caller gives the buffer away (adopted when it is the first
        contribution and a whole dense tensor of the right width)."""
        if not self.req:
            return
        if self.g is None:
            if owned and d.coff == 0 and d.C == self.a.C and d.cstride == d.Cp:
                self.g = d
                return
            self.g = ops.alloc(self.a.N, self.a.H, self.a.W, self.a.C, self.a.t.device)
            T.add_slice(d, self.g, False)
        else:
            T.add_slice(d, self.g, True)

    def grad_or_zero(self) -> Act:
        if self.g is None:
            self.g = Act(torch.zeros_like(self.a.t), self.a.C) if self.a.coff == 0 and self.a.cstride == self.a.Cp \
                else Act(torch.zeros((self.a.N, self.a.H, self.a.W, self.a.Cp), dtype=torch.float32,
                                     device=self.a.t.device), self.a.C)
        return self.g


class FlowVar:
    """A dense [N,h,w,2] flow (the reference's layout, networks.py:123) and its gradient."""
    __slots__ = ("t", "g")

    def __init__(self, t: torch.Tensor):
        self.t, self.g = t, None

    def add_grad(self, d: torch.Tensor, owned: bool):
        if self.g is None:
            self.g = d if owned else d.clone()
        else:
            n = d.numel()
            assert n % 4 == 0
            T.add_slice(Act(d.view(1, 1, n // 4, 4), 4), Act(self.g.view(1, 1, n // 4, 4), 4), True)


class Tape:
    def __init__(self):
        self.ops: List[Callable[[], None]] = []
        self.grads: Grads = {}

    def record(self, fn: Callable[[], None]):
        self.ops.append(fn)

    def backward(self) -> Grads:
        for fn in reversed(self.ops):
            fn()
        self.ops = []
        return self.grads


# ------------------------------------------------------------------------------------------ ops
def conv(tape: Tape, tc: TConv, srcs: Sequence[Var], act: int = ACT_NONE) -> Var:
    """Convolution over the channel concatenation of ``srcs`` (+ bias, + activation)."""
    acts = [(v.a, 0) for v in srcs]
    out = Var(tc.forward(acts, act=act))

    def bwd():
        d = out.g
        out.g = None
        if d is None:
            return
        if act != ACT_NONE:
            T.act_bwd_(d, out.a, act, 0.2)
        need_dx = any(v.req for v in srcs)
        # a source that already holds a gradient (a ResBlock's residual: bn2's backward delivered d(out) to it before conv_1's data
        # gradient exists, networks.py:196-198): the data gradient adds it in its epilogue (the engine's residual slot) instead of a
        # separate accumulation pass over the tensor (78 add_slice launches, 21.6 ms per configs[2] iteration before)
        g0 = srcs[0].g if len(srcs) == 1 else None
        fuse = (g0 is not None and srcs[0].req and tc.stride == 1 and not g0.bf16 and not d.bf16 and g0.coff == 0 and
                g0.C == srcs[0].a.C and g0.cstride == g0.Cp and os.environ.get("HRV_TAPE_FUSE_ACC", "1") != "0")
        dx = tc.backward(d, acts, tape.grads, need_dx=need_dx, add=g0 if fuse else None)
        if dx is None:
            return
        if fuse:
            srcs[0].g = dx                      # = conv^T(d) + the gradient that was there
        elif len(srcs) == 1:
            srcs[0].add_grad(dx, owned=True)
        else:
            c0 = 0
            for v in srcs:
                v.add_grad(dx.slice(c0, v.a.C), owned=False)
                c0 += v.a.C

    tape.record(bwd)
    return out


def bn_act(tape: Tape, bn: nn.BatchNorm2d, x: Var, act: int, residual: Optional[Var] = None) -> Var:
    """out = act(BatchNorm_train(x) (+ residual)); updates the running statistics in place."""
    mom = 0.1 if bn.momentum is None else bn.momentum
    track = bn.track_running_stats and bn.running_mean is not None
    st = T.bn_train_stats(x.a, bn.weight.data if bn.affine else None, bn.bias.data if bn.affine else None, bn.eps, mom,
                          bn.running_mean if track else None, bn.running_var if track else None)
    if track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    out = Var(T.affine_act(x.a, st.scale, st.shift, act, None if residual is None else residual.a))

    def bwd():
        d = out.g
        out.g = None
        if d is None:
            return
        if act != ACT_NONE:
            T.act_bwd_(d, out.a, act, 0.2)
        dev = d.t.device
        dg = grad_buffer(bn.weight) if bn.affine else None
        db = grad_buffer(bn.bias) if bn.affine else None
        dx = T.bn_bwd(d, x.a, st, dg, db)
        if bn.affine:
            _acc(tape.grads, bn.weight, dg)
            _acc(tape.grads, bn.bias, db)
        x.add_grad(dx, owned=True)
        if residual is not None:
            residual.add_grad(d, owned=True)

    tape.record(bwd)
    return out


def up2(tape: Tape, x: Var, addend: Optional[Var] = None, mode: str = "bilinear") -> Var:
    """F.interpolate(x, scale_factor=2, mode) (+ addend) -- networks.py:130-131,181."""
    H, W = x.a.H, x.a.W
    near = mode == "nearest"
    ad = None if addend is None else addend.a
    out = Var(ops.resize_nearest(x.a, 2 * H, 2 * W, addend=ad) if near else ops.resize_bilinear(x.a, 2 * H, 2 * W, 0.5, 0.5, addend=ad))

    def bwd():
        d = out.g
        out.g = None
        if d is None:
            return
        if x.req:
            if near:
                x.g = T.resize_nearest_bwd(d, H, W) if x.g is None else T.resize_nearest_bwd(d, H, W, dx=x.g, accumulate=True)
            elif x.g is None:
                x.g = T.resize_bilinear_bwd(d, H, W, 0.5, 0.5)
            else:
                T.resize_bilinear_bwd(d, H, W, 0.5, 0.5, dx=x.g, accumulate=True)
        if addend is not None:
            addend.add_grad(d, owned=True)

    tape.record(bwd)
    return out


def warp(tape: Tape, src: Var, flow_prev: FlowVar, Ho: int, Wo: int, norm_x: float, norm_y: float,
         want_flow_up: bool = True, mode: str = "bilinear") -> Tuple[Var, FlowVar]:
    """up2(flow) -> normalise -> + base grid -> grid_sample(src) in one kernel (networks.py:133-135).  ``mode`` 'nearest'
    (forward(upsample='nearest')): the flow is up-sampled by selection first and the kernel reads it at ratio 1."""
    fh, fw = flow_prev.t.shape[1], flow_prev.t.shape[2]
    rh, rw = fh / Ho, fw / Wo
    near = mode == "nearest"
    if near:
        warped_a, fup_t = ops.flow_warp(src.a, ops.resize_nearest_dense(flow_prev.t, Ho, Wo), Ho, Wo, 1.0, 1.0, norm_x, norm_y,
                                        want_flow_up=True)
    else:
        warped_a, fup_t = ops.flow_warp(src.a, flow_prev.t, Ho, Wo, rh, rw, norm_x, norm_y, want_flow_up=True)
    warped, fup = Var(warped_a), FlowVar(fup_t)

    def bwd():
        d = warped.g
        warped.g = None
        dflow = fup.g        # gradient that reached the upsampled flow through `flow + flow_conv(...)`
        fup.g = None
        if d is not None:
            dsrc = src.grad_or_zero() if src.req else None
            if dflow is None:
                dflow = torch.empty_like(fup.t)
                T.flow_warp_bwd(src.a, fup.t, norm_x, norm_y, d, dsrc, dflow, False)
            else:
                T.flow_warp_bwd(src.a, fup.t, norm_x, norm_y, d, dsrc, dflow, True)
        if dflow is not None:
            flow_prev.add_grad(T.resize_nearest_bwd_dense(dflow, fh, fw) if near else T.resize_bilinear_bwd_dense(dflow, fh, fw, rh, rw),
                               owned=True)

    tape.record(bwd)
    return warped, fup


class FlowConv:
    """flow_conv[i]: Conv2d(768 -> 2, 3x3) as a taps-as-channels 1x1 convolution on the MFMA engine
    + tap-sum gather (forward) / tap scatter + 1x1 data and weight gradients (backward)."""

    def __init__(self, m: nn.Conv2d, name: str):
        self.m, self.name = m, name
        self.Cout, self.cin, self.KH, self.KW = m.weight.shape
        self.pad = self.KH // 2

    def w_taps(self) -> torch.Tensor:
        # [co][c][kh][kw] -> [(kh*KW+kw)*Cout + co][c][1][1]   (layout change of the parameter: plumbing)
        return self.m.weight.data.permute(2, 3, 0, 1).reshape(self.KH * self.KW * self.Cout, self.cin, 1, 1).contiguous()

    def forward(self, tape: Tape, srcs: Sequence[Var], residual: Optional[FlowVar]) -> FlowVar:
        lib = _lib.load()
        acts = [(v.a, 0) for v in srcs]
        wt = self.w_taps()
        # the flow heads stay on the fp32 matrix cores in --fp16 mode too (as in inference, networks.py of this port): a
        # bf16-rounded 768-channel operand moves the flow by ~1e-2 px, which flips floor() cells of every warp downstream
        # (measured at 512x384 ngf=96: tocg gradient cosine vs the oracle 0.88 with bf16 heads)
        mb = T.MMA_BF16[0]
        T.MMA_BF16[0] = False
        try:
            y = T.conv_forward_dev(wt, acts, 1, 0, name=self.name + "[taps 1x1]")
        finally:
            T.MMA_BF16[0] = mb
        a0 = srcs[0].a
        out_t = torch.empty((a0.N, a0.H, a0.W, self.Cout), dtype=torch.float32, device=a0.t.device)
        _lib.check(lib.hrv_tapsum_nhwc_f32(y.t.data_ptr(), y.N, y.H, y.W, self.KH, self.KW, self.pad, self.Cout, y.cstride,
                                           self.m.bias.data.data_ptr(), None if residual is None else residual.t.data_ptr(),
                                           self.Cout, out_t.data_ptr(), self.Cout, _stream()), "hrv_tapsum_nhwc_f32")
        out = FlowVar(out_t)

        def bwd():
            d = out.g
            out.g = None
            if d is None:
                return
            N, H, W = a0.N, a0.H, a0.W
            ntap = self.KH * self.KW * self.Cout
            dy = ops.alloc(N, H, W, ntap, d.device)
            _lib.check(lib.hrv_tapsum_bwd_nhwc_f32(d.data_ptr(), N, H, W, self.KH, self.KW, self.pad, self.Cout, self.Cout,
                                                   dy.t.data_ptr(), dy.cstride, _stream()), "hrv_tapsum_bwd_nhwc_f32")
            # bias gradient: column sums of the dense [npix, 2] tensor viewed as [npix/2, 4]
            n = d.numel()
            s4 = T.colsum(Act(d.view(1, 1, n // 4, 4), 4))
            _acc(tape.grads, self.m.bias, (s4[:2] + s4[2:]) if self.Cout == 2 else s4.view(-1, self.Cout).sum(0))
            Gt = torch.empty_like(wt)
            base = 0
            mb_ = T.MMA_BF16[0]
            T.MMA_BF16[0] = False              # fp32 heads (see forward)
            try:
                for a, _ in acts:
                    T.conv_wgrad(dy, a, 0, base, self.cin, 1, 1, 1, 0, Gt, name=self.name + ".wgrad")
                    base += a.C
                dx = T.conv_dgrad(dy, wt, H, W, 1, 0, name=self.name + ".dgrad")
            finally:
                T.MMA_BF16[0] = mb_
            G = Gt.view(self.KH, self.KW, self.Cout, self.cin).permute(2, 3, 0, 1).contiguous()
            _acc(tape.grads, self.m.weight, G)
            c0 = 0
            for v in srcs:
                v.add_grad(dx.slice(c0, v.a.C), owned=False)
                c0 += v.a.C
            if residual is not None:
                residual.add_grad(d, owned=True)

        tape.record(bwd)
        return out


# ------------------------------------------------------------------------------------ the network
class _RB:
    def __init__(self, rb: nn.Module, name: str):
        self.kind = rb.kind
        if rb.kind == "down":
            self.scale = TConv(rb.scale, 2, 1, name + ".scale")
        elif rb.kind == "same":
            self.scale = TConv(rb.scale, 1, 0, name + ".scale")
        else:
            self.scale = TConv(rb.scale[1], 1, 0, name + ".scale.1")
        self.c1, self.bn1 = TConv(rb.block[0], 1, 1, name + ".block.0"), rb.block[1]
        self.c2, self.bn2 = TConv(rb.block[3], 1, 1, name + ".block.3"), rb.block[4]
        if not isinstance(self.bn1, nn.BatchNorm2d):
            raise NotImplementedError("hr-viton_amd tocg training implements norm_layer=nn.BatchNorm2d")

    def __call__(self, tape: Tape, srcs: Sequence[Var]) -> Var:
        r = conv(tape, self.scale, srcs)
        if self.kind == "up":
            # the 1x1 conv and the bilinear x2 commute (see networks._ResBlockPlan): conv at low resolution
            r = up2(tape, r)
        h = bn_act(tape, self.bn1, conv(tape, self.c1, [r]), ACT_RELU)
        return bn_act(tape, self.bn2, conv(tape, self.c2, [h]), ACT_RELU, residual=r)


class CondTrainPlan:
    def __init__(self, net: nn.Module):
        self.net = net
        self.E1 = [_RB(net.ClothEncoder[i], f"ClothEncoder.{i}") for i in range(5)]
        self.E2 = [_RB(net.PoseEncoder[i], f"PoseEncoder.{i}") for i in range(5)]
        self.mid = _RB(net.conv, "conv")
        self.seg = [_RB(net.SegDecoder[i], f"SegDecoder.{i}") for i in range(5)]
        self.encoder_warp = net.warp_feature == "encoder"
        if net.out_layer_opt == "relu":
            self.out, self.out_conv = _RB(net.out_layer, "out_layer"), None
        else:                                                                       # networks.py:57-61
            self.out, self.out_conv = _RB(net.out_layer[0], "out_layer.0"), TConv(net.out_layer[1], 1, 0, "out_layer.1")
        self.conv1 = [TConv(m, 1, 0, f"conv1.{i}") for i, m in enumerate(net.conv1)]
        self.conv2 = [TConv(m, 1, 0, f"conv2.{i}") for i, m in enumerate(net.conv2)]
        self.flow = [FlowConv(m, f"flow_conv.{i}") for i, m in enumerate(net.flow_conv)]
        self.bott = [TConv(m[0], 1, 1, f"bottleneck.{i}") for i, m in enumerate(net.bottleneck)]

    def forward(self, input1: torch.Tensor, input2: torch.Tensor, upsample: str = "bilinear"):
        tape = Tape()
        um = upsample
        N, _, H, W = input1.shape
        x1, x2 = Var(ops.to_nhwc(input1), req=False), Var(ops.to_nhwc(input2), req=False)
        E1: List[Var] = []
        E2: List[Var] = []
        for i in range(5):
            E1.append(self.E1[i](tape, [x1 if i == 0 else E1[-1]]))
            E2.append(self.E2[i](tape, [x2 if i == 0 else E2[-1]]))
        flows: List[FlowVar] = []
        T1 = T2 = x = None
        for i in range(5):
            e1, e2 = E1[4 - i], E2[4 - i]
            iH, iW = e1.a.H, e1.a.W
            if i == 0:
                T1, T2 = e1, e2
                flows.append(self.flow[0].forward(tape, [T1, T2], None))
                x = self.seg[0](tape, [self.mid(tape, [T2])])
            else:
                T1 = up2(tape, T1, addend=conv(tape, self.conv1[4 - i], [e1]), mode=um)      # networks.py:130
                T2 = up2(tape, T2, addend=conv(tape, self.conv2[4 - i], [e2]), mode=um)      # networks.py:131
                warped, fup = warp(tape, T1, flows[-1], iH, iW, (iW / 2 - 1.0) / 2.0, (iH / 2 - 1.0) / 2.0, mode=um)
                b = conv(tape, self.bott[i - 1], [x], act=ACT_RELU)
                flows.append(self.flow[i].forward(tape, [warped, b], fup))          # networks.py:137
                if not self.encoder_warp:
                    x = self.seg[i](tape, [x, e2, warped])                          # networks.py:141
                else:                                                               # networks.py:143-144
                    warped_e1, _ = warp(tape, e1, flows[-2], iH, iW, (iW / 2 - 1.0) / 2.0, (iH / 2 - 1.0) / 2.0, mode=um)
                    x = self.seg[i](tape, [x, e2, warped_e1])
        warped_in, _ = warp(tape, x1, flows[-1], H, W, (W / 2 - 1.0) / 2.0, (H / 2 - 1.0) / 2.0, mode=um)
        seg = self.out(tape, [x, x2, warped_in])
        if self.out_conv is not None:
            seg = conv(tape, self.out_conv, [seg])
        return tape, flows, seg, warped_in


class _TocgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, input1, input2, upsample, *params):
        tape, flows, seg, warped = plan.forward(input1, input2, upsample)
        ctx.tape, ctx.flows, ctx.seg, ctx.warped, ctx.params = tape, flows, seg, warped, params
        ctx.set_materialize_grads(False)       # an unused output arrives as None (handled below), not as a zero tensor
        return tuple(f.t for f in flows) + (ops.to_nchw(seg.a), ops.to_nchw(warped.a))

    @staticmethod
    def backward(ctx, *d_outs):
        tape, flows, seg, warped = ctx.tape, ctx.flows, ctx.seg, ctx.warped
        nf = len(flows)
        for f, d in zip(flows, d_outs[:nf]):
            if d is not None:
                f.add_grad(d.contiguous(), owned=False)
        if d_outs[nf] is not None:
            seg.add_grad(ops.to_nhwc(d_outs[nf].contiguous()), owned=True)
        if d_outs[nf + 1] is not None:
            warped.add_grad(ops.to_nhwc(d_outs[nf + 1].contiguous()), owned=True)
        grads = tape.backward()
        T.wgrad_join()
        ctx.tape = ctx.flows = ctx.seg = ctx.warped = None
        return (None, None, None, None) + tuple(grads.get(p) for p in ctx.params)


def condition_train_forward(net: nn.Module, input1: torch.Tensor, input2: torch.Tensor, upsample: str = "bilinear"):
    """ConditionGenerator.forward in training mode: (flow_list, x, warped_c, warped_cm)."""
    ops.require_cuda(input1, "ConditionGenerator.forward(input1)")
    ops.require_cuda(input2, "ConditionGenerator.forward(input2)")
    N, _, H, W = input1.shape
    if H % 32 or W % 32:
        raise ValueError(f"input size {H}x{W} must be a multiple of 32 (five stride-2 stages)")
    plan = getattr(net, "_train_plan", None)
    if plan is None:
        plan = net._train_plan = CondTrainPlan(net)
    if not torch.is_grad_enabled():
        _, flows, seg, warped = plan.forward(input1, input2, upsample)
        outs = tuple(f.t for f in flows) + (ops.to_nchw(seg.a), ops.to_nchw(warped.a))
    else:
        outs = _TocgFn.apply(plan, input1, input2, upsample, *list(net.parameters()))
    flow_list, x, w = list(outs[:5]), outs[5], outs[6]
    c = net.input1_nc
    return flow_list, x, w[:, :c - 1], w[:, c - 1:]


# ------------------------------------------------------------------------- tocg discriminator
class CondDPlan:
    """networks.MultiscaleDiscriminator (getIntermFeat=False) -- networks.py:302-346."""

    def __init__(self, msd: nn.Module):
        self.msd = msd
        self.plans = [DiscTrainPlan.from_sequential(getattr(msd, "layer" + str(msd.num_D - 1 - i)),
                                                    f"layer{msd.num_D - 1 - i}") for i in range(msd.num_D)]

    def forward(self, inp: torch.Tensor):
        a = ops.to_nhwc(inp)
        full = a
        if self.msd.Ddownx2:
            a = ops.avgpool3x3s2(a)
        outs, ctxs, inputs = [], [], []
        for i, p in enumerate(self.plans):
            inputs.append(a)
            feats, c = p.forward(a, power_iteration=self.msd.training)   # torch's spectral_norm iterates in train() only
            outs.append(feats[-1])
            ctxs.append((c, len(feats)))
            if i != len(self.plans) - 1:
                a = ops.avgpool3x3s2(a)
        return outs, dict(ctxs=ctxs, inputs=inputs, full=full)

    def backward(self, saved, d_outs: List[Optional[Act]], need_dx: bool, need_w: bool = True):
        grads: Grads = {}
        d_next: Optional[Act] = None
        for i in range(len(self.plans) - 1, -1, -1):
            a = saved["inputs"][i]
            c, nfe = saved["ctxs"][i]
            dfeats = [None] * (nfe - 1) + [d_outs[i]]
            d_a = self.plans[i].backward(c, dfeats, grads, need_dx, need_w=need_w) if d_outs[i] is not None else None
            if need_dx:
                if d_a is None:
                    d_a = Act(torch.zeros_like(a.t), a.C)
                if d_next is not None:
                    T.avgpool3x3s2_bwd(d_next, a.H, a.W, dx=d_a, accumulate=True)
                d_next = d_a
        if need_dx and self.msd.Ddownx2:
            full = saved["full"]
            d_next = T.avgpool3x3s2_bwd(d_next, full.H, full.W)
        return grads, d_next


class _CondDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, inp, *params):
        outs, saved = plan.forward(inp)
        ctx.plan, ctx.saved, ctx.params = plan, saved, params
        ctx.set_materialize_grads(False)
        # loss_G's gradients w.r.t. the discriminator are discarded by optimizer_D.zero_grad() (train_condition.py:284)
        ctx.need_w = not getattr(plan.msd, "_hrv_discard_param_grads", False)
        return tuple(ops.to_nchw(o) for o in outs)

    @staticmethod
    def backward(ctx, *d_outs):
        need_dx = ctx.needs_input_grad[1]
        d_acts = [None if d is None else ops.to_nhwc(d.contiguous()) for d in d_outs]
        grads, d_in = ctx.plan.backward(ctx.saved, d_acts, need_dx, ctx.need_w)
        T.wgrad_join()
        ctx.saved = None
        return (None, ops.to_nchw(d_in) if (need_dx and d_in is not None) else None) + \
            tuple(grads.get(p) for p in ctx.params)


def cond_discriminator_forward(msd: nn.Module, inp: torch.Tensor):
    ops.require_cuda(inp, "MultiscaleDiscriminator.forward")
    plan = getattr(msd, "_train_plan", None)
    if plan is None:
        plan = msd._train_plan = CondDPlan(msd)
    if not torch.is_grad_enabled():
        outs, _ = plan.forward(inp)
        flat = [ops.to_nchw(o) for o in outs]
    else:
        flat = list(_CondDFn.apply(plan, inp, *list(msd.parameters())))
    return [[f] for f in flat]
