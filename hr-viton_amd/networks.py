"""MI355X-native mirror of the reference's ``networks.py`` for the hot path.

Same class names, constructor / forward signatures and ``state_dict`` keys as
/root/reference/networks.py (ConditionGenerator :13-159, make_grid :161-168,
ResBlock :171-198, save/load_checkpoint :411-425), so the reference's entry
scripts and ``.pth`` checkpoints are drop-in -- but ``forward`` runs on
hand-written gfx950 kernels through the C ABI (include/hrviton_hip.h):
NHWC fp32 activations, implicit-GEMM MFMA convolutions with eval-BatchNorm /
bias / residual / ReLU fused into the epilogue, torch.cat folded into the conv
gather, and one fused kernel for flow-upsample + normalise + base-grid +
grid_sample.  The nn.Conv2d / nn.BatchNorm2d objects below are parameter
containers only; they are never called.

There is no CPU fallback: CPU tensors raise (ops.require_cuda).
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_RELU, Act, ConvLayer, TapConvLayer


class ResBlock(nn.Module):
    """Parameter container with the reference's key layout (networks.py:171-198):
    scale (down: 3x3 s2 | same: 1x1 | up: [Upsample, 1x1]) and block =
    [conv3x3, norm, ReLU, conv3x3, norm]."""

    def __init__(self, in_nc, out_nc, scale="down", norm_layer=nn.BatchNorm2d):
        super().__init__()
        if scale not in ("up", "down", "same"):
            raise AssertionError("ResBlock scale must be in 'up' 'down' 'same'")
        self.kind = scale
        self.in_nc, self.out_nc = in_nc, out_nc
        with_bias = norm_layer == nn.InstanceNorm2d
        if scale == "down":
            self.scale = nn.Conv2d(in_nc, out_nc, 3, stride=2, padding=1, bias=with_bias)
        elif scale == "same":
            self.scale = nn.Conv2d(in_nc, out_nc, 1, bias=True)
        else:
            self.scale = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear"),
                                       nn.Conv2d(in_nc, out_nc, 1, bias=True))
        body = []
        for k in range(2):
            body += [nn.Conv2d(out_nc, out_nc, 3, stride=1, padding=1, bias=with_bias), norm_layer(out_nc)]
            if k == 0:
                body.append(nn.ReLU(inplace=True))
        self.block = nn.Sequential(*body)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):  # pragma: no cover - containers are driven by the owning network's plan
        raise RuntimeError("hr-viton_amd ResBlock is executed by its owner's HIP plan, not called directly")


def _bn_fold(bn: nn.BatchNorm2d):
    """Eval-mode BatchNorm as per-channel (scale, shift) for the conv epilogue."""
    # host-side weight preparation (once per plan), in double on the CPU
    inv = torch.rsqrt(bn.running_var.detach().cpu().double() + bn.eps)
    g = bn.weight.detach().cpu().double() if bn.affine else torch.ones_like(inv)
    b = bn.bias.detach().cpu().double() if bn.affine else torch.zeros_like(inv)
    scale = g * inv
    shift = b - bn.running_mean.detach().cpu().double() * scale
    return scale.float(), shift.float()


class _ResBlockPlan:
    """HIP execution of one ResBlock in eval mode (3 conv launches [+1 resize])."""

    def __init__(self, rb: ResBlock, src_split: List[int], device, name: str, mixed: bool = False):
        assert isinstance(rb.block[1], nn.BatchNorm2d), "HIP plan implements the BatchNorm2d configuration"
        self.kind = rb.kind
        if rb.kind == "down":
            self.scale = ConvLayer(rb.scale.weight, src_split, device, stride=2, pad=1, name=name + ".scale",
                                   mma_bf16=mixed)
        else:
            conv = rb.scale if rb.kind == "same" else rb.scale[1]
            self.scale = ConvLayer(conv.weight, src_split, device, shift=conv.bias, stride=1, pad=0,
                                   name=name + (".scale" if rb.kind == "same" else ".scale.1"), mma_bf16=mixed)
        s1, b1 = _bn_fold(rb.block[1])
        s2, b2 = _bn_fold(rb.block[4])
        c = rb.out_nc
        self.c1 = ConvLayer(rb.block[0].weight, [c], device, scale=s1, shift=b1, act=ACT_RELU, name=name + ".block.0",
                            mma_bf16=mixed)
        self.c2 = ConvLayer(rb.block[3].weight, [c], device, scale=s2, shift=b2, act=ACT_RELU, name=name + ".block.3",
                            mma_bf16=mixed)

    def __call__(self, srcs: List[Act]) -> Act:
        r = self.scale(srcs)
        if self.kind == "up":
            # nn.Upsample(bilinear x2) and the 1x1 conv commute (both linear, bilinear weights
            # sum to 1 so the bias passes through): convolve at low resolution (4x fewer MACs,
            # no full-resolution Cin-wide intermediate), then upsample the Cout-wide result.
            r = ops.resize_bilinear(r, r.H * 2, r.W * 2, 0.5, 0.5)
        t = self.c1([r])
        return self.c2([t], residual=r)  # relu(r + bn(conv(t)))


def make_grid(N, iH, iW, opt=None):
    """Reference signature (networks.py:161-168).  The HIP warp kernel builds this
    grid in registers; this host version exists for callers that want the tensor."""
    gx = torch.linspace(-1.0, 1.0, iW).view(1, 1, iW, 1).expand(N, iH, -1, -1)
    gy = torch.linspace(-1.0, 1.0, iH).view(1, iH, 1, 1).expand(N, -1, iW, -1)
    grid = torch.cat([gx, gy], 3)
    return grid.cuda() if (opt is not None and getattr(opt, "cuda", False)) else grid


class ConditionGenerator(nn.Module):
    """Try-on condition generator (reference networks.py:13-159), HIP inference path."""

    def __init__(self, opt, input1_nc, input2_nc, output_nc, ngf=64, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.warp_feature = opt.warp_feature
        self.out_layer_opt = opt.out_layer
        if self.warp_feature not in ("T1", "encoder") or self.out_layer_opt not in ("relu", "conv"):
            raise ValueError(f"warp_feature={self.warp_feature!r} / out_layer={self.out_layer_opt!r}: the reference knows "
                             "'T1' | 'encoder' and 'relu' | 'conv' (networks.py:37-61)")
        self.input1_nc, self.input2_nc, self.output_nc, self.ngf = input1_nc, input2_nc, output_nc, ngf
        # --fp16 (test_generator.py:34): inference with bf16 matrix-core operands over fp32 tensors; the default is
        # the fp32 engine (the path held to the 1e-3 parity bar)
        self.mixed_precision = bool(getattr(opt, "fp16", False))
        enc = [ngf, ngf * 2, ngf * 4, ngf * 4, ngf * 4]

        def encoder(cin):
            chans = [cin] + enc
            return nn.Sequential(*[ResBlock(chans[i], chans[i + 1], "down", norm_layer) for i in range(5)])

        self.ClothEncoder = encoder(input1_nc)
        self.PoseEncoder = encoder(input2_nc)
        self.conv = ResBlock(ngf * 4, ngf * 8, "same", norm_layer)
        if self.warp_feature == "T1":       # in_nc = [x, skip connection, warped T1] (networks.py:37-45)
            dec_io = [(ngf * 8, ngf * 4), (ngf * 4 * 2 + ngf * 4, ngf * 4), (ngf * 4 * 2 + ngf * 4, ngf * 2),
                      (ngf * 2 * 2 + ngf * 4, ngf), (ngf * 1 * 2 + ngf * 4, ngf)]
        else:                               # 'encoder': [x, skip connection, warped cloth-encoder feature E1] (networks.py:46-54)
            dec_io = [(ngf * 8, ngf * 4), (ngf * 4 * 3, ngf * 4), (ngf * 4 * 3, ngf * 2), (ngf * 2 * 3, ngf), (ngf * 1 * 3, ngf)]
        self.SegDecoder = nn.Sequential(*[ResBlock(i, o, "up", norm_layer) for i, o in dec_io])
        if self.out_layer_opt == "relu":
            self.out_layer = ResBlock(ngf + input1_nc + input2_nc, output_nc, "same", norm_layer)
        else:                               # 'conv' (networks.py:57-61): a ResBlock to ngf channels, then a 1x1 to the logits
            self.out_layer = nn.Sequential(ResBlock(ngf + input1_nc + input2_nc, ngf, "same", norm_layer),
                                           nn.Conv2d(ngf, output_nc, kernel_size=1, bias=True))
        lat = [ngf, ngf * 2, ngf * 4, ngf * 4]
        self.conv1 = nn.Sequential(*[nn.Conv2d(c, ngf * 4, 1, bias=True) for c in lat])
        self.conv2 = nn.Sequential(*[nn.Conv2d(c, ngf * 4, 1, bias=True) for c in lat])
        self.flow_conv = nn.ModuleList([nn.Conv2d(ngf * 8, 2, 3, 1, 1, bias=True) for _ in range(5)])
        self.bottleneck = nn.Sequential(*[nn.Sequential(nn.Conv2d(c, ngf * 4, 3, 1, 1, bias=True), nn.ReLU())
                                          for c in (ngf * 4, ngf * 4, ngf * 2, ngf)])
        self._plan = None
        self._plan_key = None

    def normalize(self, x):
        return x

    # ------------------------------------------------------------------ plan
    def _state_version(self):
        return tuple(t._version for t in list(self.parameters()) + list(self.buffers()))

    def _build_plan(self, device):
        ngf, c4 = self.ngf, self.ngf * 4
        mx = self.mixed_precision
        P = {}
        P["E1"] = [_ResBlockPlan(self.ClothEncoder[i], [self.ClothEncoder[i].in_nc], device, f"ClothEncoder.{i}", mx)
                   for i in range(5)]
        P["E2"] = [_ResBlockPlan(self.PoseEncoder[i], [self.PoseEncoder[i].in_nc], device, f"PoseEncoder.{i}", mx)
                   for i in range(5)]
        P["conv"] = _ResBlockPlan(self.conv, [c4], device, "conv", mx)
        enc = [ngf, ngf * 2, ngf * 4, ngf * 4, ngf * 4]
        dec_out = [c4, c4, ngf * 2, ngf, ngf]
        seg = [_ResBlockPlan(self.SegDecoder[0], [ngf * 8], device, "SegDecoder.0", mx)]
        for i in range(1, 5):
            # cat([x, E2[4-i], warped_T1]) -- networks.py:141; 'encoder': the third source is the warped E1[4-i] (:143-144)
            third = c4 if self.warp_feature == "T1" else enc[4 - i]
            seg.append(_ResBlockPlan(self.SegDecoder[i], [dec_out[i - 1], enc[4 - i], third], device, f"SegDecoder.{i}", mx))
        P["seg"] = seg
        if self.out_layer_opt == "relu":
            P["out"] = _ResBlockPlan(self.out_layer, [ngf, self.input2_nc, self.input1_nc], device, "out_layer", mx)
        else:
            P["out"] = _ResBlockPlan(self.out_layer[0], [ngf, self.input2_nc, self.input1_nc], device, "out_layer.0", mx)
            m = self.out_layer[1]
            P["out_conv"] = ConvLayer(m.weight, [m.in_channels], device, shift=m.bias, pad=0, name="out_layer.1", mma_bf16=mx)
        P["conv1"] = [ConvLayer(m.weight, [m.in_channels], device, shift=m.bias, pad=0, name=f"conv1.{i}", mma_bf16=mx)
                      for i, m in enumerate(self.conv1)]
        P["conv2"] = [ConvLayer(m.weight, [m.in_channels], device, shift=m.bias, pad=0, name=f"conv2.{i}", mma_bf16=mx)
                      for i, m in enumerate(self.conv2)]
        # 768 -> 2 channels: taps-as-channels 1x1 on the MFMA engine + tap-sum gather.  The flow head stays on the
        # fp32 engine in mixed mode: sub-pixel flow offsets are what the 1e-3 warp parity hangs on, and it is 2 % of
        # the MACs
        P["flow"] = [TapConvLayer(m.weight, [c4, c4], device, bias=m.bias, name=f"flow_conv.{i}")
                     for i, m in enumerate(self.flow_conv)]
        P["bott"] = [ConvLayer(m[0].weight, [m[0].in_channels], device, shift=m[0].bias, pad=1, act=ACT_RELU,
                               name=f"bottleneck.{i}", mma_bf16=mx) for i, m in enumerate(self.bottleneck)]
        return P

    def _get_plan(self, device):
        key = (str(device), self._state_version(), ops.weights_epoch(self.parameters()), self.mixed_precision)
        if self._plan is None or self._plan_key != key:
            self._plan = self._build_plan(device)
            self._plan_key = key
        return self._plan

    # --------------------------------------------------------------- forward
    def forward(self, *args, upsample="bilinear"):
        """Reference signature ``forward(opt, input1, input2, upsample='bilinear')``
        (networks.py:98); the 2-argument form the reference's training scripts use
        (train_generator.py:215 ...) is accepted too."""
        if len(args) == 3:
            _, input1, input2 = args
        elif len(args) == 2:
            input1, input2 = args
        else:
            raise TypeError("forward(opt, input1, input2) or forward(input1, input2)")
        if upsample not in ("bilinear", "nearest"):
            raise ValueError(f"upsample={upsample!r}: the reference's scripts know 'bilinear' | 'nearest' (test_generator.py:63)")
        if self.training:
            # batch-statistics BatchNorm + tape-recorded backward (train_condition.py:116,158)
            from .cond_train import condition_train_forward
            if not next(self.parameters()).is_cuda:
                raise RuntimeError("hr-viton_amd ConditionGenerator: move the module to the GPU (.cuda()) first; "
                                   "there is no CPU path")
            return condition_train_forward(self, input1, input2, upsample)
        return self._forward_eval(input1, input2, upsample)

    @torch.no_grad()
    def _forward_eval(self, input1: torch.Tensor, input2: torch.Tensor, upsample: str = "bilinear"):
        ops.require_cuda(input1, "ConditionGenerator.forward(input1)")
        ops.require_cuda(input2, "ConditionGenerator.forward(input2)")
        N, _, H, W = input1.shape
        if H % 32 or W % 32:
            raise ValueError(f"input size {H}x{W} must be a multiple of 32 (five stride-2 stages)")
        P = self._get_plan(input1.device)
        x1 = ops.to_nhwc(input1)
        x2 = ops.to_nhwc(input2)
        E1: List[Act] = []
        E2: List[Act] = []
        # the two encoders are independent chains: at 4 x 1024x768 PoseEncoder runs on the side stream next to ClothEncoder (70.9 ->
        # 72.8 images/s); at the 256x192 of the frozen condition generator inside train_generator.py the fork / join costs more than the
        # 6 .. 768-tile launches gain (67.0 -> 67.4 ms per iteration, three alternations) -- HRV_TOCG_SIDE=0 / 1: never / always
        from . import train_ops as _T
        side = os.environ.get("HRV_TOCG_SIDE", "auto")
        with _T.side_region(x2, on=(side == "1" or (side != "0" and N * H * W >= (1 << 20)))):
            for i in range(5):
                E2.append(P["E2"][i]([x2 if i == 0 else E2[-1]]))
        for i in range(5):
            E1.append(P["E1"][i]([x1 if i == 0 else E1[-1]]))
        _T.wgrad_join(input1.device)
        flow_list: List[torch.Tensor] = []
        T1 = T2 = x = None
        for i in range(5):
            e1, e2 = E1[4 - i], E2[4 - i]
            iH, iW = e1.H, e1.W
            if i == 0:
                T1, T2 = e1, e2
                fl = Act(torch.empty((N, iH, iW, 2), dtype=torch.float32, device=input1.device), 2)
                P["flow"][0]([T1, T2], out=fl)
                flow_list.append(fl.t)
                x = P["seg"][0]([P["conv"]([T2])])
            else:
                # T = up2(T) + conv1x1(E)  (networks.py:130-131): 1x1 conv, then resize with fused addend
                a1 = P["conv1"][4 - i]([e1])
                a2 = P["conv2"][4 - i]([e2])
                near = upsample == "nearest"
                if not near:
                    T1 = ops.resize_bilinear(T1, iH, iW, 0.5, 0.5, addend=a1)
                    T2 = ops.resize_bilinear(T2, iH, iW, 0.5, 0.5, addend=a2)
                    flow_prev, fr = flow_list[-1], 0.5
                else:       # upsample='nearest' (networks.py:130-133): the flow is up-sampled by selection first, the warp kernel then
                    #         reads it at ratio 1 (its bilinear taps collapse to the pixel itself)
                    T1 = ops.resize_nearest(T1, iH, iW, addend=a1)
                    T2 = ops.resize_nearest(T2, iH, iW, addend=a2)
                    flow_prev, fr = ops.resize_nearest_dense(flow_list[-1], iH, iW), 1.0
                # flow upsample + flow_norm + make_grid + grid_sample in one kernel (networks.py:133-135)
                warped, fup = ops.flow_warp(T1, flow_prev, iH, iW, fr, fr,
                                            (iW / 2 - 1.0) / 2.0, (iH / 2 - 1.0) / 2.0)
                b = P["bott"][i - 1]([x])
                fl = Act(torch.empty((N, iH, iW, 2), dtype=torch.float32, device=input1.device), 2)
                # flow = up(flow) + flow_conv(cat([warped_T1, bottleneck(x)]))  (networks.py:137)
                P["flow"][i]([warped, b], out=fl, residual=Act(fup, 2))
                flow_list.append(fl.t)
                if self.warp_feature == "T1":
                    x = P["seg"][i]([x, e2, warped])
                else:       # the decoder reads the cloth-encoder feature itself, warped by the same (upsampled previous) flow
                    warped_e1, _ = ops.flow_warp(e1, flow_prev, iH, iW, fr, fr, (iW / 2 - 1.0) / 2.0, (iH / 2 - 1.0) / 2.0,
                                                 want_flow_up=False)
                    x = P["seg"][i]([x, e2, warped_e1])
        if upsample == "nearest":                                                        # networks.py:150
            warped_in, _ = ops.flow_warp(x1, ops.resize_nearest_dense(flow_list[-1], H, W), H, W, 1.0, 1.0, (W / 2 - 1.0) / 2.0,
                                         (H / 2 - 1.0) / 2.0, want_flow_up=False)
        else:
            warped_in, _ = ops.flow_warp(x1, flow_list[-1], H, W, 0.5, 0.5, (W / 2 - 1.0) / 2.0, (H / 2 - 1.0) / 2.0,
                                         want_flow_up=False)
        seg = P["out"]([x, x2, warped_in])
        if self.out_layer_opt == "conv":
            seg = P["out_conv"]([seg])
        seg_nchw = ops.to_nchw(seg)
        warped_nchw = ops.to_nchw(warped_in)
        c = self.input1_nc
        return flow_list, seg_nchw, warped_nchw[:, :c - 1], warped_nchw[:, c - 1:]


# ---------------------------------------------------------------------------------------------
# Discriminator of the condition generator + its losses (networks.py:201-408, 427-453)
# ---------------------------------------------------------------------------------------------
from .vgg import VGGLoss, Vgg19  # noqa: E402,F401  (networks.py:201-251 live in vgg.py)


class GANLoss(nn.Module):
    """networks.GANLoss (networks.py:258-299): LSGAN = MSE against a constant 1 / 0 map of the LAST
    tensor of every scale, summed over the scales.  One fused value+gradient kernel per scale."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor):
        super().__init__()
        if not use_lsgan:
            raise NotImplementedError("hr-viton_amd networks.GANLoss implements use_lsgan=True (train_condition.py:124-126)")
        self.real_label, self.fake_label = target_real_label, target_fake_label
        self.real_label_var = self.fake_label_var = None
        self.Tensor = tensor
        from .losses import MSELoss
        self.loss = MSELoss()

    def get_target_tensor(self, input, target_is_real):
        attr = "real_label_var" if target_is_real else "fake_label_var"
        cur = getattr(self, attr)
        if cur is None or cur.numel() != input.numel() or cur.device != input.device:
            cur = torch.full(tuple(input.shape), self.real_label if target_is_real else self.fake_label,
                             dtype=torch.float32, device=input.device)
            setattr(self, attr, cur)
        return cur

    def __call__(self, input, target_is_real):
        if isinstance(input[0], list):
            total = 0
            for scale in input:
                pred = scale[-1]
                total = total + self.loss(pred, self.get_target_tensor(pred, target_is_real))
            return total
        pred = input[-1]
        return self.loss(pred, self.get_target_tensor(pred, target_is_real))


class NLayerDiscriminator(nn.Module):
    """Parameter container with the reference's module layout (networks.py:348-408)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, getIntermFeat=False,
                 Ddropout=False, spectral=False):
        super().__init__()
        self.getIntermFeat, self.n_layers = getIntermFeat, n_layers
        sn = nn.utils.spectral_norm if spectral else (lambda m: m)
        kw, padw = 4, 2
        groups = [[nn.Conv2d(input_nc, ndf, kw, 2, padw), nn.LeakyReLU(0.2, True)]]
        nf = ndf
        for _ in range(1, n_layers):
            prev, nf = nf, min(nf * 2, 512)
            g = [sn(nn.Conv2d(prev, nf, kw, 2, padw)), norm_layer(nf), nn.LeakyReLU(0.2, True)]
            if Ddropout:
                g.append(nn.Dropout(0.5))
            groups.append(g)
        prev, nf = nf, min(nf * 2, 512)
        groups.append([nn.Conv2d(prev, nf, kw, 1, padw), norm_layer(nf), nn.LeakyReLU(0.2, True)])
        groups.append([nn.Conv2d(nf, 1, kw, 1, padw)])
        if use_sigmoid:
            groups.append([nn.Sigmoid()])
        if getIntermFeat:
            for n, g in enumerate(groups):
                setattr(self, "model" + str(n), nn.Sequential(*g))
        else:
            self.model = nn.Sequential(*[m for g in groups for m in g])

    def forward(self, input):  # pragma: no cover
        raise RuntimeError("hr-viton_amd NLayerDiscriminator is executed by MultiscaleDiscriminator's HIP plan")


class MultiscaleDiscriminator(nn.Module):
    """networks.MultiscaleDiscriminator (networks.py:302-346) on the HIP training kernels."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, num_D=3,
                 getIntermFeat=False, Ddownx2=False, Ddropout=False, spectral=False):
        super().__init__()
        self.num_D, self.n_layers, self.getIntermFeat, self.Ddownx2 = num_D, n_layers, getIntermFeat, Ddownx2
        for i in range(num_D):
            netD = NLayerDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid, getIntermFeat, Ddropout,
                                       spectral=spectral)
            if getIntermFeat:
                for j in range(n_layers + 2):
                    setattr(self, "scale" + str(i) + "_layer" + str(j), getattr(netD, "model" + str(j)))
            else:
                setattr(self, "layer" + str(i), netD.model)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)
        self._hip_ok = (not getIntermFeat) and (not use_sigmoid)

    def forward(self, input):
        if not self._hip_ok:
            raise NotImplementedError("hr-viton_amd tocg discriminator: the getIntermFeat / use_sigmoid variants are "
                                      "not on the HIP path (train_condition.py:484 never builds them)")
        from .cond_train import cond_discriminator_forward
        return cond_discriminator_forward(self, input)


def weights_init(m):
    """networks.py:427-433."""
    name = m.__class__.__name__
    if name.find("Conv2d") != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif name.find("BatchNorm2d") != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def get_norm_layer(norm_type="instance"):
    import functools
    if norm_type == "batch":
        return functools.partial(nn.BatchNorm2d, affine=True)
    if norm_type == "instance":
        return functools.partial(nn.InstanceNorm2d, affine=False)
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def define_D(input_nc, ndf=64, n_layers_D=3, norm="instance", use_sigmoid=False, num_D=2, getIntermFeat=False,
             gpu_ids=[], Ddownx2=False, Ddropout=False, spectral=False):
    """networks.py:445-453."""
    netD = MultiscaleDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm_type=norm), use_sigmoid, num_D,
                                   getIntermFeat, Ddownx2, Ddropout, spectral=spectral)
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        netD.cuda()
    netD.apply(weights_init)
    return netD


def save_checkpoint(model, save_path, opt=None):
    """networks.py:411-417 -- torch.save of the CPU state_dict (same file format)."""
    d = os.path.dirname(save_path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    sd = model.state_dict()
    out = type(sd)((k, v.detach().cpu()) for k, v in sd.items())
    if hasattr(sd, "_metadata"):
        out._metadata = sd._metadata      # spectral-norm version entries travel with the file
    torch.save(out, save_path)


def load_checkpoint(model, checkpoint_path, opt=None):
    """networks.py:419-425 (strict=False).  A missing file raises FileNotFoundError
    (the reference's bare ``raise`` surfaces as 'No active exception to reraise')."""
    if not os.path.exists(checkpoint_path):
        print("no checkpoint")
        raise FileNotFoundError(checkpoint_path)
    model.load_state_dict(torch.load(checkpoint_path, map_location="cpu"), strict=False)
    if opt is not None and getattr(opt, "cuda", False):
        model.cuda()
