"""MI355X-native mirror of the reference's ``network_generator.py`` for the hot path.

Same class names, constructor / forward signatures and ``state_dict`` keys
(including torch's spectral-norm ``weight_orig / weight_u / weight_v`` entries and
their ``_metadata`` versions) as /root/reference/network_generator.py:
BaseNetwork :9-50, SPADENorm :75-122, SPADEResBlock :125-173, SPADEGenerator
:176-245, NLayerDiscriminator :250-288, MultiscaleDiscriminator :291-316.  The
nn.Conv2d objects are parameter containers; ``forward`` executes a HIP plan over
the C ABI (include/hrviton_hip.h):

  * every conv on the fp32 MFMA implicit-GEMM engine, spectral-norm 1/sigma, bias,
    residual add, LeakyReLU / tanh fused into its epilogue;
  * SPADE: instance-norm statistics kernel (noise folded in), then ONE conv for
    conv_gamma||conv_beta whose epilogue applies IN(x+noise)*(1+gamma)+beta and
    the LeakyReLU -- gamma/beta never reach HBM;
  * nearest x2 upsample + torch.cat fused away: the producing conv stores each
    result to its 2x2 block of the next block's input buffer, the stem conv
    writes its 16 channels next to it; seg / x nearest-resizes are folded into
    the conv gather (power-of-two strides).

No CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init
from torch.nn.utils import spectral_norm

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, Act, ConvLayer, SpadeModulate


class BaseNetwork(nn.Module):
    """network_generator.py:9-50."""

    def __init__(self):
        super().__init__()

    def print_network(self):
        n = sum(p.numel() for p in self.parameters())
        print("Network [{}] was created. Total number of parameters: {:.1f} million. "
              "To see the architecture, do print(network).".format(self.__class__.__name__, n / 1000000))

    def init_weights(self, init_type="normal", gain=0.02):
        table = {
            "normal": lambda w: init.normal_(w, 0.0, gain),
            "xavier": lambda w: init.xavier_normal_(w, gain=gain),
            "xavier_uniform": lambda w: init.xavier_uniform_(w, gain=1.0),
            "kaiming": lambda w: init.kaiming_normal_(w, a=0, mode="fan_in"),
            "orthogonal": lambda w: init.orthogonal_(w, gain=gain),
        }

        def visit(m):
            cname = m.__class__.__name__
            if "BatchNorm2d" in cname:
                if getattr(m, "weight", None) is not None:
                    init.normal_(m.weight.data, 1.0, gain)
                if getattr(m, "bias", None) is not None:
                    init.constant_(m.bias.data, 0.0)
            elif ("Conv" in cname or "Linear" in cname) and hasattr(m, "weight"):
                if init_type == "none":
                    m.reset_parameters()
                elif init_type in table:
                    table[init_type](m.weight.data)
                else:
                    raise NotImplementedError("initialization method '{}' is not implemented".format(init_type))
                if getattr(m, "bias", None) is not None:
                    init.constant_(m.bias.data, 0.0)

        self.apply(visit)

    def forward(self, *inputs):
        pass


def _sn_sigma(conv: nn.Module) -> float:
    """Eval-mode spectral norm sigma = u . (W v) (torch SpectralNorm.compute_weight without
    the power iteration); 1.0 for a plain conv."""
    if hasattr(conv, "weight_orig"):
        w = conv.weight_orig.detach().double().cpu()
        u = conv.weight_u.detach().double().cpu()
        v = conv.weight_v.detach().double().cpu()
        return float(torch.dot(u, torch.mv(w.reshape(w.shape[0], -1), v)))
    return 1.0


def _raw_weight(conv: nn.Module) -> torch.Tensor:
    return conv.weight_orig if hasattr(conv, "weight_orig") else conv.weight


class SPADENorm(nn.Module):
    """Parameter container, network_generator.py:75-99."""

    def __init__(self, opt, norm_type, norm_nc, label_nc):
        super().__init__()
        self.param_opt = opt
        self.noise_scale = nn.Parameter(torch.zeros(norm_nc))
        assert norm_type.startswith("alias")
        kind = norm_type[len("alias"):]
        if kind != "instance":
            raise ValueError("hr-viton_amd SPADENorm implements 'aliasinstance' (the reference default "
                             "norm_G='spectralaliasinstance'); got '{}'".format(norm_type))
        self.param_free_norm = nn.InstanceNorm2d(norm_nc, affine=False)  # parameter-free; executed by the HIP plan
        nhidden, ks = 128, 3
        self.conv_shared = nn.Sequential(nn.Conv2d(label_nc, nhidden, kernel_size=ks, padding=ks // 2), nn.ReLU())
        self.conv_gamma = nn.Conv2d(nhidden, norm_nc, kernel_size=ks, padding=ks // 2)
        self.conv_beta = nn.Conv2d(nhidden, norm_nc, kernel_size=ks, padding=ks // 2)
        # layout hint for the fused optimizer's flat buffers (optim.Adam._setup): the backward computes [dW_gamma ; dW_beta]
        # (and the two bias gradients) as ONE matrix -- with the mates adjacent it is written in place, no slice copies
        self.conv_beta.weight._hrv_flat_after = self.conv_gamma.weight
        self.conv_beta.bias._hrv_flat_after = self.conv_gamma.bias


class SPADEResBlock(nn.Module):
    """Parameter container, network_generator.py:125-156."""

    def __init__(self, opt, input_nc, output_nc, use_mask_norm=True):
        super().__init__()
        if use_mask_norm:
            raise NotImplementedError("MaskNorm blocks are never constructed by the reference generator "
                                      "(network_generator.py:188-198) and are out of scope")
        self.param_opt = opt
        self.learned_shortcut = input_nc != output_nc
        middle_nc = min(input_nc, output_nc)
        self.input_nc, self.middle_nc, self.output_nc = input_nc, middle_nc, output_nc
        self.conv_0 = nn.Conv2d(input_nc, middle_nc, kernel_size=3, padding=1)
        self.conv_1 = nn.Conv2d(middle_nc, output_nc, kernel_size=3, padding=1)
        if self.learned_shortcut:
            self.conv_s = nn.Conv2d(input_nc, output_nc, kernel_size=1, bias=False)
        subnorm = opt.norm_G
        if subnorm.startswith("spectral"):
            subnorm = subnorm[len("spectral"):]
            self.conv_0 = spectral_norm(self.conv_0)
            self.conv_1 = spectral_norm(self.conv_1)
            if self.learned_shortcut:
                self.conv_s = spectral_norm(self.conv_s)
        nc = opt.gen_semantic_nc
        self.norm_0 = SPADENorm(opt, subnorm, input_nc, nc)
        self.norm_1 = SPADENorm(opt, subnorm, middle_nc, nc)
        if self.learned_shortcut:
            self.norm_s = SPADENorm(opt, subnorm, input_nc, nc)
        self.relu = nn.LeakyReLU(0.2)


class _SpadePlan:
    """One SPADENorm on the HIP path: stats -> conv_shared(+ReLU) -> fused gamma||beta+modulate."""

    def __init__(self, norm: SPADENorm, device, act: int, name: str, bf16: bool = False):
        cs = norm.conv_shared[0]
        self.label_nc = cs.in_channels
        self.cs = cs      # conv_shared parameters: the owning block batches its norms' 3x3s into one 1x1 (see _BlockPlan)
        self.norm_params = (norm.conv_gamma.weight, norm.conv_gamma.bias, norm.conv_beta.weight, norm.conv_beta.bias)
        self._gf = None
        self.mod = SpadeModulate(norm.conv_gamma.weight, norm.conv_gamma.bias, norm.conv_beta.weight,
                                 norm.conv_beta.bias, norm.noise_scale, device, act, name + ".conv_gamma|beta", bf16=bf16)

    def fused_ok(self, x: Act, seg: Act) -> bool:
        """conv_shared can run inside the gamma|beta kernel (csrc/spade_fused.hip: bf16 plan, >= 2 tiles per CU)"""
        from . import train_ops as T
        m = self.mod
        return bool(m.bf16 and m.Cp == m.Creal and seg.bf16 and seg.cstride == 8 and seg.coff == 0 and self.label_nc <= 8 and
                    x.cstride % 4 == 0 and T.spade_fused_ok(m.Creal, self.cs.out_channels, self.label_nc, x.N, x.H, x.W))

    def fused_ok_shape(self, N: int, H: int, W: int, seg: Act) -> bool:
        from . import train_ops as T
        m = self.mod
        return bool(m.bf16 and m.Cp == m.Creal and seg.bf16 and seg.cstride == 8 and seg.coff == 0 and self.label_nc <= 8 and
                    T.spade_fused_ok(m.Creal, self.cs.out_channels, self.label_nc, N, H, W))

    def __call__(self, x: Act, actv: Optional[Act], z: Optional[torch.Tensor], fused=None, stats=None) -> Act:
        zz = z if (z is not None and self.mod.has_noise) else None
        mean, rstd = stats if stats is not None else ops.instnorm_stats(x, zz, self.mod.ns if zz is not None else None)
        if fused is not None:
            # SPADENorm end to end in one launch: actv = ReLU(conv_shared(label map)) never reaches HBM
            from . import train_ops as T
            seg, shift = fused
            dev = x.t.device
            if getattr(self, "_gf", None) is None:      # frozen weights: packed once per plan
                f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()      # noqa: E731
                n = self.norm_params
                self._gf = (T.spade_fused_pack(f32(self.cs.weight), f32(self.cs.bias), f32(n[0]), f32(n[2])), f32(n[1]), f32(n[3]))
            pk, bg, bb = self._gf
            out = ops.alloc(x.N, x.H, x.W, self.mod.Creal, dev, bf16=True)
            T.spade_fused_forward(seg, shift, x, mean, rstd, zz, self.mod.ns if zz is not None else None, pk, bg, bb, self.mod.conv.act,
                                  self.mod.conv.slope, out, None, None, self.mod.conv.name.replace("conv_gamma|beta", "conv_shared+gamma|beta"))
            return out
        return self.mod(actv, x, mean, rstd, zz)


def _shared_1x1(norms, device, name: str, bf16: bool) -> ConvLayer:
    """The conv_shared 3x3s (label_nc -> 128, + ReLU) of a block's norms as ONE 1x1 convolution over the
    tap-expanded label map (ops.tap_expand: 9 taps x 8 padded channels = 72 dense inputs): the label map is
    read once per block instead of 9 x (2 or 3) times, and no K-tile is 7/8 padding."""
    ws, bs = [], []
    for n_ in norms:
        w = n_.cs.weight.detach().to("cpu", torch.float32)          # [128, C, 3, 3]
        co, c, kh, kw = w.shape
        cp = (c + 7) // 8 * 8 if bf16 else (c + 3) // 4 * 4
        wt = torch.zeros(co, kh * kw, cp)
        wt[:, :, :c] = w.permute(0, 2, 3, 1).reshape(co, kh * kw, c)   # tap-major, channel-minor (matches tap_expand)
        ws.append(wt.reshape(co, kh * kw * cp, 1, 1))
        bs.append(n_.cs.bias.detach().to("cpu", torch.float32))
    w1 = torch.cat(ws, 0)
    return ConvLayer(w1, [w1.shape[1]], device, shift=torch.cat(bs, 0), pad=0, act=ACT_RELU,
                     name=name + ".conv_shared[x%d as 1x1 over taps]" % len(norms), bf16=bf16)


class _BlockPlan:
    def __init__(self, blk: SPADEResBlock, device, name: str, bf16: bool = False):
        self.learned = blk.learned_shortcut
        self.bf16 = bf16
        self.n0 = _SpadePlan(blk.norm_0, device, ACT_LRELU, name + ".norm_0", bf16)
        self.n1 = _SpadePlan(blk.norm_1, device, ACT_LRELU, name + ".norm_1", bf16)
        s0, s1 = _sn_sigma(blk.conv_0), _sn_sigma(blk.conv_1)
        self.c0 = ConvLayer(_raw_weight(blk.conv_0), [blk.input_nc], device,
                            scale=torch.full((blk.middle_nc,), 1.0 / s0), shift=blk.conv_0.bias, pad=1,
                            name=name + ".conv_0", bf16=bf16, out_f32=True)   # dx feeds norm_1's InstanceNorm
        self.c1_scale = torch.full((blk.output_nc,), 1.0 / s1)
        self.c1_w, self.c1_b = _raw_weight(blk.conv_1), blk.conv_1.bias
        # bf16 serving: conv_0 / conv_1 of the fine levels run on the training forward's kernels (train_ops.conv_forward_fast)
        self._fast = None
        if bf16:
            dv = lambda t: None if t is None else t.detach().to(device=device, dtype=torch.float32).contiguous()    # noqa: E731
            from . import train_ops as T
            # (tok: unique per plan object -- an id() or a storage address can be recycled by a later plan and would then return
            #  this plan's packed weights from train_ops' frozen-pack cache)
            self._fast = dict(w0=dv(_raw_weight(blk.conv_0)), b0=dv(blk.conv_0.bias), s0=1.0 / s0,
                              w1=dv(self.c1_w), b1=dv(self.c1_b), s1=1.0 / s1, tok=next(T._PACK_TOKENS))
        self.device, self.name, self.blk = device, name, blk
        self._c1 = {}
        if self.learned:
            self.ns_ = _SpadePlan(blk.norm_s, device, ACT_NONE, name + ".norm_s", bf16)
            ss = _sn_sigma(blk.conv_s)
            self.cs = ConvLayer(_raw_weight(blk.conv_s), [blk.input_nc], device,
                                scale=torch.full((blk.output_nc,), 1.0 / ss), pad=0, name=name + ".conv_s", bf16=bf16,
                                out_f32=True)
        self.norms = ([self.ns_] if self.learned else []) + [self.n0, self.n1]
        self.shared = _shared_1x1(self.norms, device, name, bf16)
        self.nh = blk.norm_0.conv_shared[0].out_channels

    def conv1(self, act: int) -> ConvLayer:
        if act not in self._c1:
            # the block output is the next block's InstanceNorm input (fp32) -- except the last block,
            # whose LeakyReLU'd output only feeds conv_img (bf16)
            self._c1[act] = ConvLayer(self.c1_w, [self.blk.middle_nc], self.device, scale=self.c1_scale,
                                      shift=self.c1_b, pad=1, act=act, name=self.name + ".conv_1", bf16=self.bf16,
                                      out_f32=(act == ACT_NONE))
        return self._c1[act]

    def _conv0(self, h0: Act) -> Act:
        if self._fast is not None:
            from . import train_ops as T
            f = self._fast
            r = T.conv_forward_fast(f["w0"], h0, 1, f["s0"], f["b0"], None, ACT_NONE, 0.2, None, False, self.name + ".conv_0",
                                    ("serve", f["tok"], 0))
            if r is not None:
                return r
        return self.c0([h0])

    def _conv1(self, h1: Act, x_s: Act, out: Optional[Act], out_up: int, out_act: int) -> Act:
        if self._fast is not None and out_up == 0 and type(x_s) is Act:
            from . import train_ops as T
            f = self._fast
            r = T.conv_forward_fast(f["w1"], h1, 1, f["s1"], f["b1"], x_s, out_act, 0.2, out, out_act != ACT_NONE, self.name + ".conv_1",
                                    ("serve", f["tok"], 1))
            if r is not None:
                return r
        return self.conv1(out_act)([h1], out=out, residual=x_s, out_up=out_up)

    def reads_upsampled_input(self, N: int, H: int, W: int, C_in: int, seg: Act) -> bool:
        """This block can read its input as ops.ActUp = cat(up2(previous output), stem) without the 4-fold fp32 copy ever being
        written (bf16 serving: a learned-shortcut block whose norms all run on csrc/spade_fused.hip -- the one-pass statistics of
        norm_s / norm_0 and that kernel are the only readers of x; a zero noise_scale rides through as z * 0).  HRV_XUP=0: off."""
        if os.environ.get("HRV_XUP", "1") == "0" or os.environ.get("HRV_SPADE_FUSED", "1") == "0" or not (self.bf16 and self.learned):
            return False
        if H % 2 or W % 2 or (C_in - 16) % 32 != 0 or self.n0.mod.Creal != C_in or self.ns_.mod.Creal != C_in:
            return False
        return all(n_.fused_ok_shape(N, H, W, seg) for n_ in self.norms)

    def __call__(self, x: Act, seg: Act, seg_shift: int, zs: Sequence[Optional[torch.Tensor]], out: Optional[Act],
                 out_up: int, out_act: int) -> Act:
        """x_s + conv_1(lrelu(norm_1(conv_0(lrelu(norm_0(x)))))) -- network_generator.py:163-173.
        ``zs``: noise draws in the reference's call order (norm_s, norm_0, norm_1)."""
        zi = iter(zs)
        if isinstance(x, ops.ActUp):
            # (the caller checked reads_upsampled_input) norm_s and norm_0 normalise the same x: one statistics pass over lo / hi
            f = (seg, seg_shift)
            z_s, z_0 = next(zi), next(zi)
            st_s, st_0 = ops.instnorm_stats2(x, z_s, self.ns_.mod.ns, z_0, self.n0.mod.ns)
            x_s = self.cs([self.ns_(x, None, z_s, f, st_s)])
            dx = self._conv0(self.n0(x, None, z_0, f, st_0))
            h1 = self.n1(dx, None, next(zi), f)
            return self._conv1(h1, x_s, out, out_up, out_act)
        if os.environ.get("HRV_SPADE_FUSED", "1") != "0" and all(n_.fused_ok(x, seg) for n_ in self.norms):
            # every norm of the block computes its conv_shared inside its gamma|beta kernel: no conv_shared launch, no actv tensor
            f = (seg, seg_shift)
            x_s = self.cs([self.ns_(x, None, next(zi), f)]) if self.learned else x
            dx = self._conv0(self.n0(x, None, next(zi), f))
            h1 = self.n1(dx, None, next(zi), f)
            return self._conv1(h1, x_s, out, out_up, out_act)
        # every norm of the block sees the same nearest-resized label map (network_generator.py:112-113)
        actv_all = self.shared([ops.tap_expand(seg, seg_shift, 3)])
        actv = [actv_all.slice(i * self.nh, self.nh) for i in range(len(self.norms))]
        ai = iter(actv)
        if self.learned:
            x_s = self.cs([self.ns_(x, next(ai), next(zi))])
        else:
            x_s = x
        dx = self._conv0(self.n0(x, next(ai), next(zi)))
        h1 = self.n1(dx, next(ai), next(zi))
        return self._conv1(h1, x_s, out, out_up, out_act)


class SPADEGenerator(BaseNetwork):
    """Try-on image generator (network_generator.py:176-245), HIP inference path."""

    BLOCKS = ["head_0", "G_middle_0", "G_middle_1", "up_0", "up_1", "up_2", "up_3", "up_4"]

    def __init__(self, opt, input_nc):
        super().__init__()
        self.num_upsampling_layers = opt.num_upsampling_layers
        self.param_opt = opt
        self.sh, self.sw = self.compute_latent_vector_size(opt)
        self.input_nc = input_nc
        nf = opt.ngf
        self.conv_0 = nn.Conv2d(input_nc, nf * 16, kernel_size=3, padding=1)
        for i in range(1, 8):
            self.add_module("conv_{}".format(i), nn.Conv2d(input_nc, 16, kernel_size=3, padding=1))
        self.head_0 = SPADEResBlock(opt, nf * 16, nf * 16, use_mask_norm=False)
        self.G_middle_0 = SPADEResBlock(opt, nf * 16 + 16, nf * 16, use_mask_norm=False)
        self.G_middle_1 = SPADEResBlock(opt, nf * 16 + 16, nf * 16, use_mask_norm=False)
        self.up_0 = SPADEResBlock(opt, nf * 16 + 16, nf * 8, use_mask_norm=False)
        self.up_1 = SPADEResBlock(opt, nf * 8 + 16, nf * 4, use_mask_norm=False)
        self.up_2 = SPADEResBlock(opt, nf * 4 + 16, nf * 2, use_mask_norm=False)
        self.up_3 = SPADEResBlock(opt, nf * 2 + 16, nf * 1, use_mask_norm=False)
        if self.num_upsampling_layers == "most":
            self.up_4 = SPADEResBlock(opt, nf * 1 + 16, nf // 2, use_mask_norm=False)
            nf = nf // 2
        self.conv_img = nn.Conv2d(nf, 3, kernel_size=3, padding=1)
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.relu = nn.LeakyReLU(0.2)
        self.tanh = nn.Tanh()
        self._plan = None
        self._plan_key = None

    def compute_latent_vector_size(self, opt):
        table = {"normal": 5, "more": 6, "most": 7}
        if self.num_upsampling_layers not in table:
            raise ValueError("opt.num_upsampling_layers '{}' is not recognized".format(self.num_upsampling_layers))
        n = table[self.num_upsampling_layers]
        return opt.fine_height // 2 ** n, opt.fine_width // 2 ** n

    # ------------------------------------------------------------------ plan
    def _blocks(self) -> List[str]:
        names = list(self.BLOCKS[:7])
        if self.num_upsampling_layers == "most":
            names.append("up_4")
        return names

    def _use_bf16(self) -> bool:
        """The reference's ``--fp16`` (apex O1) switch selects the bf16-storage / fp32-accumulate engine
        (no loss scaling needed); default fp32."""
        return bool(getattr(self.param_opt, "fp16", False))

    def _build_plan(self, device):
        bf = self._use_bf16()
        P = {"bf16": bf, "blocks": [_BlockPlan(getattr(self, n), device, n, bf) for n in self._blocks()]}
        P["stem"] = [ConvLayer(getattr(self, f"conv_{i}").weight, [self.input_nc], device,
                               shift=getattr(self, f"conv_{i}").bias, pad=1, name=f"conv_{i}", bf16=bf, out_f32=True)
                     for i in range(len(P["blocks"]))]
        P["img"] = ConvLayer(self.conv_img.weight, [self.conv_img.in_channels], device, shift=self.conv_img.bias, pad=1,
                             act=ACT_TANH, name="conv_img", bf16=bf, out_f32=True)
        if bf:
            dv = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()      # noqa: E731
            P["stem_fast"] = [dict(w=dv(getattr(self, f"conv_{i}").weight), b=dv(getattr(self, f"conv_{i}").bias))
                              for i in range(len(P["blocks"]))]
            from . import train_ops as T
            P["tok"] = next(T._PACK_TOKENS)
            P["img_fast"] = dict(w=self.conv_img.weight.detach().to(device=device, dtype=torch.float32).contiguous(),
                                 b=self.conv_img.bias.detach().to(device=device, dtype=torch.float32).contiguous())
        return P

    def _get_plan(self, device):
        key = (str(device), tuple(t._version for t in list(self.parameters()) + list(self.buffers())), ops.weights_epoch(self.parameters()),
               self._use_bf16())
        if self._plan is None or self._plan_key != key:
            old = self._plan
            if old is not None:          # the replaced plan's packed streams in train_ops' frozen-pack cache are dead entries now
                from . import train_ops as T
                toks = [old["tok"]] if "tok" in old else []
                toks += [b._fast["tok"] for b in old.get("blocks", []) if getattr(b, "_fast", None)]
                if toks:
                    T.evict_serving_packs(toks)
            self._plan = self._build_plan(device)
            self._plan_key = key
        return self._plan

    # --------------------------------------------------------------- forward
    def forward(self, x, seg, noise: Optional[Dict[str, Sequence[torch.Tensor]]] = None):
        """``forward(x, seg)`` as the reference (network_generator.py:221).  ``noise`` optionally
        injects the per-SPADENorm N(0,1) draws ([b,w,h,1] each, in the reference's call order per
        block) so a run can be compared against a recorded reference run; by default they are drawn
        with torch.randn on the device exactly where the reference draws them (:104-107)."""
        if self.num_upsampling_layers == "normal":
            raise ValueError("num_upsampling_layers='normal' is broken in the reference itself (shape mismatch at "
                             "G_middle_1, network_generator.py:228-230)")
        if self.training:
            # one autograd.Function whose backward is the hand-written HIP plan (gen_train.py)
            from .gen_train import generator_train_forward
            return generator_train_forward(self, x, seg, noise)
        with torch.no_grad():
            return self._forward_eval(x, seg, noise)

    def _forward_eval(self, x, seg, noise):
        ops.require_cuda(x, "SPADEGenerator.forward(x)")
        if not isinstance(seg, Act):
            ops.require_cuda(seg, "SPADEGenerator.forward(seg)")
        N, _, H, W = x.shape
        names = self._blocks()
        nb = len(names)
        top = nb - 1                        # blocks run at (sh,sw) * 2^j, j = 0..top
        if (self.sh << top, self.sw << top) != (H, W):
            raise ValueError(f"input {H}x{W} does not match fine_height/fine_width and num_upsampling_layers="
                             f"'{self.num_upsampling_layers}' (needs {self.sh << top}x{self.sw << top}; 'most' "
                             "needs H, W multiples of 128 -- see network_generator.py:207-218)")
        P = self._get_plan(x.device)
        bf = P["bf16"]
        xin = ops.to_nhwc(x, bf16=bf)       # [N,H,W,12|16] (9 real channels)
        if isinstance(seg, Act):
            sg = seg if seg.bf16 == bf else ops.to_nhwc(ops.to_nchw(seg), bf16=bf)
        else:
            sg = ops.to_nhwc(seg, bf16=bf)  # [N,H,W,8]  (7 real channels)
        dev = x.device

        def draws(name, blk, h, w):
            if noise is not None:
                return [z.to(dev).contiguous() for z in noise[name]]
            k = 3 if blk.learned else 2
            return [torch.randn(N, w, h, 1, device=dev) for _ in range(k)]

        cur: Optional[Act] = None
        noise_ok = noise is None or all(all(z is not None for z in v) for v in noise.values())
        for j, name in enumerate(names):
            blk = P["blocks"][j]
            h, w = self.sh << j, self.sw << j
            shift = top - j                 # log2 of the nearest down-sampling of x / seg at this scale
            cin = getattr(self, name).input_nc
            if j == 0:
                cur = P["stem"][0]([(xin, -shift, ACT_NONE)])
            else:
                # cur already holds up(prev) in channels [0, cin-16); the stem conv fills the rest
                hi = cur.hi if isinstance(cur, ops.ActUp) else cur.slice(cin - 16, 16)
                done = None
                if bf and shift == 0 and "stem_fast" in P:      # conv_7: 9 -> 16 channels over every pixel, memory-bound (thin_conv.hip)
                    from . import train_ops as T
                    sf = P["stem_fast"][j]
                    done = T.conv_forward_fast(sf["w"], xin, 1, 1.0, sf["b"], None, ACT_NONE, 0.2, hi, False, f"conv_{j}", ("serve", P["tok"], 10 + j))
                if done is None:
                    P["stem"][j]([(xin, -shift, ACT_NONE)], out=hi)
            assert cur.C == cin and (cur.H, cur.W) == (h, w)
            last = j == nb - 1
            if last:
                # the generator ends with conv_img(leaky_relu(x)): fuse that LeakyReLU here
                cur = blk(cur, sg, shift, draws(name, blk, h, w), None, 0, ACT_LRELU)
            else:
                nxt_c = getattr(self, names[j + 1]).input_nc
                if noise_ok and P["blocks"][j + 1].reads_upsampled_input(N, h * 2, w * 2, nxt_c, sg):
                    # the next block reads cat(up2(this output), its stem) in place (ops.ActUp): no 4-fold fp32 copy
                    lo = ops.alloc(N, h, w, nxt_c - 16, dev)
                    blk(cur, sg, shift, draws(name, blk, h, w), lo, 0, ACT_NONE)
                    cur = ops.ActUp(lo, ops.alloc(N, h * 2, w * 2, 16, dev))
                else:
                    nxt = ops.alloc(N, h * 2, w * 2, nxt_c, dev)     # residual stream / InstanceNorm input: fp32
                    blk(cur, sg, shift, draws(name, blk, h, w), nxt.slice(0, nxt_c - 16), 1, ACT_NONE)
                    cur = nxt
        img = None
        if bf and "img_fast" in P:
            from . import train_ops as T
            fi = P["img_fast"]
            img = T.conv_forward_fast(fi["w"], cur, 1, 1.0, fi["b"], None, ACT_TANH, 0.2, None, False, "conv_img", ("serve", P["tok"], 2))
        if img is None:
            img = P["img"]([cur])
        return ops.to_nchw(img)


# ----------------------------------------------------------------------------
# PatchGAN discriminator of the generator (network_generator.py:250-316)
# ----------------------------------------------------------------------------

def get_nonspade_norm_layer(norm_type="instance"):
    """network_generator.py:401-433 (spectral + instance / none)."""

    def add_norm_layer(layer):
        sub = norm_type
        if norm_type.startswith("spectral"):
            layer = spectral_norm(layer)
            sub = norm_type[len("spectral"):]
        if sub == "none" or len(sub) == 0:
            return layer
        if getattr(layer, "bias", None) is not None:
            delattr(layer, "bias")
            layer.register_parameter("bias", None)
        if sub == "instance":
            norm_layer = nn.InstanceNorm2d(layer.out_channels, affine=False)
        elif sub == "batch":
            norm_layer = nn.BatchNorm2d(layer.out_channels, affine=True)
        else:
            raise ValueError("normalization layer %s is not recognized" % sub)
        return nn.Sequential(layer, norm_layer)

    return add_norm_layer


class NLayerDiscriminator(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        self.no_ganFeat_loss = opt.no_ganFeat_loss
        nf = opt.ndf
        kw = 4
        pw = int(np.ceil((kw - 1.0) / 2))
        norm_layer = get_nonspade_norm_layer(opt.norm_D)
        input_nc = opt.gen_semantic_nc + 3
        seq = [[nn.Conv2d(input_nc, nf, kernel_size=kw, stride=2, padding=pw), nn.LeakyReLU(0.2, False)]]
        for _ in range(1, opt.n_layers_D):
            nf_prev, nf = nf, min(nf * 2, 512)
            seq += [[norm_layer(nn.Conv2d(nf_prev, nf, kernel_size=kw, stride=2, padding=pw)), nn.LeakyReLU(0.2, False)]]
        seq += [[nn.Conv2d(nf, 1, kernel_size=kw, stride=1, padding=pw)]]
        for n, mods in enumerate(seq):
            self.add_module("model" + str(n), nn.Sequential(*mods))
        self.n_models = len(seq)
        self._plan = None
        self._plan_key = None

    def _build_plan(self, device):
        plan = []
        for n in range(self.n_models):
            m = getattr(self, "model" + str(n))
            first = m[0]
            if isinstance(first, nn.Sequential):          # [SN conv (no bias), InstanceNorm] + LeakyReLU
                conv = first[0]
                if not isinstance(first[1], nn.InstanceNorm2d):
                    raise NotImplementedError("HIP discriminator implements norm_D='spectralinstance'")
                layer = ConvLayer(_raw_weight(conv), [conv.in_channels], device,
                                  scale=torch.full((conv.out_channels,), 1.0 / _sn_sigma(conv)),
                                  stride=conv.stride[0], pad=conv.padding[0], name=f"model{n}")
                plan.append(("in_lrelu", layer))
            else:
                conv = first
                act = ACT_LRELU if len(m) > 1 else ACT_NONE
                layer = ConvLayer(_raw_weight(conv), [conv.in_channels], device,
                                  scale=torch.full((conv.out_channels,), 1.0 / _sn_sigma(conv)), shift=conv.bias,
                                  stride=conv.stride[0], pad=conv.padding[0], act=act, name=f"model{n}")
                plan.append(("plain", layer))
        return plan

    def _get_plan(self, device):
        key = (str(device), tuple(t._version for t in list(self.parameters()) + list(self.buffers())), ops.weights_epoch(self.parameters()))
        if self._plan is None or self._plan_key != key:
            self._plan = self._build_plan(device)
            self._plan_key = key
        return self._plan

    def forward_act(self, a: Act) -> List[Act]:
        feats = []
        for kind, layer in self._get_plan(a.t.device):
            a = layer([a])
            if kind == "in_lrelu":
                mean, rstd = ops.instnorm_stats(a)
                a = ops.instnorm_apply(a, mean, rstd, ACT_LRELU, 0.2, out=a)
            feats.append(a)
        return feats

    def forward(self, input):
        if self.training:
            # one scale of the multi-scale training plan (network_generator.py:278-289 called on its own): same kernels, same
            # autograd Function; the holder only stands in for the MultiscaleDiscriminator that normally owns the plan
            from .gen_train import discriminator_train_forward
            solo = self.__dict__.get("_solo_scale")
            if solo is None:
                solo = self.__dict__["_solo_scale"] = _SoloScale(self)
            feats = discriminator_train_forward(solo, input)[0]
            return feats if not self.no_ganFeat_loss else feats[-1]
        with torch.no_grad():
            feats = [ops.to_nchw(f) for f in self.forward_act(ops.to_nhwc(input))]
        return feats if not self.no_ganFeat_loss else feats[-1]


class _SoloScale:
    """What gen_train.discriminator_train_forward asks of a MultiscaleDiscriminator, for ONE NLayerDiscriminator used on its own
    in training mode: its scales (one), its parameters, the feature-list switch.  Not an nn.Module (the discriminator must not
    become its own grandchild)."""

    def __init__(self, D):
        self._D = D

    # (forwarded, not snapshotted: a flag flipped after the first call must be the one the training forward sees)
    no_ganFeat_loss = property(lambda self: self._D.no_ganFeat_loss)

    def children(self):
        return iter([self._D])

    def parameters(self):
        return self._D.parameters()


class MultiscaleDiscriminator(BaseNetwork):
    def __init__(self, opt):
        super().__init__()
        self.no_ganFeat_loss = opt.no_ganFeat_loss
        for i in range(opt.num_D):
            self.add_module("discriminator_%d" % i, NLayerDiscriminator(opt))

    def downsample(self, input):
        raise RuntimeError("executed inside forward() by hrv_avgpool3x3s2_nhwc_f32")

    def forward_pair(self, parse7, fake, real):
        """Training mode: (pred_fake, pred_real) of discriminator(cat((cat((parse, fake), 1), cat((parse, real), 1)), 0)) --
        train_generator.py:283-295 -- with the input assembled NHWC by one kernel per half instead of three torch.cat's
        and a layout pass; ``parse7``: the label-map activation (ops.Act), ``fake`` / ``real``: NCHW images."""
        assert self.training, "forward_pair is the training-step form"
        from .gen_train import discriminator_train_forward_pair
        return discriminator_train_forward_pair(self, parse7, fake.contiguous(), real.contiguous())

    def forward(self, input, split: bool = False):
        """List (scales) of lists (layers) of NCHW tensors -- network_generator.py:306-316.  ``split=True`` (an
        extension for the [fake ; real] batches of train_generator.py:283-295) returns (pred_fake, pred_real)."""
        if self.training:
            from .gen_train import discriminator_train_forward
            return discriminator_train_forward(self, input, split)
        assert not split, "split is a training-mode option"
        ops.require_cuda(input, "MultiscaleDiscriminator.forward")
        result = []
        with torch.no_grad():
            a = ops.to_nhwc(input)
            ds = list(self.children())
            for k, D in enumerate(ds):
                feats = [ops.to_nchw(f) for f in D.forward_act(a)]
                result.append(feats if not self.no_ganFeat_loss else [feats[-1]])
                if k + 1 < len(ds):
                    a = ops.avgpool3x3s2(a)
        return result
