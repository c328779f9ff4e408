"""Deterministic model/input recipes shared by oracle/make_golden.py (which applies them to the
REAL reference classes) and the tests (which apply them to the product classes): the same seed and
the same construction order give bit-identical weights, so golden files need not store them.
Test infrastructure only."""
from argparse import Namespace

import torch


def trainstep_opt(cuda: bool):
    return Namespace(cuda=cuda, norm_G="spectralaliasinstance", gen_semantic_nc=7, ngf=8, num_upsampling_layers="most",
                     fine_height=256, fine_width=128, ndf=8, norm_D="spectralinstance", n_layers_D=3, num_D=2,
                     no_ganFeat_loss=False)


def trainstep_build(gen_cls, dis_cls, cuda: bool = False):
    """SPADEGenerator(ngf=8, 'most', 256x128) + MultiscaleDiscriminator(ndf=8), xavier init, weights x25,
    random biases / noise_scale, plus one batch of synthetic inputs and the SPADE noise draws."""
    opt = trainstep_opt(cuda)
    torch.manual_seed(21)
    gen = gen_cls(opt, 9)
    gen.init_weights("xavier", 0.02)
    dis = dis_cls(opt)
    dis.init_weights("xavier", 0.02)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for name, p in list(gen.named_parameters()) + list(dis.named_parameters()):
            if name.endswith("noise_scale"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("weight") or name.endswith("weight_orig"):
                p.mul_(25.0)
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    N, H, W = 2, 256, 128
    x = torch.rand(N, 9, H, W, generator=g) * 2 - 1
    lab = torch.randint(0, 7, (N, 1, H // 16, W // 16), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3)
    seg = torch.zeros(N, 7, H, W).scatter_(1, lab, 1.0)
    real = torch.rand(N, 3, H, W, generator=g) * 2 - 1
    blocks = ["head_0", "G_middle_0", "G_middle_1", "up_0", "up_1", "up_2", "up_3", "up_4"]
    noise = {}
    for j, b in enumerate(blocks):
        h, w = 2 << j, 1 << j
        noise[b] = [torch.randn(N, w, h, 1, generator=g) for _ in range(2 if b == "head_0" else 3)]
    return opt, gen, dis, x, seg, real, noise


def condstep_opt(cuda: bool):
    return Namespace(cuda=cuda, warp_feature="T1", out_layer="relu", semantic_nc=13, output_nc=13)


def condstep_build(tocg_cls, define_D, cuda: bool = False, ngf: int = 8, N: int = 2, H: int = 128, W: int = 96,
                   warp_feature: str = "T1", out_layer: str = "relu"):
    """ConditionGenerator(ngf=8) + define_D(33 ch, Ddownx2, num_D=2) + one synthetic train_condition.py
    batch (default N=2, 128x96, ngf=8 -- the golden recipe; the full-size parity tests pass the timed sizes).  tocg keeps torch's default conv init with randomised BatchNorm affine terms
    and non-trivial running statistics; D uses the reference's weights_init (N(0, 0.02)) scaled x2."""
    opt = condstep_opt(cuda)
    opt.warp_feature, opt.out_layer = warp_feature, out_layer      # (defaults: the golden recipe, networks.py:37-61)
    torch.manual_seed(31)
    tocg = tocg_cls(opt, input1_nc=4, input2_nc=16, output_nc=13, ngf=ngf, norm_layer=torch.nn.BatchNorm2d)
    D = define_D(input_nc=4 + 16 + 13, Ddownx2=True, Ddropout=False, n_layers_D=3, spectral=False, num_D=2)
    g = torch.Generator().manual_seed(91)
    with torch.no_grad():
        for m in tocg.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
        for name, p in tocg.named_parameters():
            if name.startswith("flow_conv") and name.endswith("weight"):
                p.mul_(0.3)   # keep the synthetic flows inside the image so the warps have gradients
        for p in D.parameters():
            p.copy_(0.02 * 2.0 * torch.randn(p.shape, generator=g))
    lab = torch.randint(0, 13, (N, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    parse = torch.zeros(N, 13, H, W).scatter_(1, lab, 1.0)
    agn = torch.randint(0, 13, (N, 1, H // 8, W // 8), generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3)
    def smooth(c):
        # low-frequency images: the warps' coordinate gradients are piecewise constant per source pixel,
        # white noise would make every gradient comparison hinge on floor() flips at pixel borders
        lo = torch.rand(N, c, H // 8, W // 8, generator=g) * 2 - 1
        return torch.nn.functional.interpolate(lo, scale_factor=8, mode="bilinear", align_corners=False)

    batch = {
        "cloth": smooth(3),
        "cloth_mask": (torch.rand(N, 1, H // 4, W // 4, generator=g) > 0.4).float().repeat_interleave(4, 2)
        .repeat_interleave(4, 3),
        "parse_agnostic": torch.zeros(N, 13, H, W).scatter_(1, agn, 1.0),
        "densepose": smooth(3),
        "parse_onehot": lab.float(),          # label indices [N,1,H,W] (cp_dataset.py 'parse_onehot')
        "parse": parse,                       # one-hot 13
        "pcm": (lab == 3).float(),            # parse cloth mask
        "parse_cloth": smooth(3),
    }
    return opt, tocg, D, batch
