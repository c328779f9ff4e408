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
