#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (test infrastructure).

Runs only in the build container, where ``/root/reference`` is mounted
read-only: it imports ``networks.py`` / ``network_generator.py`` from there
(CPU, fp32, bytecode writing disabled, ``torchvision`` stubbed because
networks.py:5 imports it and it is not installed), runs the reference modules
on seeded synthetic inputs and writes small fixtures to ``tests/golden/``.
The fixtures travel to the GPU box; the reference does not.

    python oracle/make_golden.py            # rewrites tests/golden/*.pt
"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace

sys.dont_write_bytecode = True
REF = os.environ.get("HRV_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float  # numpy>=1.24 removed the alias the reference uses
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    sys.path.insert(0, REF)
    import networks as ref_networks  # noqa
    import network_generator as ref_gen  # noqa
    sys.path.pop(0)
    return ref_networks, ref_gen


def _randomize_bn(model, g):
    import torch
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(1.0 + 0.2 * torch.randn(m.weight.shape, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.bias.shape, generator=g))


def main():
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ref_networks, ref_gen = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    g = torch.Generator().manual_seed(1234)

    # ---------------- primitives ----------------
    prim = {}
    x = torch.randn(2, 5, 7, 6, generator=g)
    grid = (torch.rand(2, 9, 8, 2, generator=g) * 2.6 - 1.3)
    prim["gs_in"], prim["gs_grid"] = x, grid
    prim["gs_out"] = F.grid_sample(x, grid, padding_mode="border")
    prim["bil_in"] = torch.randn(2, 3, 5, 4, generator=g)
    prim["bil_x2"] = F.interpolate(prim["bil_in"], scale_factor=2, mode="bilinear")
    prim["bil_size"] = F.interpolate(prim["bil_in"], size=(13, 9), mode="bilinear")
    prim["near_size"] = F.interpolate(prim["bil_in"], size=(10, 12), mode="nearest")
    opt_cpu = Namespace(cuda=False)
    prim["grid_7x5"] = ref_networks.make_grid(2, 7, 5, opt_cpu)
    prim["grid_24x18"] = ref_networks.make_grid(1, 24, 18, opt_cpu)
    prim["in_in"] = torch.randn(2, 4, 9, 7, generator=g) * 3 + 1
    prim["in_out"] = nn.InstanceNorm2d(4, affine=False)(prim["in_in"])
    torch.save(prim, os.path.join(OUT, "primitives.pt"))

    # ---------------- tocg ----------------
    torch.manual_seed(7)
    opt = Namespace(cuda=False, warp_feature="T1", out_layer="relu")
    NGF = 8
    tocg = ref_networks.ConditionGenerator(opt, input1_nc=4, input2_nc=16, output_nc=13, ngf=NGF,
                                           norm_layer=nn.BatchNorm2d)
    with torch.no_grad():
        _randomize_bn(tocg, g)
        # default conv init leaves flows tiny; scale flow_conv so warps are exercised
        for fc in tocg.flow_conv:
            fc.weight.mul_(4.0)
    tocg.eval()
    N, H, W = 2, 96, 64
    input1 = torch.cat([torch.rand(N, 3, H, W, generator=g) * 2 - 1,
                        (torch.rand(N, 1, H, W, generator=g) > 0.5).float()], 1)
    lab = torch.randint(0, 13, (N, 1, H, W), generator=g)
    input2 = torch.cat([torch.zeros(N, 13, H, W).scatter_(1, lab, 1.0),
                        torch.rand(N, 3, H, W, generator=g) * 2 - 1], 1)
    with torch.no_grad():
        flow_list, seg, wc, wcm = tocg(opt, input1, input2)
    torch.save({"ngf": NGF, "state_dict": {k: v.clone() for k, v in tocg.state_dict().items()},
                "input1": input1, "input2": input2, "flow_list": flow_list, "seg": seg,
                "warped_c": wc, "warped_cm": wcm}, os.path.join(OUT, "tocg_ngf8_96x64.pt"))

    # ---------------- SPADE generator ----------------
    torch.manual_seed(11)
    gopt = Namespace(cuda=False, norm_G="spectralaliasinstance", gen_semantic_nc=7, ngf=2,
                     num_upsampling_layers="most", fine_height=256, fine_width=128,
                     ndf=8, norm_D="spectralinstance", n_layers_D=3, num_D=2, no_ganFeat_loss=False)
    gen = ref_gen.SPADEGenerator(gopt, 9)
    gen.init_weights("xavier", 0.02)
    with torch.no_grad():
        for name, p in gen.named_parameters():
            if name.endswith("noise_scale"):
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            elif name.endswith("weight") or name.endswith("weight_orig"):
                p.mul_(30.0)  # xavier(gain .02) weights are ~1e-3: scale so activations are O(1)
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    gen.eval()
    N = 2
    x9 = torch.rand(N, 9, 256, 128, generator=g) * 2 - 1
    lab7 = torch.randint(0, 7, (N, 1, 256, 128), generator=g)
    # blob-ify labels a little: nearest-upsampled coarse map
    lab7 = F.interpolate(lab7[:, :, ::8, ::8].float(), size=(256, 128), mode="nearest").long()
    seg7 = torch.zeros(N, 7, 256, 128).scatter_(1, lab7, 1.0)
    draws = []
    real_randn = torch.randn

    def rec_randn(*a, **k):
        z = real_randn(*a, **k)
        draws.append(z.clone())
        return z

    torch.randn = rec_randn
    try:
        with torch.no_grad():
            gout = gen(x9, seg7)
    finally:
        torch.randn = real_randn
    # map the draws (call order) to blocks: norm_s, norm_0, norm_1 per block; head_0 has no norm_s
    blocks = ["head_0", "G_middle_0", "G_middle_1", "up_0", "up_1", "up_2", "up_3", "up_4"]
    noise, k = {}, 0
    for b in blocks:
        n = 2 if b == "head_0" else 3
        noise[b] = draws[k:k + n]
        k += n
    assert k == len(draws) == 23
    gsd = gen.state_dict()
    torch.save({"opt": vars(gopt), "state_dict": {k: v.clone() for k, v in gsd.items()},
                "metadata": dict(gsd._metadata), "x": x9, "seg": seg7, "noise": noise, "out": gout},
               os.path.join(OUT, "gen_ngf2_256x128.pt"))

    # ---------------- generator's discriminator ----------------
    torch.manual_seed(13)
    D = ref_gen.MultiscaleDiscriminator(gopt)
    D.init_weights("xavier", 0.02)
    with torch.no_grad():
        for name, p in D.named_parameters():
            if name.endswith("weight") or name.endswith("weight_orig"):
                p.mul_(20.0)
    D.eval()
    dinp = torch.cat([seg7[:, :, ::2, ::2], torch.rand(N, 3, 128, 64, generator=g) * 2 - 1], 1)
    with torch.no_grad():
        dout = D(dinp)
    torch.save({"state_dict": {k: v.clone() for k, v in D.state_dict().items()}, "input": dinp, "out": dout},
               os.path.join(OUT, "gend_ndf8_128x64.pt"))

    # ---------------- one generator + discriminator training step of the REAL reference --------
    # (train_generator.py:279-360 re-composed on CPU without the VGG term: torchvision weights are
    # unavailable).  Only summaries are stored (losses, per-parameter gradient max/sum, updated u/v
    # norms): they pin the oracle's training-mode restatement (spectral-norm power iteration, hinge,
    # feature matching); full gradients are compared against the oracle live in the GPU tests.
    from oracle.recipes import trainstep_build
    topt, tgen, tD, tx, tseg, treal, tnoise = trainstep_build(ref_gen.SPADEGenerator, ref_gen.MultiscaleDiscriminator)
    tgen.train()
    tD.train()
    # feed the recipe's noise draws to the reference through its own torch.randn call sites
    feed = [z for b in blocks for z in tnoise[b]]
    feed_it = iter(feed)

    def fed_randn(*a, **k):
        z = next(feed_it)
        assert tuple(z.shape) == tuple(a), (z.shape, a)
        return z.clone()

    torch.randn = fed_randn
    try:
        tout = tgen(tx, tseg)
    finally:
        torch.randn = real_randn
    crit = ref_gen.GANLoss("hinge", tensor=torch.FloatTensor)
    pred = tD(torch.cat((torch.cat((tseg, tout), 1), torch.cat((tseg, treal), 1)), 0))
    pf = [[t[: t.size(0) // 2] for t in p] for p in pred]
    pr = [[t[t.size(0) // 2:] for t in p] for p in pred]
    l_gan = crit(pf, True, for_discriminator=False)
    l_feat = torch.zeros(1)
    for i in range(2):
        for j in range(len(pf[i]) - 1):
            l_feat = l_feat + nn.L1Loss()(pf[i][j], pr[i][j].detach()) * 10.0 / 2
    (l_gan + l_feat).mean().backward()
    summ = {n_: (p.grad.abs().max().item(), p.grad.sum().item(), p.grad.abs().sum().item())
            for n_, p in list(tgen.named_parameters()) + [("D." + a, b_) for a, b_ in tD.named_parameters()]
            if p.grad is not None}
    torch.save({"recipe": "oracle.recipes.trainstep_build", "out": tout.detach(), "l_gan": l_gan.detach(),
                "l_feat": l_feat.detach(),
                "grad_summary": summ,
                "u_after": {"up_0.conv_0": tgen.up_0.conv_0.weight_u.clone(),
                            "D.discriminator_0.model1.0.0": tD.discriminator_0.model1[0][0].weight_u.clone()},
                "sample_grads": {"conv_img.weight": tgen.conv_img.weight.grad.clone(),
                                 "up_4.norm_1.noise_scale": tgen.up_4.norm_1.noise_scale.grad.clone(),
                                 "head_0.conv_0.bias": tgen.head_0.conv_0.bias.grad.clone(),
                                 "D.discriminator_1.model0.0.weight": tD.discriminator_1.model0[0].weight.grad.clone()}},
               os.path.join(OUT, "trainstep_ngf8_256x128.pt"))

    # ---------------- parse glue inputs (restated blur: parity unpinned) ----------------
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


def tocg_variants():
    """ConditionGenerator(warp_feature='encoder', out_layer='conv') of the REAL reference -- the decoder reads the warped cloth-encoder
    feature instead of the warped T1 (networks.py:46-54,142-144) and ends in ResBlock + Conv2d 1x1 (networks.py:57-61) -- eval
    forward on seeded inputs.  Its own generator and seeds: the other fixtures regenerate byte-identically."""
    import torch
    import torch.nn as nn
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ref_networks, _ = _import_reference()
    g = torch.Generator().manual_seed(4321)
    torch.manual_seed(17)
    opt = Namespace(cuda=False, warp_feature="encoder", out_layer="conv")
    NGF = 8
    tocg = ref_networks.ConditionGenerator(opt, input1_nc=4, input2_nc=16, output_nc=13, ngf=NGF, norm_layer=nn.BatchNorm2d)
    with torch.no_grad():
        _randomize_bn(tocg, g)
        for fc in tocg.flow_conv:
            fc.weight.mul_(4.0)
    tocg.eval()
    N, H, W = 1, 96, 64
    input1 = torch.cat([torch.rand(N, 3, H, W, generator=g) * 2 - 1, (torch.rand(N, 1, H, W, generator=g) > 0.5).float()], 1)
    lab = torch.randint(0, 13, (N, 1, H, W), generator=g)
    input2 = torch.cat([torch.zeros(N, 13, H, W).scatter_(1, lab, 1.0), torch.rand(N, 3, H, W, generator=g) * 2 - 1], 1)
    with torch.no_grad():
        flow_list, seg, wc, wcm = tocg(opt, input1, input2)
        nf, ns, nwc, nwcm = tocg(opt, input1, input2, upsample="nearest")      # networks.py:98,130-133,150: the forward's third mode
    name = "tocg_encoder_conv_ngf8_96x64.pt"
    torch.save({"ngf": NGF, "warp_feature": "encoder", "out_layer": "conv",
                "state_dict": {k: v.clone() for k, v in tocg.state_dict().items()},
                "input1": input1, "input2": input2, "flow_list": flow_list, "seg": seg, "warped_c": wc, "warped_cm": wcm,
                "nearest": {"flow_list": nf, "seg": ns, "warped_c": nwc, "warped_cm": nwcm}},
               os.path.join(OUT, name))
    print(name, os.path.getsize(os.path.join(OUT, name)) // 1024, "KiB")


def condstep():
    """One tocg + D training iteration of the REAL reference (train_condition.py:136-286 re-composed on
    CPU from the reference's own classes; config --Ddownx2 --lasttvonly --interflowloss, VGG terms
    dropped: torchvision weights are unavailable).  Summaries only."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ref_networks, _ = _import_reference()
    from oracle.recipes import condstep_build
    opt, tocg, D, b = condstep_build(ref_networks.ConditionGenerator, ref_networks.define_D)
    tocg.train()
    D.train()
    crit = ref_networks.GANLoss(use_lsgan=True, tensor=torch.Tensor)
    l1 = nn.L1Loss()
    opt_G = torch.optim.Adam(tocg.parameters(), lr=0.0002, betas=(0.5, 0.999))
    opt_D = torch.optim.Adam(D.parameters(), lr=0.0002, betas=(0.5, 0.999))
    c, cm = b["cloth"], b["cloth_mask"]
    in1, in2 = torch.cat([c, cm], 1), torch.cat([b["parse_agnostic"], b["densepose"]], 1)
    flows, seg, w_c, w_cm = tocg(opt, in1, in2)
    seg_raw = seg
    mask = torch.ones_like(seg.detach())
    mask[:, 3:4, :, :] = w_cm                       # clothmask_composition == 'warp_grad'
    seg = seg * mask
    loss_l1 = l1(w_cm, b["pcm"])
    last = flows[-1]
    loss_tv = torch.abs(last[:, 1:] - last[:, :-1]).mean() + torch.abs(last[:, :, 1:] - last[:, :, :-1]).mean()
    N, _, iH, iW = c.size()

    def rm_overlap(so, wcm):   # train_condition.py:38-43
        return wcm - (torch.cat([so[:, 1:3], so[:, 5:]], dim=1)).sum(dim=1, keepdim=True) * wcm

    for i in range(len(flows) - 1):
        fl = flows[i]
        _, fH, fW, _ = fl.size()
        grid = ref_networks.make_grid(N, iH, iW, opt)
        fl = F.interpolate(fl.permute(0, 3, 1, 2), size=c.shape[2:], mode="bilinear").permute(0, 2, 3, 1)
        fn = torch.cat([fl[:, :, :, 0:1] / ((fW - 1.0) / 2.0), fl[:, :, :, 1:2] / ((fH - 1.0) / 2.0)], 3)
        wcm_i = F.grid_sample(cm, fn + grid, padding_mode="border")
        wcm_i = rm_overlap(F.softmax(seg, dim=1), wcm_i)
        loss_l1 = loss_l1 + l1(wcm_i, b["pcm"]) / (2 ** (4 - i))
    tgt = b["parse_onehot"].transpose(0, 1)[0].long()
    ce = F.cross_entropy(seg.transpose(1, 2).transpose(2, 3).contiguous().view(-1, 13), tgt.view(-1), ignore_index=250)
    soft = torch.softmax(seg, 1)
    pred = D(torch.cat((in1.detach(), in2.detach(), soft), dim=1))
    g_gan = crit(pred, True)
    pred_f = D(torch.cat((in1.detach(), in2.detach(), soft.detach()), dim=1))
    pred_r = D(torch.cat((in1.detach(), in2.detach(), b["parse"]), dim=1))
    d_fake, d_real = crit(pred_f, False), crit(pred_r, True)
    loss_G = (10 * loss_l1 + 2 * loss_tv) + (ce * 10 + g_gan * 1)
    loss_D = d_fake + d_real
    opt_G.zero_grad()
    loss_G.backward()
    g_summ = {n_: (p.grad.abs().max().item(), p.grad.sum().item(), p.grad.abs().sum().item())
              for n_, p in tocg.named_parameters() if p.grad is not None}
    samples = {k: dict(tocg.named_parameters())[k].grad.clone() for k in
               ("ClothEncoder.0.scale.weight", "flow_conv.2.weight", "SegDecoder.4.block.1.weight",
                "out_layer.block.4.bias", "conv1.1.bias")}
    opt_G.step()
    opt_D.zero_grad()
    loss_D.backward()
    d_summ = {n_: (p.grad.abs().max().item(), p.grad.sum().item(), p.grad.abs().sum().item())
              for n_, p in D.named_parameters() if p.grad is not None}
    samples_d = {k: dict(D.named_parameters())[k].grad.clone() for k in ("layer0.0.weight", "layer1.11.bias")}
    opt_D.step()
    sd = tocg.state_dict()
    torch.save({"recipe": "oracle.recipes.condstep_build",
                "losses": {"loss_G": loss_G.item(), "loss_D": loss_D.item(), "l1": loss_l1.item(), "tv": loss_tv.item(),
                           "ce": ce.item(), "g_gan": g_gan.item(), "d_fake": d_fake.item(), "d_real": d_real.item()},
                "fake_segmap": seg_raw.detach()[:, :, ::4, ::4].clone(), "flow_last": last.detach().clone(),
                "warped_cm": w_cm.detach()[:, :, ::2, ::2].clone(),
                "pred_shapes": [tuple(p[-1].shape) for p in pred],
                "grad_summary_G": g_summ, "grad_summary_D": d_summ, "sample_grads_G": samples,
                "sample_grads_D": samples_d,
                "bn_after": {k: sd[k].clone() for k in ("ClothEncoder.0.block.1.running_mean",
                                                        "ClothEncoder.0.block.1.running_var",
                                                        "SegDecoder.4.block.4.running_var",
                                                        "out_layer.block.1.num_batches_tracked")},
                "param_after": {"flow_conv.4.bias": sd["flow_conv.4.bias"].clone(),
                                "PoseEncoder.2.block.0.weight": sd["PoseEncoder.2.block.0.weight"].clone(),
                                "D.layer1.0.bias": D.state_dict()["layer1.0.bias"].clone()}},
               os.path.join(OUT, "condstep_ngf8_128x96.pt"))
    print("condstep_ngf8_128x96.pt", os.path.getsize(os.path.join(OUT, "condstep_ngf8_128x96.pt")) // 1024, "KiB",
          {k: round(v, 5) for k, v in {"G": loss_G.item(), "D": loss_D.item(), "ce": ce.item(), "tv": loss_tv.item(),
                                       "l1": loss_l1.item(), "gan": g_gan.item()}.items()})


def dataset():
    """One sample of the REAL reference CPDatasetTest (cp_dataset_test.py) on the synthetic on-disk data set
    (hr_viton_amd.cp_dataset.write_synthetic_dataset, seed 0).  torchvision is absent: its three transforms are
    supplied by a stub module built from the product's restatements (so THOSE are not pinned by this golden --
    the label merging, the agnostic-person drawing, the mask threshold and the dictionary layout are)."""
    import tempfile
    import torch
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import cp_dataset as P
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor:
        def __call__(self, img):
            import numpy as np
            a = np.asarray(img)
            a = a[:, :, None] if a.ndim == 2 else a
            return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255)

    class Normalize:
        def __init__(self, m, s):
            self.m, self.s = torch.tensor(m).view(-1, 1, 1), torch.tensor(s).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.m) / self.s

    class Resize:
        def __init__(self, size, interpolation=2):
            self.size, self.interp = size, interpolation

        def __call__(self, img):
            return P.resize_to_width(img, self.size, self.interp)

    tvt.Compose, tvt.ToTensor, tvt.Normalize, tvt.Resize = Compose, ToTensor, Normalize, Resize
    tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    tv.transforms = tvt
    sys.modules["torchvision.transforms"] = tvt
    _import_reference()
    sys.path.insert(0, REF)
    import cp_dataset_test as ref_ds  # noqa
    sys.path.pop(0)
    with tempfile.TemporaryDirectory() as root:
        P.write_synthetic_dataset(root, n=2, seed=0)
        opt = Namespace(dataroot=root, datamode="test", data_list="test_pairs.txt", fine_height=64, fine_width=48,
                        semantic_nc=13)
        item = ref_ds.CPDatasetTest(opt)[1]
    keep = {k: (v if not isinstance(v, dict) else dict(v)) for k, v in item.items()}
    torch.save({"recipe": "hr_viton_amd.cp_dataset.write_synthetic_dataset(root, n=2, seed=0); CPDatasetTest(opt)[1], "
                          "fine 64x48", "item": keep}, os.path.join(OUT, "cpdataset_item1_64x48.pt"))
    print("cpdataset_item1_64x48.pt", os.path.getsize(os.path.join(OUT, "cpdataset_item1_64x48.pt")) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "condstep":
        condstep()
    elif len(sys.argv) > 1 and sys.argv[1] == "dataset":
        dataset()
    elif len(sys.argv) > 1 and sys.argv[1] == "tocg_variants":
        tocg_variants()
    else:
        main()
        condstep()
        dataset()
        tocg_variants()
