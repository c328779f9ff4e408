"""Diagnostic (GPU): per-parameter tocg gradient error of one train_condition.py iteration against the oracle for
variants of the synthetic batch / loss configuration -- localises a data-dependent discrepancy."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import networks, pipeline  # noqa: E402
from hr_viton_amd.losses import L1Loss  # noqa: E402
from hr_viton_amd.optim import Adam  # noqa: E402
from oracle import hrviton_oracle as O  # noqa: E402
from oracle.recipes import condstep_build  # noqa: E402


def run(tag, vary, gan=True, clamp=True, amp=0.2, tv=2.0, ce=10.0, ganl=1.0, comp="warp_grad"):
    opt, tocg, D, batch0 = condstep_build(networks.ConditionGenerator, networks.define_D)
    opt.lasttvonly, opt.interflowloss, opt.occlusion, opt.clothmask_composition = True, True, False, comp
    opt.tvlambda, opt.CElamda, opt.GANlambda, opt.no_GAN_loss = tv, ce, ganl if gan else 0.0, False
    g = torch.Generator().manual_seed(7)
    b = dict(batch0)
    for k in ("cloth", "densepose", "parse_cloth"):
        n = amp * F.interpolate(torch.randn(2, 3, 16, 12, generator=g), scale_factor=8, mode="bilinear")
        if k in vary:
            b[k] = batch0[k] + n
            if clamp:
                b[k] = b[k].clamp(-1, 1)
    sd_g = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running_" not in k)
            for k, v in tocg.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    r = O.condition_train_losses(sd_g, sd_d, None, b, composition=comp)
    r["loss_G"] = 10 * r["l1"] + tv * r["tv"] + ce * r["ce"] + (ganl if gan else 0.0) * r["g_gan"]
    r["loss_G"].backward(retain_graph=True)
    want = {k: v.grad.clone() for k, v in sd_g.items() if v.grad is not None}
    tocg.cuda().train()
    D.cuda().train()
    # floor() decisions of the warps: sample coordinates from the HIP flows vs the oracle's flows (same formula)
    with torch.no_grad():
        in1 = torch.cat([b["cloth"], b["cloth_mask"]], 1).cuda()
        in2 = torch.cat([b["parse_agnostic"], b["densepose"]], 1).cuda()
        mom = {m: m.momentum for m in tocg.modules() if isinstance(m, torch.nn.BatchNorm2d)}
        for m in mom:
            m.momentum = 0.0                      # leave the running statistics alone
        hflows = tocg(opt, in1, in2)[0]
        for m, v in mom.items():
            m.momentum = v

    def coords(flow, size):
        fH, fW = flow.shape[1:3]
        H, W = size
        f = flow if (fH, fW) == (H, W) else F.interpolate(flow.permute(0, 3, 1, 2), size=size, mode="bilinear").permute(0, 2, 3, 1)
        gx = torch.linspace(-1.0, 1.0, W).view(1, 1, W) + f[..., 0] / ((fW - 1.0) / 2.0)
        gy = torch.linspace(-1.0, 1.0, H).view(1, H, 1) + f[..., 1] / ((fH - 1.0) / 2.0)
        return ((gx + 1) * W - 1) / 2, ((gy + 1) * H - 1) / 2

    flips, near, dmax = 0, 0, 0.0
    for hf, of in zip(hflows, r["flow_list"]):
        for size in {tuple(of.shape[1:3]), (128, 96)}:
            for a, c in zip(coords(hf.detach().cpu().float(), size), coords(of.detach(), size)):
                flips += int((a.floor() != c.floor()).sum())
                near += int(((c - c.round()).abs() < 1e-4).sum())
                dmax = max(dmax, float((a - c).abs().max()))
    print(f"    warp coordinates: {flips} floor() mismatches HIP vs oracle, {near} oracle samples within 1e-4 px of an "
          f"integer, max |d coord| {dmax:.2e} px")
    hg = Adam(tocg.parameters(), lr=2e-4, betas=(0.5, 0.999))
    hd = Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    cap = {}
    real = hg.step

    def capture():
        cap.update({n: p.grad.detach().cpu().clone() for n, p in tocg.named_parameters() if p.grad is not None})
        return real()

    hg.step = capture
    losses = pipeline.condition_train_step(opt, tocg, D, L1Loss(), None, networks.GANLoss(use_lsgan=True), hg, hd,
                                           {k: v.cuda() for k, v in b.items()})
    gmax = max(v.abs().max().item() for v in want.values())
    errs = sorted(((cap[k] - w).abs().max().item() / max(w.abs().max().item(), 1e-3 * gmax), k) for k, w in want.items())
    terms = {k: (float(losses[k].detach()), float(r[k].detach())) for k in ("l1", "tv", "ce", "g_gan", "loss_G") if k in losses and k in r}
    print(f"{tag:34s} median {errs[len(errs) // 2][0]:.2e} worst {errs[-1][0]:.2e} {errs[-1][1]}")
    print("    ", " ".join(f"{k}={a:.6f}/{w:.6f}" for k, (a, w) in terms.items()))
    return errs


if __name__ == "__main__":
    run("baseline batch0", ())
    run("all varied (test iteration 0)", ("cloth", "densepose", "parse_cloth"))
    run("cloth only", ("cloth",))
    run("densepose only", ("densepose",))
    run("densepose amp 1e-3", ("densepose",), amp=1e-3)
    run("densepose amp 1e-2", ("densepose",), amp=1e-2)
    run("all varied amp 0.05", ("cloth", "densepose", "parse_cloth"), amp=0.05)
