"""End-to-end try-on inference step on the HIP path: the body of the reference's
``test()`` loop (test_generator.py:118-219) without its host round trips."""
from __future__ import annotations

from typing import Dict

import torch

from . import glue, ops


@torch.no_grad()
def tryon_step(opt, tocg, generator, inputs: Dict[str, torch.Tensor], noise=None) -> Dict[str, torch.Tensor]:
    """inputs (cuda, NCHW fp32, fine resolution): 'cloth' [N,3,H,W], 'cloth_mask' [N,1,H,W],
    'parse_agnostic' [N,13,H,W], 'densepose' [N,3,H,W], 'agnostic' [N,3,H,W].
    Returns 'output' [N,3,H,W] plus the intermediates the reference visualises."""
    clothes, densepose, agnostic = inputs["cloth"], inputs["densepose"], inputs["agnostic"]
    H, W = opt.fine_height, opt.fine_width
    # test_generator.py:128 -- (mask > 0.5) stays on the device
    pre_clothes_mask = (inputs["cloth_mask"] > 0.5).to(torch.float32)
    # :144-150 down-sampling to the tocg resolution
    lo = (256, 192)
    input1 = torch.cat([glue.resize_nchw(clothes, lo, "bilinear"), glue.resize_nchw(pre_clothes_mask, lo, "nearest")], 1)
    input2 = torch.cat([glue.resize_nchw(inputs["parse_agnostic"], lo, "nearest"),
                        glue.resize_nchw(densepose, lo, "bilinear")], 1)
    flow_list, fake_segmap, warped_cloth_paired, warped_cm_paired = tocg(opt, input1, input2)   # :159
    comp = getattr(opt, "clothmask_composition", "warp_grad")
    gauss, labels, parse7 = glue.make_parse(fake_segmap, warped_cm_paired, H, W, comp)           # :167-203
    warped = glue.hires_warp(flow_list[-1], clothes, pre_clothes_mask)                           # :206-213
    if getattr(opt, "occlusion", False):
        glue.occlusion(gauss, warped)                                                            # :214-216
    warped_cloth = ops.to_nchw(warped, 0, 3)
    warped_clothmask = ops.to_nchw(warped, 3, 1)
    x = torch.cat((agnostic, densepose, warped_cloth), dim=1)                                    # :219
    output = generator(x, parse7, noise=noise) if noise is not None else generator(x, parse7)
    return {"output": output, "warped_cloth": warped_cloth, "warped_clothmask": warped_clothmask,
            "fake_parse_gauss": gauss, "fake_parse": labels, "parse": parse7, "flow_list": flow_list,
            "fake_segmap": fake_segmap}


def make_generator_inputs(opt, tocg, inputs: Dict[str, torch.Tensor]):
    """train_generator.py:201-275 (the no_grad block): frozen tocg at 256x192 -> parse glue ->
    high-resolution cloth warp.  Returns (x [N,9,H,W], parse7 Act)."""
    with torch.no_grad():
        c_paired, pose, agnostic = inputs["cloth"], inputs["densepose"], inputs["agnostic"]
        H, W = opt.fine_height, opt.fine_width
        cm = (inputs["cloth_mask"] > 0.5).to(torch.float32)              # :218 without the numpy round trip
        lo = (256, 192)
        input1 = torch.cat([glue.resize_nchw(c_paired, lo, "bilinear"), glue.resize_nchw(cm, lo, "nearest")], 1)
        input2 = torch.cat([glue.resize_nchw(inputs["parse_agnostic"], lo, "nearest"),
                            glue.resize_nchw(pose, lo, "bilinear")], 1)
        flow_list, fake_segmap, _, warped_cm_paired = tocg(input1, input2)  # :215
        comp = getattr(opt, "clothmask_composition", "warp_grad")
        gauss, _, parse7 = glue.make_parse(fake_segmap, warped_cm_paired, H, W, comp, want_labels=False)
        warped = glue.hires_warp(flow_list[-1], c_paired, cm)
        if getattr(opt, "occlusion", False):
            glue.occlusion(gauss, warped)
        x = torch.cat((agnostic, pose, ops.to_nchw(warped, 0, 3)), dim=1)    # :279
    return x, parse7


def generator_train_step(opt, generator, discriminator, crit_gan, crit_feat, crit_vgg, opt_g, opt_d, x, parse7, im,
                         sync_g=None, sync_d=None):
    """One G step + one D step of train_generator.py:279-360.  ``parse7``: Act [N,H,W,8] (7 real)."""
    parse_nchw = ops.to_nchw(parse7)
    # ---------------- generator ----------------
    if sync_g is not None:
        sync_g.begin()
    if sync_d is not None:
        sync_d.enabled = False          # D's gradients of the G step are discarded (:354)
    output_paired = generator(x, parse7)
    fake_concat = torch.cat((parse_nchw, output_paired), dim=1)
    real_concat = torch.cat((parse_nchw, im), dim=1)
    pred = discriminator(torch.cat((fake_concat, real_concat), dim=0))
    pred_fake = [[t[: t.size(0) // 2] for t in p] for p in pred]
    pred_real = [[t[t.size(0) // 2:] for t in p] for p in pred]
    losses = {"GAN": crit_gan(pred_fake, True, for_discriminator=False)}
    if not getattr(opt, "no_ganFeat_loss", False):
        num_D = len(pred_fake)
        feat = 0
        for i in range(num_D):
            for j in range(len(pred_fake[i]) - 1):
                feat = feat + crit_feat(pred_fake[i][j], pred_real[i][j].detach()) * opt.lambda_feat / num_D
        losses["GAN_Feat"] = feat
    if crit_vgg is not None and not getattr(opt, "no_vgg_loss", False):
        losses["VGG"] = crit_vgg(output_paired, im) * opt.lambda_vgg
    loss_gen = sum(losses.values()).mean()
    opt_g.zero_grad()
    loss_gen.backward()
    opt_g.step()
    # ---------------- discriminator ----------------
    if sync_d is not None:
        sync_d.enabled = True
        sync_d.begin()
    with torch.no_grad():
        output = generator(x, parse7)       # new noise, post-update weights (:327-330)
    fake_concat = torch.cat((parse_nchw, output), dim=1)
    pred = discriminator(torch.cat((fake_concat, real_concat), dim=0))
    pred_fake = [[t[: t.size(0) // 2] for t in p] for p in pred]
    pred_real = [[t[t.size(0) // 2:] for t in p] for p in pred]
    d_losses = {"D_Fake": crit_gan(pred_fake, False, for_discriminator=True),
                "D_Real": crit_gan(pred_real, True, for_discriminator=True)}
    loss_dis = sum(d_losses.values()).mean()
    opt_d.zero_grad()
    loss_dis.backward()
    opt_d.step()
    losses.update(d_losses)
    return losses, output_paired
