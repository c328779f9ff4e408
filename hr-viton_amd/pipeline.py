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
    if getattr(opt, "GT", False):
        # --GT (train_generator.py:253-256): ground-truth parse map and ground-truth warped cloth, no tocg
        with torch.no_grad():
            _, parse7 = glue.parse_from_scores(inputs["parse"])
            x = torch.cat((inputs["agnostic"], inputs["densepose"], inputs["parse_cloth"]), dim=1)
        return x, parse7
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
                         sync_g=None, sync_d=None, noise=None, noise_d=None):
    """One G step + one D step of train_generator.py:279-360.  ``parse7``: Act [N,H,W,8] (7 real).
    ``noise`` / ``noise_d``: the SPADE noise draws of the two generator forwards (default: drawn like the reference,
    network_generator.py:104-107); the data-parallel equivalence test injects them."""
    pair = hasattr(discriminator, "forward_pair") and not getattr(opt, "_hrv_no_pair", False)
    parse_nchw = None if pair else ops.to_nchw(parse7)
    # ---------------- generator ----------------
    if sync_g is not None:
        sync_g.begin()
    if sync_d is not None:
        sync_d.enabled = False          # D's gradients of the G step are discarded (:354)
    output_paired = generator(x, parse7, noise=noise)
    if not pair:
        fake_concat = torch.cat((parse_nchw, output_paired), dim=1)
        real_concat = torch.cat((parse_nchw, im), dim=1)
    discriminator._hrv_discard_param_grads = True      # :354 zeroes D's gradients of loss_gen before they are ever used
    try:
        if pair:      # the same [fake ; real] batch, assembled NHWC inside (no cat / layout round trip)
            pred_fake, pred_real = discriminator.forward_pair(parse7, output_paired, im)
        else:
            pred_fake, pred_real = discriminator(torch.cat((fake_concat, real_concat), dim=0), split=True)   # :283-295
    finally:
        discriminator._hrv_discard_param_grads = False
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
        output = generator(x, parse7, noise=noise_d)       # new noise, post-update weights (:327-330)
    if pair:
        pred_fake, pred_real = discriminator.forward_pair(parse7, output, im)
    else:
        fake_concat = torch.cat((parse_nchw, output), dim=1)
        pred_fake, pred_real = discriminator(torch.cat((fake_concat, real_concat), dim=0), split=True)
    d_losses = {"D_Fake": crit_gan(pred_fake, False, for_discriminator=True),
                "D_Real": crit_gan(pred_real, True, for_discriminator=True)}
    loss_dis = sum(d_losses.values()).mean()
    opt_d.zero_grad()
    loss_dis.backward()
    opt_d.step()
    losses.update(d_losses)
    return losses, output_paired


def remove_overlap(seg_out, warped_cm):
    """train_condition.py:38-43 (same as test_generator.py:19-24)."""
    assert len(warped_cm.shape) == 4
    return warped_cm - (torch.cat([seg_out[:, 1:3, :, :], seg_out[:, 5:, :, :]], dim=1)).sum(dim=1, keepdim=True) * warped_cm


def condition_train_step(opt, tocg, D, crit_l1, crit_vgg, crit_gan, opt_g, opt_d, inputs: Dict[str, torch.Tensor],
                         sync_g=None, sync_d=None):
    """One iteration of train_condition.py:136-286 (the default `not G_D_seperate` order): tocg forward with
    batch-statistics BatchNorm, warping / TV / interflow / cross-entropy / LSGAN losses, G step, D step.
    The networks, warps, softmax, cross entropy, TV, L1, VGG and LSGAN terms run on the HIP kernels
    (cond_train.py, functional.py, losses.py, vgg.py); torch only stitches the scalar sums and the
    elementwise mask compositions.  ``inputs`` as produced by cp_dataset.py: cloth, cloth_mask,
    parse_agnostic, densepose, parse_onehot (label indices [N,1,H,W]), parse (one-hot), pcm, parse_cloth.

    Stated deviations from the reference's schedule (same losses, gradients and updates per iteration):
    * the reference calls D three times before either optimizer step (G pass, fake, real: :262-274); here the fake and
      real batches of the D loss share ONE forward over [fake; real] and it runs after opt_g.step() -- InstanceNorm is
      per sample, so the values are the same, but with --spectral D's power iteration advances twice per iteration
      instead of three times (the (u, v) trajectory differs from a reference run), and with --Ddropout the mask draws
      are consumed in a different order;
    * under data-parallel training tocg's BatchNorm uses per-rank batch statistics (the reference's nn.DataParallel
      does too: its sync_batchnorm package is never instantiated), and checkpoints hold rank 0's running statistics."""
    from . import functional as HF
    from .networks import make_grid
    c_paired = inputs["cloth"]
    cm_paired = (inputs["cloth_mask"] > 0.5).to(torch.float32)                 # :140 without the numpy round trip
    label_onehot, label, pcm, im_c = inputs["parse_onehot"], inputs["parse"], inputs["pcm"], inputs["parse_cloth"]
    input1 = torch.cat([c_paired, cm_paired], 1)
    input2 = torch.cat([inputs["parse_agnostic"], inputs["densepose"]], 1)
    if sync_g is not None:
        sync_g.begin()
    if sync_d is not None:
        sync_d.enabled = False          # D's gradients of loss_G are discarded by optimizer_D.zero_grad() (:284)
    flow_list, fake_segmap, warped_cloth, warped_cm = tocg(input1, input2)      # :158
    comp = getattr(opt, "clothmask_composition", "warp_grad")
    if comp != "no_composition":                                               # :164-173
        cloth_mask = torch.ones_like(fake_segmap.detach())
        cloth_mask[:, 3:4, :, :] = (warped_cm.detach() > 0.5).to(torch.float32) if comp == "detach" else warped_cm
        fake_segmap = fake_segmap * cloth_mask
    if getattr(opt, "occlusion", False):                                       # :174-176
        warped_cm = remove_overlap(HF.softmax(fake_segmap, dim=1), warped_cm)
        warped_cloth = warped_cloth * warped_cm + torch.ones_like(warped_cloth) * (1 - warped_cm)
    loss_l1_cloth = crit_l1(warped_cm, pcm)                                    # :184
    use_vgg = crit_vgg is not None
    loss_vgg = crit_vgg(warped_cloth, im_c) if use_vgg else torch.zeros((), device=c_paired.device)   # :185
    loss_tv = 0
    edge = getattr(opt, "edgeawaretv", "no_edge")
    if edge == "no_edge":
        for flow in (flow_list[-1:] if getattr(opt, "lasttvonly", False) else flow_list):   # :190-199
            loss_tv = loss_tv + HF.tv_loss(flow)
    else:
        # edge-aware TV (:200-229): |d flow| weighted by exp(-150 |d mask|) of the warped cloth mask resized to the
        # flow's resolution.  The resize runs on the HIP kernel; the weighting is elementwise glue on flow-sized maps.
        levels = [4] if edge == "last_only" else list(range(5))
        for i in levels:
            flow = flow_list[i]
            wcd = HF.interpolate(warped_cm, size=flow.shape[1:3], mode="bilinear").permute(0, 2, 3, 1)
            y_tv = torch.abs(flow[:, 1:, :, :] - flow[:, :-1, :, :]) * torch.exp(-150 * torch.abs(wcd[:, 1:] - wcd[:, :-1]))
            x_tv = torch.abs(flow[:, :, 1:, :] - flow[:, :, :-1, :]) * torch.exp(-150 * torch.abs(wcd[:, :, 1:] - wcd[:, :, :-1]))
            scale = 1.0 if edge == "last_only" else 1.0 / (2 ** (4 - i))
            loss_tv = loss_tv + y_tv.mean() * scale + x_tv.mean() * scale
        if getattr(opt, "add_lasttv", False):
            loss_tv = loss_tv + HF.tv_loss(flow_list[-1])
    N, _, iH, iW = c_paired.size()
    if getattr(opt, "interflowloss", False):                                   # :235-248
        soft_for_overlap = HF.softmax(fake_segmap, dim=1)
        grid = make_grid(N, iH, iW).to(c_paired.device)
        for i in range(len(flow_list) - 1):
            flow = flow_list[i]
            _, fH, fW, _ = flow.size()
            flow = HF.interpolate(flow.permute(0, 3, 1, 2), size=c_paired.shape[2:],
                                  mode=getattr(opt, "upsample", "bilinear")).permute(0, 2, 3, 1)      # :242 (--upsample)
            flow_norm = torch.cat([flow[:, :, :, 0:1] / ((fW - 1.0) / 2.0), flow[:, :, :, 1:2] / ((fH - 1.0) / 2.0)], 3)
            warped_cm_i = HF.grid_sample(cm_paired, flow_norm + grid, padding_mode="border")
            warped_cm_i = remove_overlap(soft_for_overlap, warped_cm_i)
            loss_l1_cloth = loss_l1_cloth + crit_l1(warped_cm_i, pcm) / (2 ** (4 - i))
            if use_vgg:
                warped_c_i = HF.grid_sample(c_paired, flow_norm + grid, padding_mode="border")
                loss_vgg = loss_vgg + crit_vgg(warped_c_i, im_c) / (2 ** (4 - i))
    CE_loss = HF.cross_entropy2d(fake_segmap, label_onehot.transpose(0, 1)[0].long())   # :252
    losses = {"l1": loss_l1_cloth, "vgg": loss_vgg, "tv": loss_tv, "ce": CE_loss}
    if getattr(opt, "no_GAN_loss", False):
        loss_G = (10 * loss_l1_cloth + loss_vgg + opt.tvlambda * loss_tv) + (CE_loss * opt.CElamda)
        opt_g.zero_grad()
        loss_G.backward()
        opt_g.step()
        losses["loss_G"] = loss_G
        return losses
    fake_segmap_softmax = HF.softmax(fake_segmap, 1)                            # :260
    D._hrv_discard_param_grads = True                   # optimizer_D.zero_grad() (:284) discards them
    try:
        pred_segmap = D(torch.cat((input1.detach(), input2.detach(), fake_segmap_softmax), dim=1))
    finally:
        D._hrv_discard_param_grads = False
    loss_G_GAN = crit_gan(pred_segmap, True)
    loss_G = (10 * loss_l1_cloth + loss_vgg + opt.tvlambda * loss_tv) + (CE_loss * opt.CElamda + loss_G_GAN * opt.GANlambda)
    opt_g.zero_grad()
    loss_G.backward()
    opt_g.step()
    # discriminator (:267-277,284-286).  InstanceNorm is per sample, so the fake and real batches share one
    # launch sequence; the fake half repeats the G pass's forward on detached inputs, as the reference does.
    if sync_d is not None:
        sync_d.enabled = True
        sync_d.begin()
    if getattr(opt, "G_D_seperate", False):                                   # :296-300 -- D sees the UPDATED generator
        with torch.no_grad():
            _, fake_segmap_new, _, _ = tocg(input1, input2)
        fake_segmap_softmax = HF.softmax(fake_segmap_new, 1)
    both = torch.cat((torch.cat((input1, input2, fake_segmap_softmax.detach()), dim=1),
                      torch.cat((input1, input2, label), dim=1)), dim=0)
    pred = D(both)
    pred_f = [[t[:N] for t in p] for p in pred]
    pred_r = [[t[N:] for t in p] for p in pred]
    loss_D_fake, loss_D_real = crit_gan(pred_f, False), crit_gan(pred_r, True)
    loss_D = loss_D_fake + loss_D_real
    opt_d.zero_grad()
    loss_D.backward()
    opt_d.step()
    losses.update({"g_gan": loss_G_GAN, "loss_G": loss_G, "d_fake": loss_D_fake, "d_real": loss_D_real, "loss_D": loss_D})
    return losses
