"""Discriminator-rejection flow of the condition generator on the HIP path:
``get_norm_const.py:60-132`` (largest odds l/(1-l) of the segmentation discriminator's logit over a
data set) and ``test_condition.py:64-127`` (per-sample rejection score = odds / norm_const, plus the
misalignment mask).  Everything heavy -- tocg (eval BatchNorm), the mask composition, the channel
softmax and the two-scale PatchGAN -- runs on the HIP kernels; what is left in torch are means of
[N,1,h,w] maps and scalar arithmetic."""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

import torch

from . import functional as HF


def D_logit(pred) -> torch.Tensor:
    """get_norm_const.py:60-64 / test_condition.py: mean of the last map of every scale, halved and summed."""
    score = 0
    for p in pred:
        score = score + p[-1].mean((1, 2, 3)) / 2
    return score


@torch.no_grad()
def condition_outputs(opt, tocg, batch: Dict[str, torch.Tensor]):
    """tocg forward + cloth-mask composition (test_condition.py:98-116).  Returns input1, input2, the composed
    fake_segmap, warped cloth, warped mask and its binarisation."""
    c, cm = batch["cloth"], (batch["cloth_mask"] > 0.5).to(torch.float32)
    input1 = torch.cat([c, cm], 1)
    input2 = torch.cat([batch["parse_agnostic"], batch["densepose"]], 1)
    flow_list, fake_segmap, warped_c, warped_cm = tocg(input1, input2)
    warped_cm_onehot = (warped_cm > 0.5).to(torch.float32)
    comp = getattr(opt, "clothmask_composition", "warp_grad")
    if comp != "no_composition":
        mask = torch.ones_like(fake_segmap)
        mask[:, 3:4, :, :] = warped_cm_onehot if comp == "detach" else warped_cm
        fake_segmap = fake_segmap * mask
    return input1, input2, fake_segmap, warped_c, warped_cm, warped_cm_onehot, flow_list


@torch.no_grad()
def segmap_logits(opt, tocg, D, batch: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """(logit_real, logit_fake) per sample -- get_norm_const.py:98-118."""
    input1, input2, fake_segmap, *_ = condition_outputs(opt, tocg, batch)
    soft = HF.softmax(fake_segmap, dim=1)
    real = D(torch.cat((input1, input2, batch["parse"]), dim=1))
    fake = D(torch.cat((input1, input2, soft), dim=1))
    return D_logit(real), D_logit(fake)


@torch.no_grad()
def get_const(opt, batches: Iterable[Dict[str, torch.Tensor]], tocg, D) -> float:
    """get_norm_const.py:65-132: the largest odds l/(1-l) over real and fake logits of all batches."""
    tocg.eval()
    D.eval()
    odds: List[float] = []
    for batch in batches:
        lr, lf = segmap_logits(opt, tocg, D, batch)
        for l in torch.cat([lr, lf]).tolist():
            odds.append(l / (1 - l))
    odds.sort()
    return odds[-1]


@torch.no_grad()
def rejection_scores(opt, tocg, D, batch: Dict[str, torch.Tensor], norm_const: float):
    """test_condition.py:98-127: per-sample rejection score and the misalignment mask."""
    input1, input2, fake_segmap, warped_c, warped_cm, warped_cm_onehot, _ = condition_outputs(opt, tocg, batch)
    score = None
    if D is not None:
        soft = HF.softmax(fake_segmap, dim=1)
        s = D_logit(D(torch.cat((input1, input2, soft), dim=1)))
        score = (s / (1 - s)) / norm_const
    fake_clothmask = (torch.argmax(fake_segmap, dim=1, keepdim=True) == 3).long()
    misalign = (fake_clothmask - warped_cm_onehot.long()).clamp_min(0)
    return score, misalign, fake_segmap, warped_c, warped_cm_onehot
