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
