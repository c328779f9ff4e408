"""Parse-map glue between the condition generator and the image generator
(test_generator.py:161-217; train_generator.py:217-275) on the HIP path: cloth-mask
composition, bilinear up-sampling, 15x15 Gaussian, argmax -> one-hot -> 13->7
merge, high-resolution cloth warp, occlusion handling.  No host round trips (the
reference's ``.cpu().numpy() > 0.5`` becomes a device-side compare)."""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib, ops
from .ops import Act


def gaussian_taps(ksize: int = 15, sigma: float = 3.0) -> torch.Tensor:
    """torchgeometry 0.1.2 ``gaussian``: exp(-(x - k//2)^2 / (2 sigma^2)), normalised (fp32)."""
    xs = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    g = torch.exp(-(xs ** 2) / float(2 * sigma ** 2))
    return (g / g.sum()).contiguous()


_TAPS = {}


def make_parse(fake_segmap: torch.Tensor, warped_cm: torch.Tensor, fine_height: int, fine_width: int,
               composition: str = "warp_grad", want_labels: bool = True):
    """test_generator.py:167-203.  fake_segmap [N,13,h,w], warped_cm [N,1,h,w] (NCHW, cuda).
    Returns (gauss Act [N,H,W,16] (13 real), labels int64 [N,1,H,W] or None, parse7 Act [N,H,W,8] (7 real))."""
    lib = _lib.load()
    ops.require_cuda(fake_segmap, "make_parse(fake_segmap)")
    ops.require_cuda(warped_cm, "make_parse(warped_cm)")
    N, Cn, h, w = fake_segmap.shape
    assert Cn == 13, "13-class segmentation map expected"
    seg = ops.to_nhwc(fake_segmap)                 # [N,h,w,16]
    st = ops._stream()
    if composition != "no_composition":
        cm = warped_cm.contiguous()                # [N,1,h,w] == NHWC with 1 channel
        _lib.check(lib.hrv_mul_channel_nhwc_f32(seg.t.data_ptr(), seg.cstride, 3, cm.data_ptr(), 1, 0,
                                                1 if composition == "detach" else 0, N * h * w, st),
                   "hrv_mul_channel_nhwc_f32")
    up = ops.resize_bilinear(seg, fine_height, fine_width, h / fine_height, w / fine_width)
    key = (15, 3.0)
    if key not in _TAPS:
        _TAPS[key] = gaussian_taps(*key)
    taps = _TAPS[key]
    tmp = torch.empty_like(up.t)
    out = torch.empty_like(up.t)
    with ops._Timed("glue", "gauss_blur15", 0.0, 4.0 * up.t.numel() * 4):
        _lib.check(lib.hrv_gauss_blur_nhwc_f32(up.t.data_ptr(), N, fine_height, fine_width, up.cstride, up.cstride,
                                               taps.data_ptr(), 15, tmp.data_ptr(), out.data_ptr(), st),
                   "hrv_gauss_blur_nhwc_f32")
    gauss = Act(out, 13)
    labels = torch.empty((N, 1, fine_height, fine_width), dtype=torch.int64, device=out.device) if want_labels else None
    parse7 = Act(torch.empty((N, fine_height, fine_width, 8), dtype=torch.float32, device=out.device), 7)
    with ops._Timed("glue", "parse_argmax", 0.0, 4.0 * N * fine_height * fine_width * (16 + 8 + 2)):
        _lib.check(lib.hrv_parse_argmax_nhwc_f32(out.data_ptr(), gauss.cstride, 13, N * fine_height * fine_width,
                                                 None if labels is None else labels.data_ptr(),
                                                 parse7.t.data_ptr(), 8, st), "hrv_parse_argmax_nhwc_f32")
    return gauss, labels, parse7


def parse_from_scores(scores: torch.Tensor, want_labels: bool = False):
    """argmax over 13 classes -> one-hot -> 13->7 merge of an NCHW score / one-hot map (the --GT branch of
    train_generator.py:253-274: ``fake_parse = parse_GT.argmax(dim=1)``).  Returns (labels or None, parse7 Act)."""
    lib = _lib.load()
    ops.require_cuda(scores, "parse_from_scores")
    N, Cn, H, W = scores.shape
    assert Cn == 13
    a = ops.to_nhwc(scores)
    labels = torch.empty((N, 1, H, W), dtype=torch.int64, device=scores.device) if want_labels else None
    parse7 = Act(torch.empty((N, H, W, 8), dtype=torch.float32, device=scores.device), 7)
    _lib.check(lib.hrv_parse_argmax_nhwc_f32(a.t.data_ptr(), a.cstride, 13, N * H * W,
                                             None if labels is None else labels.data_ptr(), parse7.t.data_ptr(), 8,
                                             ops._stream()), "hrv_parse_argmax_nhwc_f32")
    return labels, parse7


def hires_warp(flow_last: torch.Tensor, clothes: torch.Tensor, cloth_mask: torch.Tensor,
               norm_x: float = (96 - 1.0) / 2.0, norm_y: float = (128 - 1.0) / 2.0) -> Act:
    """test_generator.py:206-213: up-sample the last flow to the cloth size (size= resize), normalise
    by the hard-coded ((96-1)/2, (128-1)/2), warp cloth + mask with ONE launch.
    Returns Act [N,H,W,4] (cloth rgb, mask)."""
    ops.require_cuda(clothes, "hires_warp(clothes)")
    N, _, iH, iW = clothes.shape
    src = ops.to_nhwc(torch.cat([clothes, cloth_mask], 1))
    fh, fw = flow_last.shape[1], flow_last.shape[2]
    out, _ = ops.flow_warp(src, flow_last.contiguous(), iH, iW, fh / iH, fw / iW, norm_x, norm_y, want_flow_up=False)
    return out


def occlusion(gauss: Act, warped: Act) -> Act:
    """test_generator.py:214-216 (--occlusion), in place on ``warped`` ([N,H,W,4]: cloth rgb + mask)."""
    lib = _lib.load()
    n = warped.N * warped.H * warped.W
    _lib.check(lib.hrv_occlusion_nhwc_f32(gauss.t.data_ptr(), gauss.cstride, 13, warped.t.data_ptr(), warped.cstride,
                                          warped.t.data_ptr(), warped.cstride, 3, n, ops._stream()),
               "hrv_occlusion_nhwc_f32")
    return warped


def resize_nchw(x: torch.Tensor, size: Tuple[int, int], mode: str = "bilinear") -> torch.Tensor:
    """F.interpolate(x, size=size, mode=mode) on NCHW (test_generator.py:144-150)."""
    lib = _lib.load()
    ops.require_cuda(x, "resize_nchw")
    x = x.contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc, size[0], size[1]), dtype=torch.float32, device=x.device)
    _lib.check(lib.hrv_resize_nchw_f32(x.data_ptr(), N * Cc, H, W, size[0], size[1], 1 if mode == "nearest" else 0,
                                       out.data_ptr(), ops._stream()), "hrv_resize_nchw_f32")
    return out
