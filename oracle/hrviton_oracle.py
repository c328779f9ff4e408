"""CPU oracle for the HR-VITON hot path -- TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (functional, state-dict driven, CPU, fp32) of the
reference algorithm on the hot path.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, as the
checker -- never by the product package ``hr-viton_amd`` and never as the thing
that is measured or shipped.

Parity pinning: ``oracle/make_golden.py`` imports the real reference modules
from ``/root/reference`` (read-only, CPU) and writes golden input/output
vectors to ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
function below against those vectors.  The one piece that is **parity
unpinned** is ``gaussian_blur`` (torchgeometry 0.1.2 ``GaussianBlur`` is a
third-party dependency that is not vendored in /root/reference and not
installed here); its published algorithm is restated and the call sites
(test_generator.py:91,179) are cited.

Every function cites the reference file:line it follows.  Convolutions,
batch-norm and elementwise math are executed with plain ``torch`` CPU fp32 ops
("oracle by execution"); the sampling operators whose semantics are
parity-critical (grid_sample, bilinear / nearest resize, base grid) are
restated explicitly in index arithmetic so the HIP kernels have a spelled-out
definition to match.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# --------------------------------------------------------------------------
# Sampling primitives (explicit restatements)
# --------------------------------------------------------------------------


def linspace_m1_1(n: int) -> Tensor:
    """torch.linspace(-1, 1, n) in fp32 -- networks.py:162-163.

    torch's CPU kernel evaluates ``start + i*step`` for the first half and
    ``end - (n-1-i)*step`` for the second half with ``step=(end-start)/(n-1)``
    in fp32; restated so the HIP kernel can reproduce the exact values.
    """
    if n == 1:
        return torch.tensor([-1.0], dtype=torch.float32)
    step = torch.tensor(2.0, dtype=torch.float32) / torch.tensor(float(n - 1), dtype=torch.float32)
    i = torch.arange(n, dtype=torch.float32)
    lo = -1.0 + i * step
    hi = 1.0 - (float(n - 1) - i) * step
    return torch.where(torch.arange(n) < n // 2, lo, hi).to(torch.float32)


def make_grid(N: int, iH: int, iW: int) -> Tensor:
    """Base sampling grid [N,iH,iW,2] (x,y) -- networks.py:161-168."""
    gx = linspace_m1_1(iW).view(1, 1, iW, 1).expand(N, iH, iW, 1)
    gy = linspace_m1_1(iH).view(1, iH, 1, 1).expand(N, iH, iW, 1)
    return torch.cat([gx, gy], 3).contiguous()


def grid_sample_bilinear_border(inp: Tensor, grid: Tensor) -> Tensor:
    """F.grid_sample(inp, grid, mode='bilinear', padding_mode='border',
    align_corners=False) restated -- call sites networks.py:135,143,152;
    test_generator.py:212-213.  inp [N,C,H,W], grid [N,Ho,Wo,2] -> [N,C,Ho,Wo].
    """
    N, C, H, W = inp.shape
    gx = grid[..., 0]
    gy = grid[..., 1]
    ix = ((gx + 1.0) * W - 1.0) / 2.0
    iy = ((gy + 1.0) * H - 1.0) / 2.0
    ix = ix.clamp(0.0, float(W - 1))
    iy = iy.clamp(0.0, float(H - 1))
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1.0
    y1 = y0 + 1.0
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)

    def tap(xf: Tensor, yf: Tensor, w: Tensor) -> Tensor:
        xi = xf.long()
        yi = yf.long()
        valid = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        xi = xi.clamp(0, W - 1)
        yi = yi.clamp(0, H - 1)
        lin = (yi * W + xi).view(N, 1, -1).expand(N, C, -1)
        v = inp.reshape(N, C, H * W).gather(2, lin).view(N, C, *xf.shape[1:])
        return v * (w * valid.to(inp.dtype)).unsqueeze(1)

    return tap(x0, y0, w_nw) + tap(x1, y0, w_ne) + tap(x0, y1, w_sw) + tap(x1, y1, w_se)


def _lin_src(out_size: int, in_size: int, rscale: float) -> Tuple[Tensor, Tensor, Tensor]:
    dst = torch.arange(out_size, dtype=torch.float32)
    src = (rscale * (dst + 0.5) - 0.5).clamp(min=0.0)
    i0 = src.floor().long().clamp(max=in_size - 1)
    i1 = torch.where(i0 < in_size - 1, i0 + 1, i0)
    lam = (src - i0.to(torch.float32)).clamp(0.0, 1.0)
    return i0, i1, lam


def resize_bilinear(x: Tensor, size: Optional[Tuple[int, int]] = None,
                    scale_factor: Optional[float] = None) -> Tensor:
    """F.interpolate(mode='bilinear', align_corners=False) restated --
    networks.py:130-133,150,181; test_generator.py:144-150,179,207.
    With ``scale_factor`` the source-index ratio is 1/scale_factor; with
    ``size`` it is in/out.
    """
    N, C, H, W = x.shape
    if scale_factor is not None:
        Ho, Wo = int(math.floor(H * scale_factor)), int(math.floor(W * scale_factor))
        rh = rw = 1.0 / scale_factor
    else:
        Ho, Wo = size
        rh, rw = H / Ho, W / Wo
    y0, y1, ly = _lin_src(Ho, H, rh)
    x0, x1, lx = _lin_src(Wo, W, rw)
    ly = ly.view(1, 1, Ho, 1)
    lx = lx.view(1, 1, 1, Wo)
    r0 = x[:, :, y0]
    r1 = x[:, :, y1]
    top = r0[:, :, :, x0] * (1 - lx) + r0[:, :, :, x1] * lx
    bot = r1[:, :, :, x0] * (1 - lx) + r1[:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def resize_nearest(x: Tensor, size: Tuple[int, int]) -> Tensor:
    """F.interpolate(mode='nearest') restated: src = floor(dst*in/out) --
    network_generator.py:164,222."""
    N, C, H, W = x.shape
    Ho, Wo = size
    yi = torch.floor(torch.arange(Ho, dtype=torch.float32) * (H / Ho)).long().clamp(max=H - 1)
    xi = torch.floor(torch.arange(Wo, dtype=torch.float32) * (W / Wo)).long().clamp(max=W - 1)
    return x[:, :, yi][:, :, :, xi]


def instance_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.InstanceNorm2d(affine=False): biased var over H*W per (n,c) --
    network_generator.py:86,427; networks.py:440."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


# --------------------------------------------------------------------------
# tocg: ConditionGenerator (networks.py:13-198)
# --------------------------------------------------------------------------


# training mode of the tocg BatchNorms (train_condition.py:116 `tocg.train()`): batch statistics
# (biased variance over N*H*W) normalise; "stats" records (batch mean, UNBIASED batch var) per
# BatchNorm prefix = what nn.BatchNorm2d folds into running_mean / running_var with momentum 0.1.
BN_TRAIN = {"on": False, "stats": {}}


def _bn_eval(x: Tensor, sd: SD, p: str, eps: float = 1e-5) -> Tensor:
    if BN_TRAIN["on"]:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        m = x.numel() // x.shape[1]
        BN_TRAIN["stats"][p] = (mean.detach().clone(), (var * (m / max(m - 1, 1))).detach().clone())
        xh = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps)
        return xh * sd[p + ".weight"].view(1, -1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _tq(x: Tensor, w: Tensor, b, **kw) -> Tensor:
    """A tocg convolution.  With QUANT["fn"] set (emulation of the --fp16 / bf16-matrix-core engine, networks.py's
    mixed_precision mode of the HIP port) the input and the weight are rounded and the accumulation stays fp32 -- the
    engine's rounding points: ResBlock convolutions, the 1x1 laterals, the bottlenecks; the flow heads stay fp32."""
    q = globals()["QUANT"]["fn"]
    if q is not None:
        x, w = q(x), q(w)
    return F.conv2d(x, w, b, **kw)


def resblock(sd: SD, p: str, x: Tensor, scale: str) -> Tensor:
    """ResBlock.forward -- networks.py:171-198 (BatchNorm2d, eval mode)."""
    if scale == "down":
        r = _tq(x, sd[p + ".scale.weight"], None, stride=2, padding=1)
    elif scale == "same":
        r = _tq(x, sd[p + ".scale.weight"], sd[p + ".scale.bias"])
    elif globals()["QUANT"]["fn"] is not None:
        # the engine runs the 1x1 before the bilinear upsample (linear operators commute): its rounding point is the
        # low-resolution input
        r = resize_bilinear(_tq(x, sd[p + ".scale.1.weight"], None), scale_factor=2) + sd[p + ".scale.1.bias"].view(1, -1, 1, 1)
    else:  # up
        r = resize_bilinear(x, scale_factor=2)
        r = F.conv2d(r, sd[p + ".scale.1.weight"], sd[p + ".scale.1.bias"])
    t = _tq(r, sd[p + ".block.0.weight"], None, padding=1)
    t = F.relu(_bn_eval(t, sd, p + ".block.1"))
    t = _tq(t, sd[p + ".block.3.weight"], None, padding=1)
    t = _bn_eval(t, sd, p + ".block.4")
    return F.relu(r + t)


def tocg_forward(sd: SD, input1: Tensor, input2: Tensor,
                 warp_feature: str = "T1", out_layer: str = "relu", upsample: str = "bilinear"):
    """ConditionGenerator.forward -- networks.py:98-159.

    Returns (flow_list [5 x [N,h,w,2]], x [N,13,H,W], warped_c, warped_cm).
    """
    assert warp_feature in ("T1", "encoder") and out_layer in ("relu", "conv"), (warp_feature, out_layer)
    assert upsample in ("bilinear", "nearest"), upsample
    up2 = (lambda t: resize_bilinear(t, scale_factor=2)) if upsample == "bilinear" else \
        (lambda t: resize_nearest(t, (2 * t.shape[2], 2 * t.shape[3])))      # F.interpolate(t, scale_factor=2, mode=upsample): :130-133,150
    E1: List[Tensor] = []
    E2: List[Tensor] = []
    for i in range(5):
        E1.append(resblock(sd, f"ClothEncoder.{i}", input1 if i == 0 else E1[-1], "down"))
        E2.append(resblock(sd, f"PoseEncoder.{i}", input2 if i == 0 else E2[-1], "down"))
    flow_list: List[Tensor] = []
    T1 = T2 = x = None
    for i in range(5):
        N, _, iH, iW = E1[4 - i].shape
        grid = make_grid(N, iH, iW)
        if i == 0:
            T1, T2 = E1[4], E2[4]
            E4 = torch.cat([T1, T2], 1)
            flow = F.conv2d(E4, sd["flow_conv.0.weight"], sd["flow_conv.0.bias"], padding=1).permute(0, 2, 3, 1)
            flow_list.append(flow)
            x = resblock(sd, "conv", T2, "same")
            x = resblock(sd, "SegDecoder.0", x, "up")
        else:
            T1 = up2(T1) + _tq(
                E1[4 - i], sd[f"conv1.{4 - i}.weight"], sd[f"conv1.{4 - i}.bias"])
            T2 = up2(T2) + _tq(
                E2[4 - i], sd[f"conv2.{4 - i}.weight"], sd[f"conv2.{4 - i}.bias"])
            flow = up2(flow_list[i - 1].permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
            flow_norm = torch.cat([flow[..., 0:1] / ((iW / 2 - 1.0) / 2.0),
                                   flow[..., 1:2] / ((iH / 2 - 1.0) / 2.0)], 3)
            warped_T1 = grid_sample_bilinear_border(T1, flow_norm + grid)
            b = F.relu(_tq(x, sd[f"bottleneck.{i - 1}.0.weight"], sd[f"bottleneck.{i - 1}.0.bias"], padding=1))
            flow = flow + F.conv2d(torch.cat([warped_T1, b], 1), sd[f"flow_conv.{i}.weight"],
                                   sd[f"flow_conv.{i}.bias"], padding=1).permute(0, 2, 3, 1)
            flow_list.append(flow)
            if warp_feature == "T1":                                                   # networks.py:140-141
                x = resblock(sd, f"SegDecoder.{i}", torch.cat([x, E2[4 - i], warped_T1], 1), "up")
            else:                                                                      # 'encoder', networks.py:142-144
                warped_E1 = grid_sample_bilinear_border(E1[4 - i], flow_norm + grid)
                x = resblock(sd, f"SegDecoder.{i}", torch.cat([x, E2[4 - i], warped_E1], 1), "up")
    N, _, iH, iW = input1.shape
    grid = make_grid(N, iH, iW)
    flow = up2(flow_list[-1].permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    flow_norm = torch.cat([flow[..., 0:1] / ((iW / 2 - 1.0) / 2.0),
                           flow[..., 1:2] / ((iH / 2 - 1.0) / 2.0)], 3)
    warped_input1 = grid_sample_bilinear_border(input1, flow_norm + grid)
    if out_layer == "relu":                                                            # networks.py:54-55
        x = resblock(sd, "out_layer", torch.cat([x, input2, warped_input1], 1), "same")
    else:                                                                              # 'conv', networks.py:56-60
        x = resblock(sd, "out_layer.0", torch.cat([x, input2, warped_input1], 1), "same")
        x = _tq(x, sd["out_layer.1.weight"], sd["out_layer.1.bias"])
    return flow_list, x, warped_input1[:, :-1], warped_input1[:, -1:]


# --------------------------------------------------------------------------
# SPADE generator (network_generator.py:75-245)
# --------------------------------------------------------------------------


def spectral_sigma(w_orig: Tensor, u: Tensor, v: Tensor) -> Tensor:
    """Eval-mode spectral norm: sigma = u . (W_mat v) -- torch
    nn/utils/spectral_norm.py SpectralNorm.compute_weight (no power iteration
    when not training); applied at network_generator.py:138-143,409-410."""
    w_mat = w_orig.reshape(w_orig.shape[0], -1)
    return torch.dot(u, torch.mv(w_mat, v))


# Training-mode spectral norm (torch SpectralNorm.compute_weight with module.training): one power
# iteration v <- normalize(W^T u), u <- normalize(W v) (eps 1e-12) under no_grad, then
# sigma = u.(W v) differentiated through W only.  When SN_TRAIN["on"] is set the oracle applies
# it and records the updated (u, v) in SN_TRAIN["uv"] (the reference updates its buffers in place).
SN_TRAIN = {"on": False, "uv": {}}


def _sn_weight(sd: SD, p: str) -> Tensor:
    if p + ".weight_orig" in sd:
        w = sd[p + ".weight_orig"]
        u, v = sd[p + ".weight_u"], sd[p + ".weight_v"]
        if SN_TRAIN["on"]:
            with torch.no_grad():
                wm = w.reshape(w.shape[0], -1)
                v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
                u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
            SN_TRAIN["uv"][p] = (u.clone(), v.clone())
        return w / spectral_sigma(w, u, v)
    return sd[p + ".weight"]


# Emulation of the bf16 engine's rounding points: when QUANT["fn"] is set (e.g. lambda t:
# t.bfloat16().float()) every convolution of the generator rounds its input and weight with it and
# accumulates in fp32 -- exactly where the HIP bf16 path rounds (conv operands); InstanceNorm inputs,
# statistics, biases and the residual stream stay fp32.
QUANT = {"fn": None}


def _qconv(x: Tensor, w: Tensor, b, **kw) -> Tensor:
    q = QUANT["fn"]
    if q is not None:
        x, w = q(x), q(w)
    return F.conv2d(x, w, b, **kw)


def _qconv_sn(sd: SD, p: str, x: Tensor, b, **kw) -> Tensor:
    """Spectral-normed conv.  The bf16 engine rounds weight_orig and applies 1/sigma in the fp32 epilogue."""
    q = QUANT["fn"]
    if q is None or p + ".weight_orig" not in sd:
        return _qconv(x, _sn_weight(sd, p), b, **kw)
    w = sd[p + ".weight_orig"]
    u, v = sd[p + ".weight_u"], sd[p + ".weight_v"]
    if SN_TRAIN["on"]:                      # training mode: one power iteration first (as _sn_weight does)
        with torch.no_grad():
            wm = w.reshape(w.shape[0], -1)
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
        SN_TRAIN["uv"][p] = (u.clone(), v.clone())
    sigma = spectral_sigma(w, u, v)
    y = F.conv2d(q(x), q(w), None, **kw) / sigma
    return y if b is None else y + b.view(1, -1, 1, 1)


def spade_norm(sd: SD, p: str, x: Tensor, seg: Tensor, z: Optional[Tensor]) -> Tensor:
    """SPADENorm.forward -- network_generator.py:101-122.  ``z`` is the
    [b,w,h,1] standard-normal draw of :104-107 (None => zeros, which is what a
    zero ``noise_scale`` makes of it anyway)."""
    if z is not None:
        noise = (z * sd[p + ".noise_scale"]).transpose(1, 3)
        x = x + noise
    normalized = instance_norm(x)
    actv = F.relu(_qconv(seg, sd[p + ".conv_shared.0.weight"], sd[p + ".conv_shared.0.bias"], padding=1))
    gamma = _qconv(actv, sd[p + ".conv_gamma.weight"], sd[p + ".conv_gamma.bias"], padding=1)
    beta = _qconv(actv, sd[p + ".conv_beta.weight"], sd[p + ".conv_beta.bias"], padding=1)
    return normalized * (1 + gamma) + beta


def spade_resblock(sd: SD, p: str, x: Tensor, seg: Tensor, zs: Optional[Sequence[Tensor]]) -> Tensor:
    """SPADEResBlock.forward -- network_generator.py:163-173.  Noise draw
    order is norm_s, norm_0, norm_1 (:168-171)."""
    seg = resize_nearest(seg, tuple(x.shape[2:]))
    learned = (p + ".conv_s.weight_orig" in sd) or (p + ".conv_s.weight" in sd)
    zi = iter(zs) if zs is not None else None
    if learned:
        xs = spade_norm(sd, p + ".norm_s", x, seg, next(zi) if zi else None)
        x_s = _qconv_sn(sd, p + ".conv_s", xs, None)
    else:
        x_s = x
    dx = F.leaky_relu(spade_norm(sd, p + ".norm_0", x, seg, next(zi) if zi else None), 0.2)
    dx = _qconv_sn(sd, p + ".conv_0", dx, sd[p + ".conv_0.bias"], padding=1)
    dx = F.leaky_relu(spade_norm(sd, p + ".norm_1", dx, seg, next(zi) if zi else None), 0.2)
    dx = _qconv_sn(sd, p + ".conv_1", dx, sd[p + ".conv_1.bias"], padding=1)
    return x_s + dx


def spade_generator_forward(sd: SD, x: Tensor, seg: Tensor, fine_height: int, fine_width: int,
                            num_upsampling_layers: str = "most",
                            noise: Optional[Dict[str, Sequence[Tensor]]] = None) -> Tensor:
    """SPADEGenerator.forward -- network_generator.py:221-245 (eval mode)."""
    nup = {"normal": 5, "more": 6, "most": 7}[num_upsampling_layers]
    sh, sw = fine_height // 2 ** nup, fine_width // 2 ** nup
    samples = [resize_nearest(x, (sh * 2 ** i, sw * 2 ** i)) for i in range(8)]
    feats = [_qconv(samples[i], sd[f"conv_{i}.weight"], sd[f"conv_{i}.bias"], padding=1) for i in range(8)]

    def up(t: Tensor) -> Tensor:
        return t.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)

    def blk(name: str, t: Tensor) -> Tensor:
        return spade_resblock(sd, name, t, seg, None if noise is None else noise[name])

    h = blk("head_0", feats[0])
    h = up(h)
    h = blk("G_middle_0", torch.cat((h, feats[1]), 1))
    if num_upsampling_layers in ("more", "most"):
        h = up(h)
    h = blk("G_middle_1", torch.cat((h, feats[2]), 1))
    for j, name in enumerate(["up_0", "up_1", "up_2", "up_3"]):
        h = up(h)
        h = blk(name, torch.cat((h, feats[3 + j]), 1))
    if num_upsampling_layers == "most":
        h = up(h)
        h = blk("up_4", torch.cat((h, feats[7]), 1))
    h = _qconv(F.leaky_relu(h, 0.2), sd["conv_img.weight"], sd["conv_img.bias"], padding=1)
    return torch.tanh(h)


# --------------------------------------------------------------------------
# Generator's multi-scale PatchGAN (network_generator.py:250-316)
# --------------------------------------------------------------------------


def gen_discriminator_forward(sd: SD, inp: Tensor, num_D: int = 2, n_layers_D: int = 3) -> List[List[Tensor]]:
    """MultiscaleDiscriminator.forward with intermediate features --
    network_generator.py:306-316 (+ NLayerDiscriminator.forward :278-288)."""
    out: List[List[Tensor]] = []
    for d in range(num_D):
        p = f"discriminator_{d}"
        feats = []
        # (_qconv / _qconv_sn: plain F.conv2d unless QUANT["fn"] emulates the bf16 engine's operand rounding)
        h = F.leaky_relu(_qconv(inp, sd[p + ".model0.0.weight"], sd[p + ".model0.0.bias"], stride=2, padding=2), 0.2)
        feats.append(h)
        for n in range(1, n_layers_D):
            h = _qconv_sn(sd, f"{p}.model{n}.0.0", h, None, stride=2, padding=2)
            h = F.leaky_relu(instance_norm(h), 0.2)
            feats.append(h)
        h = _qconv(h, sd[f"{p}.model{n_layers_D}.0.weight"], sd[f"{p}.model{n_layers_D}.0.bias"], stride=1, padding=2)
        feats.append(h)
        out.append(feats)
        inp = F.avg_pool2d(inp, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
    return out


# --------------------------------------------------------------------------
# Parse glue (test_generator.py:161-217)
# --------------------------------------------------------------------------


def gaussian_kernel1d(ksize: int, sigma: float) -> Tensor:
    """torchgeometry 0.1.2 image/gaussian.py ``gaussian``: exp(-(x-k//2)^2 /
    (2 sigma^2)) normalised to sum 1.  PARITY UNPINNED (third-party, absent)."""
    xs = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    g = torch.exp(-(xs ** 2) / float(2 * sigma ** 2))
    return g / g.sum()


def gaussian_blur(x: Tensor, ksize: Tuple[int, int] = (15, 15), sigma: Tuple[float, float] = (3.0, 3.0)) -> Tensor:
    """tgm.image.GaussianBlur((15,15),(3,3)) -- depthwise conv2d with the outer
    product kernel and zero padding (k-1)//2; call sites test_generator.py:91,179.
    PARITY UNPINNED (see module docstring)."""
    kx = gaussian_kernel1d(ksize[0], sigma[0])
    ky = gaussian_kernel1d(ksize[1], sigma[1])
    k2 = torch.matmul(kx.unsqueeze(-1), ky.unsqueeze(-1).t())
    C = x.shape[1]
    w = k2.view(1, 1, *k2.shape).repeat(C, 1, 1, 1)
    return F.conv2d(x, w, padding=((ksize[0] - 1) // 2, (ksize[1] - 1) // 2), groups=C)


PARSE_MERGE = {0: [0], 1: [2, 4, 7, 8, 9, 10, 11], 2: [3], 3: [1], 4: [5], 5: [6], 6: [12]}


def remove_overlap(seg_out: Tensor, warped_cm: Tensor) -> Tensor:
    """test_generator.py:19-24."""
    s = torch.cat([seg_out[:, 1:3], seg_out[:, 5:]], dim=1).sum(dim=1, keepdim=True)
    return warped_cm - s * warped_cm


def parse_glue(fake_segmap: Tensor, warped_cm: Tensor, fine_height: int, fine_width: int,
               composition: str = "warp_grad"):
    """test_generator.py:167-203: cloth-mask composition, bilinear up to fine
    size, 15x15 Gaussian, argmax, one-hot(13), 13->7 merge.
    Returns (fake_parse_gauss [N,13,H,W], labels int64 [N,H,W], parse7 [N,7,H,W])."""
    if composition != "no_composition":
        cm = warped_cm if composition == "warp_grad" else (warped_cm > 0.5).to(fake_segmap.dtype)
        mask = torch.ones_like(fake_segmap)
        mask[:, 3:4] = cm
        fake_segmap = fake_segmap * mask
    g = gaussian_blur(resize_bilinear(fake_segmap, size=(fine_height, fine_width)))
    lab = g.argmax(dim=1)
    old = torch.zeros(lab.shape[0], 13, fine_height, fine_width)
    old.scatter_(1, lab[:, None], 1.0)
    parse = torch.zeros(lab.shape[0], 7, fine_height, fine_width)
    for i, src in PARSE_MERGE.items():
        for l in src:
            parse[:, i] += old[:, l]
    return g, lab, parse


def hires_warp(flow_last: Tensor, clothes: Tensor, cloth_mask: Tensor):
    """test_generator.py:206-213: upsample the last flow to the cloth size,
    normalise by the hard-coded ((96-1)/2, (128-1)/2), add the base grid and
    warp cloth + mask."""
    N, _, iH, iW = clothes.shape
    flow = resize_bilinear(flow_last.permute(0, 3, 1, 2), size=(iH, iW)).permute(0, 2, 3, 1)
    flow_norm = torch.cat([flow[..., 0:1] / ((96 - 1.0) / 2.0), flow[..., 1:2] / ((128 - 1.0) / 2.0)], 3)
    wg = make_grid(N, iH, iW) + flow_norm
    return grid_sample_bilinear_border(clothes, wg), grid_sample_bilinear_border(cloth_mask, wg)


# --------------------------------------------------------------------------
# Losses of the generator training step (train_generator.py:279-360)
# --------------------------------------------------------------------------


def hinge_loss(preds: List[List[Tensor]], target_is_real: bool, for_discriminator: bool) -> Tensor:
    """GANLoss('hinge').__call__ on a list (scales) of lists (layers) -- network_generator.py:365-398:
    the last tensor of each scale, averaged over scales."""
    total = 0
    for p in preds:
        x = p[-1]
        if for_discriminator:
            l = -torch.mean(torch.min(x - 1, torch.zeros_like(x))) if target_is_real else \
                -torch.mean(torch.min(-x - 1, torch.zeros_like(x)))
        else:
            l = -torch.mean(x)
        total = total + l
    return total / len(preds)


def feat_match_loss(pred_fake: List[List[Tensor]], pred_real: List[List[Tensor]], lambda_feat: float = 10.0) -> Tensor:
    """train_generator.py:300-309."""
    num_D = len(pred_fake)
    total = 0
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            total = total + F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) * lambda_feat / num_D
    return total


def split_fake_real(pred: List[List[Tensor]]):
    """train_generator.py:287-295: first half of the batch is fake, second half real."""
    fake = [[t[: t.size(0) // 2] for t in p] for p in pred]
    real = [[t[t.size(0) // 2:] for t in p] for p in pred]
    return fake, real


VGG_CONVS = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28]
VGG_POOLS = [4, 9, 18, 27]
VGG_TAPS = [0, 5, 10, 19, 28]


def vgg19_features(sd: SD, x: Tensor) -> List[Tensor]:
    """Vgg19.forward -- networks.py:201-232: torchvision vgg19 ``features[0:30]`` tapped after
    relu1_1, 2_1, 3_1, 4_1, 5_1.  ``sd`` uses the reference module's keys (sliceK.IDX.weight)."""
    def key(i):
        k = 1 + sum(i >= b for _, b in [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)])
        return f"slice{k}.{i}"
    out = []
    h = x
    for i in VGG_CONVS:
        if (i - 1) in VGG_POOLS:
            h = F.max_pool2d(h, 2, 2)
        h = F.relu(F.conv2d(h, sd[key(i) + ".weight"], sd[key(i) + ".bias"], padding=1))
        if i in VGG_TAPS:
            out.append(h)
    return out


def vgg_loss(sd: SD, x: Tensor, y: Tensor) -> Tensor:
    """VGGLoss.forward -- networks.py:244-251."""
    w = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
    fx, fy = vgg19_features(sd, x), vgg19_features(sd, y)
    return sum(w[i] * F.l1_loss(fx[i], fy[i].detach()) for i in range(5))


# --------------------------------------------------------------------------
# tocg training step (train_condition.py:113-286; config: --Ddownx2 --lasttvonly --interflowloss)
# --------------------------------------------------------------------------


def avgpool3x3s2(x: Tensor) -> Tensor:
    """nn.AvgPool2d(3, stride=2, padding=[1,1], count_include_pad=False) -- networks.py:322."""
    return F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)


def tocg_discriminator_forward(sd: SD, inp: Tensor, num_D: int = 2, n_layers: int = 3,
                               Ddownx2: bool = False, drop_masks: Optional[List[Tensor]] = None) -> List[List[Tensor]]:
    """networks.MultiscaleDiscriminator.forward (getIntermFeat=False, InstanceNorm2d(affine=False), no
    spectral norm, no dropout) -- networks.py:302-408.  Flattened nn.Sequential per scale `layer{k}`:
    [conv4x4 s2, LReLU] + (n_layers-1) x [conv4x4 s2, IN, LReLU] + [conv4x4 s1, IN, LReLU] + [conv4x4 s1 -> 1],
    every conv padded by ceil(3/2) = 2.  Scale i of the result uses `layer{num_D-1-i}` on the input
    avg-pooled i (+1 with Ddownx2) times."""
    result = []
    x = avgpool3x3s2(inp) if Ddownx2 else inp
    for i in range(num_D):
        p = f"layer{num_D - 1 - i}"
        h = F.leaky_relu(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], stride=2, padding=2), 0.2)
        idx = 2
        for n in range(1, n_layers):
            w = _sn_weight(sd, f"{p}.{idx}") if f"{p}.{idx}.weight_orig" in sd else sd[f"{p}.{idx}.weight"]
            h = F.conv2d(h, w, sd[f"{p}.{idx}.bias"], stride=2, padding=2)
            h = F.leaky_relu(instance_norm(h), 0.2)
            idx += 3
            if drop_masks is not None:     # --Ddropout: nn.Dropout(0.5) after the activation (networks.py:363-368);
                # the keep masks (already x 1/(1-p)) are supplied: a list consumed in call order, or a callable shape -> mask
                h = h * (drop_masks(tuple(h.shape)) if callable(drop_masks) else drop_masks.pop(0))
                idx += 1
        h = F.conv2d(h, sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"], stride=1, padding=2)
        h = F.leaky_relu(instance_norm(h), 0.2)
        idx += 3
        h = F.conv2d(h, sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"], stride=1, padding=2)
        result.append([h])
        if i != num_D - 1:
            x = avgpool3x3s2(x)
    return result


def lsgan_loss(preds: List[List[Tensor]], target_is_real: bool) -> Tensor:
    """networks.GANLoss(use_lsgan=True).__call__ on a list of lists -- networks.py:258-299:
    MSE of the last tensor of every scale against a constant 1 / 0 map, SUMMED over scales."""
    total = 0
    for p in preds:
        x = p[-1]
        total = total + F.mse_loss(x, torch.full_like(x, 1.0 if target_is_real else 0.0))
    return total


def cross_entropy2d(inp: Tensor, target: Tensor) -> Tensor:
    """utils.py:29-42 with matching sizes: mean softmax cross entropy over N*H*W pixels
    (ignore_index=250 never occurs: labels are 0..12)."""
    n, c, h, w = inp.shape
    return F.cross_entropy(inp.permute(0, 2, 3, 1).reshape(-1, c), target.reshape(-1), ignore_index=250)


def tv_loss(flow: Tensor) -> Tensor:
    """train_condition.py:190-199 on one [N,h,w,2] flow: mean |d/dy| + mean |d/dx|."""
    return (flow[:, 1:] - flow[:, :-1]).abs().mean() + (flow[:, :, 1:] - flow[:, :, :-1]).abs().mean()


def condition_train_losses(sd_g: SD, sd_d: SD, sd_vgg: Optional[SD], batch: Dict[str, Tensor],
                           lasttvonly: bool = True, interflowloss: bool = True, occlusion: bool = False,
                           Ddownx2: bool = True, composition: str = "warp_grad", tvlambda: float = 2.0,
                           CElamda: float = 10.0, GANlambda: float = 1.0, num_D: int = 2, drop_masks=None,
                           edgeawaretv: str = "no_edge", add_lasttv: bool = False, warp_feature: str = "T1", out_layer: str = "relu",
                           upsample: str = "bilinear"):
    """One iteration of train_condition.py:136-277 up to the two loss sums (the caller runs backward).
    `batch`: cloth, cloth_mask (already binarised :140), parse_agnostic, densepose, parse_onehot (label
    indices [N,1,H,W]), parse (one-hot 13), pcm, parse_cloth.  sd_vgg None drops the VGG terms
    (torchvision weights are unavailable offline).  Returns a dict of the loss terms and tocg outputs."""
    c_paired, cm_paired = batch["cloth"], batch["cloth_mask"]
    input1 = torch.cat([c_paired, cm_paired], 1)
    input2 = torch.cat([batch["parse_agnostic"], batch["densepose"]], 1)
    label_onehot, label, pcm, im_c = batch["parse_onehot"], batch["parse"], batch["pcm"], batch["parse_cloth"]
    BN_TRAIN["on"], BN_TRAIN["stats"] = True, {}
    try:
        flow_list, fake_segmap, warped_c, warped_cm = tocg_forward(sd_g, input1, input2, warp_feature, out_layer)
    finally:
        BN_TRAIN["on"] = False
    seg_raw = fake_segmap
    warped_cm_onehot = (warped_cm.detach() > 0.5).float()
    if composition != "no_composition":                                     # :164-173
        mask = torch.ones_like(fake_segmap.detach())
        mask[:, 3:4] = warped_cm_onehot if composition == "detach" else warped_cm
        fake_segmap = fake_segmap * mask
    if occlusion:                                                           # :174-176
        warped_cm = remove_overlap(F.softmax(fake_segmap, dim=1), warped_cm)
        warped_c = warped_c * warped_cm + torch.ones_like(warped_c) * (1 - warped_cm)
    vgg = (lambda a, b: vgg_loss(sd_vgg, a, b)) if sd_vgg is not None else (lambda a, b: torch.zeros(()))
    loss_l1 = F.l1_loss(warped_cm, pcm)                                     # :184
    loss_vgg = vgg(warped_c, im_c)                                          # :185
    loss_tv = 0
    if edgeawaretv == "no_edge":
        for flow in (flow_list[-1:] if lasttvonly else flow_list):          # :190-199
            loss_tv = loss_tv + tv_loss(flow)
    else:                                                                   # :200-229
        for i in ([4] if edgeawaretv == "last_only" else range(5)):
            flow = flow_list[i]
            wcd = resize_bilinear(warped_cm, size=tuple(flow.shape[1:3])).permute(0, 2, 3, 1)
            y_tv = (flow[:, 1:] - flow[:, :-1]).abs() * torch.exp(-150 * (wcd[:, 1:] - wcd[:, :-1]).abs())
            x_tv = (flow[:, :, 1:] - flow[:, :, :-1]).abs() * torch.exp(-150 * (wcd[:, :, 1:] - wcd[:, :, :-1]).abs())
            sc = 1.0 if edgeawaretv == "last_only" else 1.0 / (2 ** (4 - i))
            loss_tv = loss_tv + y_tv.mean() * sc + x_tv.mean() * sc
        if add_lasttv:
            loss_tv = loss_tv + tv_loss(flow_list[-1])
    N, _, iH, iW = c_paired.shape
    if interflowloss:                                                       # :235-248
        for i in range(len(flow_list) - 1):
            flow = flow_list[i]
            _, fH, fW, _ = flow.shape
            grid = make_grid(N, iH, iW)
            flow = (resize_bilinear if upsample == "bilinear" else resize_nearest)(flow.permute(0, 3, 1, 2), size=(iH, iW)).permute(0, 2, 3, 1)   # :242
            flow_norm = torch.cat([flow[..., 0:1] / ((fW - 1.0) / 2.0), flow[..., 1:2] / ((fH - 1.0) / 2.0)], 3)
            w_c = grid_sample_bilinear_border(c_paired, flow_norm + grid)
            w_cm = grid_sample_bilinear_border(cm_paired, flow_norm + grid)
            w_cm = remove_overlap(F.softmax(fake_segmap, dim=1), w_cm)
            loss_l1 = loss_l1 + F.l1_loss(w_cm, pcm) / (2 ** (4 - i))
            loss_vgg = loss_vgg + vgg(w_c, im_c) / (2 ** (4 - i))
    ce = cross_entropy2d(fake_segmap, label_onehot.transpose(0, 1)[0].long())   # :252
    soft = torch.softmax(fake_segmap, 1)                                    # :260
    dm = drop_masks or {}
    pred = tocg_discriminator_forward(sd_d, torch.cat((input1.detach(), input2.detach(), soft), 1), num_D, 3, Ddownx2,
                                      dm.get("g"))
    loss_g_gan = lsgan_loss(pred, True)                                     # :264
    pred_f = tocg_discriminator_forward(sd_d, torch.cat((input1.detach(), input2.detach(), soft.detach()), 1), num_D,
                                        3, Ddownx2, dm.get("f"))
    pred_r = tocg_discriminator_forward(sd_d, torch.cat((input1.detach(), input2.detach(), label), 1), num_D, 3, Ddownx2,
                                        dm.get("r"))
    loss_d_fake, loss_d_real = lsgan_loss(pred_f, False), lsgan_loss(pred_r, True)
    loss_G = (10 * loss_l1 + loss_vgg + tvlambda * loss_tv) + (ce * CElamda + loss_g_gan * GANlambda)   # :276
    loss_D = loss_d_fake + loss_d_real                                      # :277
    return {"loss_G": loss_G, "loss_D": loss_D, "l1": loss_l1, "vgg": loss_vgg, "tv": loss_tv, "ce": ce,
            "g_gan": loss_g_gan, "d_fake": loss_d_fake, "d_real": loss_d_real, "flow_list": flow_list,
            "fake_segmap": seg_raw, "warped_c": warped_c, "warped_cm": warped_cm, "bn_stats": dict(BN_TRAIN["stats"])}


def d_logit(pred: List[List[Tensor]]) -> Tensor:
    """D_logit -- get_norm_const.py:60-64 (identical in test_condition.py)."""
    score = 0
    for p in pred:
        score = score + p[-1].mean((1, 2, 3)) / 2
    return score


def rejection_logits(sd_g: SD, sd_d: SD, batch: Dict[str, Tensor], composition: str = "warp_grad",
                     Ddownx2: bool = True, num_D: int = 2):
    """get_norm_const.py:86-118 / test_condition.py:98-121 with tocg and D in eval mode:
    (logit_real, logit_fake, composed fake_segmap)."""
    input1 = torch.cat([batch["cloth"], batch["cloth_mask"]], 1)
    input2 = torch.cat([batch["parse_agnostic"], batch["densepose"]], 1)
    flow_list, fake_segmap, warped_c, warped_cm = tocg_forward(sd_g, input1, input2)
    if composition != "no_composition":
        mask = torch.ones_like(fake_segmap)
        mask[:, 3:4] = (warped_cm > 0.5).float() if composition == "detach" else warped_cm
        fake_segmap = fake_segmap * mask
    soft = F.softmax(fake_segmap, dim=1)
    real = tocg_discriminator_forward(sd_d, torch.cat((input1, input2, batch["parse"]), 1), num_D, 3, Ddownx2)
    fake = tocg_discriminator_forward(sd_d, torch.cat((input1, input2, soft), 1), num_D, 3, Ddownx2)
    return d_logit(real), d_logit(fake), fake_segmap
