"""GPU: training-side conv kernels vs torch autograd on the CPU (oracle by execution):
device weight packing, data gradient (stride 1 / stride-2 phases, fused activation
derivative), weight gradient (multi-source, folded nearest up-sampling), bias gradient."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mods():
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops
    return ops, train_ops


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


CASES = [
    # cins, cout, k, stride, pad, H, W
    ([32], 48, 3, 1, 1, 12, 10),
    ([96, 16, 4], 13, 1, 1, 0, 9, 7),
    ([10], 64, 4, 2, 2, 18, 14),     # PatchGAN 4x4 s2 p2 (odd output extents)
    ([64], 128, 4, 2, 2, 9, 7),
    ([16], 96, 3, 2, 1, 16, 12),     # tocg 'down' 3x3 s2 p1
    ([128], 160, 3, 1, 1, 8, 6),
    ([8], 128, 3, 1, 1, 16, 12),     # conv_shared-like: tiny Cin
    ([256], 1, 4, 1, 2, 10, 8),      # PatchGAN head: Cout = 1
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv_forward_dgrad_wgrad_vs_autograd(case):
    ops, T = _mods()
    cins, cout, k, stride, pad, H, W = case
    g = torch.Generator().manual_seed(sum(cins) + cout + k)
    N = 2
    xs = [torch.randn(N, c, H, W, generator=g, requires_grad=True) for c in cins]
    w = (torch.randn(cout, sum(cins), k, k, generator=g) * (1.0 / (sum(cins) * k * k) ** 0.5)).requires_grad_()
    b = torch.randn(cout, generator=g, requires_grad=True)
    wscale = 0.7
    y = F.conv2d(torch.cat(xs, 1), w * wscale, b, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    # ---- HIP
    wd = w.detach().cuda()
    acts = [ops.to_nhwc(x.detach().cuda()) for x in xs]
    yo = T.conv_forward_dev(wd, [(a, 0) for a in acts], stride, pad, wscale=wscale, shift=b.detach().cuda())
    assert _rel(ops.to_nchw(yo), y) < 2e-5
    dya = ops.to_nhwc(dy.cuda())
    dx = T.conv_dgrad(dya, wd, H, W, stride, pad, wscale=wscale)
    want_dx = torch.cat([x.grad for x in xs], 1)
    assert _rel(ops.to_nchw(dx), want_dx) < 3e-5, _rel(ops.to_nchw(dx), want_dx)
    dw = torch.full(w.shape, float("nan"), device="cuda")
    base = 0
    for a, c in zip(acts, cins):
        T.conv_wgrad(dya, a, 0, base, sum(cins), k, k, stride, pad, dw)
        base += c
    assert _rel(dw, w.grad / wscale) < 5e-5, _rel(dw, w.grad / wscale)   # dW wrt (w*wscale)
    db = T.colsum(dya)
    assert _rel(db, b.grad) < 2e-5


def test_pack_dev_equals_host_pack():
    ops, T = _mods()
    g = torch.Generator().manual_seed(0)
    w = torch.randn(40, 96 + 16 + 4, 3, 3, generator=g)
    layer = ops.ConvLayer(w, [96, 16, 4], "cuda", name="x")
    for cfg in (0, 1, 5, 6):
        host = layer._get_packed(cfg)
        dev, geom = T.pack_weight_dev(w.cuda(), [96, 16, 4], [96, 16, 4], cfg, 0, 1, 1)
        assert dev.numel() == host.numel() == geom[7]
        assert torch.equal(dev, host)


def test_dgrad_fused_activation_derivative_and_accumulate():
    ops, T = _mods()
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 2, 32, 10, 8
    pre = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    x = F.leaky_relu(pre, 0.2)
    w = torch.randn(24, C, 3, 3, generator=g) * 0.1
    y = F.conv2d(x, w, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xa = ops.to_nhwc(x.detach().cuda())
    dpre = T.conv_dgrad(ops.to_nhwc(dy.cuda()), w.cuda(), H, W, 1, 1, act_mask=xa, slope=0.2)
    assert _rel(ops.to_nchw(dpre), pre.grad) < 3e-5
    # accumulate=True adds into an existing weight gradient
    dw = torch.ones(24, C, 3, 3, device="cuda")
    T.conv_wgrad(ops.to_nhwc(dy.cuda()), xa, 0, 0, C, 3, 3, 1, 1, dw, accumulate=True)
    xd = x.detach().requires_grad_(False)
    wg = torch.autograd.grad(F.conv2d(xd, w.requires_grad_(), padding=1), w, dy)[0]
    assert _rel(dw - 1.0, wg) < 5e-5


def test_wgrad_nearest_upsampled_source():
    """cat([up(x_prev), feat]) -> conv: the weight gradient reads x_prev through the folded upsample."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(6)
    N, H, W = 2, 12, 8
    xp = torch.randn(N, 32, H // 2, W // 2, generator=g)
    ft = torch.randn(N, 16, H, W, generator=g)
    w = (torch.randn(40, 48, 3, 3, generator=g) * 0.1).requires_grad_()
    xin = torch.cat([xp.repeat_interleave(2, 2).repeat_interleave(2, 3), ft], 1)
    y = F.conv2d(xin, w, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw = torch.empty(40, 48, 3, 3, device="cuda")
    dya = ops.to_nhwc(dy.cuda())
    T.conv_wgrad(dya, ops.to_nhwc(xp.cuda()), 1, 0, 48, 3, 3, 1, 1, dw)
    T.conv_wgrad(dya, ops.to_nhwc(ft.cuda()), 0, 32, 48, 3, 3, 1, 1, dw)
    assert _rel(dw, w.grad) < 5e-5
