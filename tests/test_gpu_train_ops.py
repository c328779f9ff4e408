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


@pytest.mark.parametrize("C,H,W,noise,spade", [(32, 12, 10, True, True), (80, 9, 7, False, True), (64, 17, 13, False, False),
                                               (20, 33, 5, True, True)])
def test_spade_norm_backward_vs_autograd(C, H, W, noise, spade):
    ops, T = _mods()
    g = torch.Generator().manual_seed(C + H)
    N = 2
    x = (torch.randn(N, C, H, W, generator=g) * 2 + 1).requires_grad_()
    ns = (torch.randn(C, generator=g) * 0.5).requires_grad_() if noise else None
    z = torch.randn(N, W, H, 1, generator=g) if noise else None
    gamma = torch.randn(N, C, H, W, generator=g, requires_grad=True) if spade else None
    beta = torch.randn(N, C, H, W, generator=g, requires_grad=True) if spade else None
    v = x + ((z * ns).transpose(1, 3) if noise else 0)
    mean = v.mean(dim=(2, 3), keepdim=True)
    var = v.var(dim=(2, 3), unbiased=False, keepdim=True)
    nh = (v - mean) / torch.sqrt(var + 1e-5)
    pre = nh * (1 + gamma) + beta if spade else nh
    out = F.leaky_relu(pre, 0.2)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    # ---- HIP
    xa = ops.to_nhwc(x.detach().cuda())
    zc = z.cuda().contiguous() if noise else None
    nsc = ns.detach().cuda() if noise else None
    mu, rs = ops.instnorm_stats(xa, zc, nsc)
    dns = torch.zeros(C, device="cuda") if noise else None
    dx, dgb = T.norm_bwd(xa, mu, rs, ops.to_nhwc(dout.cuda()), act=ops.ACT_LRELU, slope=0.2,
                         out=ops.to_nhwc(out.detach().cuda()),
                         g1p=ops.to_nhwc((1 + gamma).detach().cuda()) if spade else None, z=zc, noise_scale=nsc,
                         want_dgb=spade, dnoise_scale=dns)
    assert _rel(ops.to_nchw(dx), x.grad) < 5e-5, _rel(ops.to_nchw(dx), x.grad)
    if spade:
        assert _rel(ops.to_nchw(dgb, 0, C), gamma.grad) < 2e-5
        assert _rel(ops.to_nchw(dgb, C, C), beta.grad) < 2e-5
    if noise:
        assert _rel(dns, ns.grad) < 5e-5


def test_losses_value_and_gradient():
    ops, T = _mods()
    g = torch.Generator().manual_seed(1)
    a = torch.randn(3, 5, 17, 11, generator=g, requires_grad=True)
    b = torch.randn(3, 5, 17, 11, generator=g)
    n = a.numel()
    cases = [(T.LOSS_L1, lambda: (a - b).abs().mean()), (T.LOSS_HINGE_D_FAKE, lambda: -torch.min(-a - 1, torch.zeros(1)).mean()),
             (T.LOSS_HINGE_D_REAL, lambda: -torch.min(a - 1, torch.zeros(1)).mean()), (T.LOSS_NEG_MEAN, lambda: -a.mean()),
             (T.LOSS_MSE, lambda: ((a - b) ** 2).mean())]
    for mode, fn in cases:
        a.grad = None
        want = fn()
        want.backward()
        lo = torch.zeros(1, device="cuda")
        grad = T.loss(a.detach().cuda().contiguous(), b.cuda().contiguous(), mode, 1.0 / n, 1.0 / n, lo, accumulate=False)
        assert abs(lo.item() - want.item()) < 1e-5 * max(1.0, abs(want.item())), (mode, lo.item(), want.item())
        assert _rel(grad, a.grad) < 1e-6, mode


def test_downsum_avgpool_maxpool_backward():
    ops, T = _mods()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 8, 6, 5, generator=g, requires_grad=True)
    up = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
    d = torch.randn(up.shape, generator=g)
    up.backward(d)
    assert _rel(ops.to_nchw(T.downsum2x2(ops.to_nhwc(d.cuda()))), x.grad) < 1e-6
    for shape in [(2, 12, 9, 7), (1, 8, 16, 12), (2, 4, 5, 5)]:
        x = torch.randn(*shape, generator=g, requires_grad=True)
        y = F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
        d = torch.randn(y.shape, generator=g)
        y.backward(d)
        got = T.avgpool3x3s2_bwd(ops.to_nhwc(d.cuda()), shape[2], shape[3])
        assert _rel(ops.to_nchw(got), x.grad) < 1e-6
    x = torch.randn(2, 16, 12, 8, generator=g, requires_grad=True)
    y = F.max_pool2d(x, 2, 2)
    d = torch.randn(y.shape, generator=g)
    y.backward(d)
    xa = ops.to_nhwc(x.detach().cuda())
    assert torch.equal(ops.to_nchw(T.maxpool2x2(xa)).cpu(), y.detach())
    assert torch.equal(ops.to_nchw(T.maxpool2x2_bwd(xa, ops.to_nhwc(d.cuda()))).cpu(), x.grad)


def test_adam_matches_torch():
    ops, T = _mods()
    g = torch.Generator().manual_seed(3)
    w = torch.randn(1000, generator=g)
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.Adam([p], lr=2e-4, betas=(0.0, 0.9))
    wd, m, v = w.clone().cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        p.grad = gr.clone()
        opt.step()
        T.adam_step(wd, gr.cuda(), m, v, 2e-4, 0.0, 0.9, 1e-8, 0.0, step)
        assert _rel(wd, p.detach()) < 1e-6


def test_spectral_norm_power_iteration_and_gradient():
    ops, T = _mods()
    from torch.nn.utils import spectral_norm
    torch.manual_seed(4)
    conv = spectral_norm(torch.nn.Conv2d(20, 24, 3, padding=1))
    w0 = conv.weight_orig.detach().clone()
    u0, v0 = conv.weight_u.clone(), conv.weight_v.clone()
    x = torch.randn(2, 20, 8, 6)
    conv.train()
    y = conv(x)                      # one power iteration (in place on u, v), then W/sigma
    dy = torch.randn_like(y)
    y.backward(dy)
    # ---- HIP: same iteration from the same starting u, v
    wc, uc, vc = w0.cuda(), u0.cuda(), v0.cuda()
    sigma = T.spectral_sigma(wc, uc, vc, 1)
    assert _rel(uc, conv.weight_u) < 1e-5 and _rel(vc, conv.weight_v) < 1e-5
    w_sn = w0 / sigma.cpu()
    assert _rel(w_sn, conv.weight.detach()) < 1e-5
    # G = dL/dW_sn from our wgrad; transform to dW_orig
    G = torch.empty_like(wc)
    T.conv_wgrad(ops.to_nhwc(dy.cuda()), ops.to_nhwc(x.cuda()), 0, 0, 20, 3, 3, 1, 1, G)
    dwo = torch.empty_like(wc)
    T.spectral_grad(G, wc, uc, vc, sigma, dwo)
    assert _rel(dwo, conv.weight_orig.grad) < 1e-4, _rel(dwo, conv.weight_orig.grad)


def test_training_state_checkpoint_resumes_exactly(tmp_path):
    """save_training_state / load_training_state (weights + Adam moments + scheduler + step): a run resumed from the
    file takes bit-identical further steps; the optimizer state_dict has torch.optim.Adam's layout."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.checkpoint import load_training_state, save_training_state
    from hr_viton_amd.optim import Adam

    def make():
        torch.manual_seed(4)
        m = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3), torch.nn.Conv2d(8, 5, 1)).cuda()
        o = Adam(m.parameters(), lr=1e-2, betas=(0.5, 0.999))
        s = torch.optim.lr_scheduler.LambdaLR(o, lr_lambda=lambda e: 1.0 / (1 + e))
        return m, o, s

    def step(m, o, s, k):
        g = torch.Generator(device="cuda").manual_seed(100 + k)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=g, device="cuda")
        o.step()
        s.step()

    m1, o1, s1 = make()
    for k in range(3):
        step(m1, o1, s1, k)
    path = str(tmp_path / "state.pth")
    save_training_state(path, {"net": m1}, {"opt": o1}, {"sched": s1}, step=3, extra={"note": "x"})
    sd = o1.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][0]["exp_avg"].shape == (8, 4, 3, 3)
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m1.parameters()], lr=1e-2, betas=(0.5, 0.999))
    ref.load_state_dict({"state": {k: {n: (v.clone() if torch.is_tensor(v) else v) for n, v in e.items()}
                                   for k, e in sd["state"].items()}, "param_groups": sd["param_groups"]})
    for k in range(3, 5):
        step(m1, o1, s1, k)
    m2, o2, s2 = make()
    got_step, extra = load_training_state(path, {"net": m2}, {"opt": o2}, {"sched": s2})
    assert got_step == 3 and extra == {"note": "x"}
    for k in range(3, 5):
        step(m2, o2, s2, k)
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_device_step_adam_resumes_from_the_loaded_step_count():
    """optim.Adam(device_step=True) keeps its step count / learning rate on the device (hipGraph iterations).  Loading a
    state dict into an optimizer that has ALREADY stepped must move the device count to the loaded one: the next steps
    are bit-identical to an optimizer that took the same steps without interruption."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.ops import HrvError
    from hr_viton_amd.optim import Adam

    def make():
        torch.manual_seed(9)
        m = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3), torch.nn.Conv2d(8, 5, 1)).cuda()
        return m, Adam(m.parameters(), lr=1e-2, betas=(0.5, 0.999), device_step=True)

    def step(m, o, k):
        g = torch.Generator(device="cuda").manual_seed(200 + k)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, generator=g, device="cuda")
        o.step()

    m1, o1 = make()
    for k in range(2):
        step(m1, o1, k)
    sd = o1.state_dict()
    w2 = [p.detach().clone() for p in m1.parameters()]
    assert int(sd["state"][0]["step"]) == 2
    for k in range(2, 4):
        step(m1, o1, k)
    # a second optimizer that has gone further (7 steps) is wound back to the 2-step state
    m2, o2 = make()
    for k in range(7):
        step(m2, o2, 50 + k)
    with torch.no_grad():
        for p, w in zip(m2.parameters(), w2):
            p.copy_(w)
    o2.load_state_dict(sd)
    assert int(o2.state_dict()["state"][0]["step"]) == 2
    for k in range(2, 4):
        step(m2, o2, k)
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    # a parameter without a gradient cannot be skipped under one device-side count: loud
    list(m2.parameters())[0].grad = None
    with pytest.raises(HrvError):
        o2.step()


@pytest.mark.parametrize("case", [("c3x3", [24], 40, 3, 1, 1, 12, 10), ("cat", [16, 8], 130, 3, 1, 1, 8, 12),
                                  ("s2_4x4", [12], 20, 4, 2, 2, 13, 9), ("c1x1", [72], 64, 1, 1, 0, 8, 8),
                                  ("in4", [4], 16, 3, 2, 1, 16, 12)], ids=lambda c: c[0])
def test_mixed_precision_conv_forward_and_dgrad(case):
    """T.MMA_BF16: bf16 matrix cores over fp32 tensors (fp32 accumulate / epilogue).  Reference: fp32 conv of the
    bf16-ROUNDED operands (the only rounding the mode introduces), tolerance = accumulation-order noise."""
    ops, T = _mods()
    name, cins, cout, k, stride, pad, H, W = case
    g = torch.Generator().manual_seed(len(name) + cout)
    N = 2
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    xs = [torch.randn(N, c, H, W, generator=g) for c in cins]
    w = torch.randn(cout, sum(cins), k, k, generator=g) * (1.0 / (sum(cins) * k * k) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.cat(xs, 1)
    ref = F.conv2d(rb(x), rb(w), b, stride=stride, padding=pad)
    dy = torch.randn(ref.shape, generator=g)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, rb(w), rb(dy), stride=stride, padding=pad)
    T.MMA_BF16[0] = True
    try:
        srcs = [(ops.to_nhwc(t.cuda()), 0) for t in xs]
        out = T.conv_forward_dev(w.cuda(), srcs, stride, pad, shift=b.cuda(), name=name)
        dx = T.conv_dgrad(ops.to_nhwc(dy.cuda()), w.cuda(), H, W, stride, pad, name=name + ".dgrad")
    finally:
        T.MMA_BF16[0] = False
    assert out.t.dtype == torch.float32
    got, gdx = ops.to_nchw(out).cpu(), ops.to_nchw(dx).cpu()
    assert (got - ref).abs().max() <= 2e-5 * ref.abs().max() + 1e-5, (got - ref).abs().max()
    assert (gdx - dx_ref).abs().max() <= 2e-5 * dx_ref.abs().max() + 1e-5, (gdx - dx_ref).abs().max()
    # and it IS a different rounding than the fp32 engine (the bf16 operands lose ~3 decimal digits)
    full = F.conv2d(x, w, b, stride=stride, padding=pad)
    assert (got - full).abs().max() > 1e-4 * full.abs().max()


@pytest.mark.parametrize("case", [("w3x3", 24, 40, 3, 1, 1, 12, 12), ("wcat_base", 16, 130, 3, 1, 1, 8, 16),
                                  ("ws2", 12, 96, 4, 2, 1, 16, 24), ("w1x1", 72, 192, 1, 1, 0, 8, 8),
                                  ("wtiny", 4, 16, 3, 2, 1, 16, 24), ("wup", 8, 32, 3, 1, 1, 8, 8),
                                  ("patchgan_odd_s2", 12, 64, 4, 2, 2, 40, 96), ("patchgan_odd_s1", 32, 8, 4, 1, 2, 20, 33)],
                         ids=lambda c: c[0])
def test_mixed_precision_wgrad(case):
    """hrv_conv2d_wgrad_bf16mma_nhwc_f32 (quad-transposed staging, v_mfma_f32_32x32x16_bf16) vs the fp32 weight
    gradient of the bf16-rounded operands.  The odd-width cases (PatchGAN 4x4 convolutions: Wo = W/2 + 1) reach the kernel
    through a zero-padded dY (conv_wgrad)."""
    ops, T = _mods()
    name, cin, cout, k, stride, pad, H, W = case
    g = torch.Generator().manual_seed(len(name) + cout)
    N = 2
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    up = 1 if name == "wup" else 0
    x = torch.randn(N, cin, H >> up, W >> up, generator=g)
    xf = x.repeat_interleave(2, 2).repeat_interleave(2, 3) if up else x
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    assert Wo % 4 == 0 or (name.startswith("patchgan_odd") and Wo >= 32)
    dy = torch.randn(N, cout, Ho, Wo, generator=g)
    ref = torch.nn.grad.conv2d_weight(rb(xf), (cout, cin + 5, k, k)[:1] + (cin,) + (k, k), rb(dy), stride=stride, padding=pad)
    dw = torch.zeros(cout, cin + 8, k, k, device="cuda")       # the source sits at ci_base 4 of a wider Cin axis
    T.MMA_BF16[0] = True
    try:
        T.conv_wgrad(ops.to_nhwc(dy.cuda()), ops.to_nhwc(x.cuda()), up, 4, cin + 8, k, k, stride, pad, dw, name=name)
    finally:
        T.MMA_BF16[0] = False
    got = dw[:, 4:4 + cin].cpu()
    assert (got - ref).abs().max() <= 3e-5 * ref.abs().max() + 1e-5, ((got - ref).abs().max(), ref.abs().max())
    assert dw[:, :4].abs().max() == 0 and dw[:, 4 + cin:].abs().max() == 0


@pytest.mark.parametrize("mixed", [False, True], ids=["fp32", "bf16mma"])
@pytest.mark.parametrize("cin,cout,k", [(128, 48, 3), (24, 130, 3), (64, 64, 1)], ids=["exact_fit_extra_tile", "free_slot", "1x1"])
def test_wgrad_fused_bias_gradient(mixed, cin, cout, k):
    """dbias = sum over pixels of dY, produced by the weight-gradient kernel itself as a ones-column of the same
    MFMA reduction (also when taps*Cin exactly fills the column tiles and the column needs a tile of its own)."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(cin + cout)
    N, H, W = 2, 8, 12
    x = torch.randn(N, cin, H, W, generator=g)
    dy = torch.randn(N, cout, H, W, generator=g)
    rb = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mixed else (lambda t: t)
    want_w = torch.nn.grad.conv2d_weight(rb(x), (cout, cin, k, k), rb(dy), stride=1, padding=k // 2)
    want_b = rb(dy).sum((0, 2, 3))
    dw = torch.empty(cout, cin, k, k, device="cuda")
    db = torch.full((cout,), 7.0, device="cuda")
    T.MMA_BF16[0] = mixed
    try:
        T.conv_wgrad(ops.to_nhwc(dy.cuda()), ops.to_nhwc(x.cuda()), 0, 0, cin, k, k, 1, k // 2, dw, dbias=db)
        db2 = db.clone()
        T.conv_wgrad(ops.to_nhwc(dy.cuda()), ops.to_nhwc(x.cuda()), 0, 0, cin, k, k, 1, k // 2, dw, dbias=db2,
                     dbias_accumulate=True)
    finally:
        T.MMA_BF16[0] = False
    tol = 3e-5
    assert (dw.cpu() - want_w).abs().max() <= tol * want_w.abs().max() + 1e-5
    assert (db.cpu() - want_b).abs().max() <= tol * want_b.abs().max() + 1e-5
    assert (db2.cpu() - 2 * want_b).abs().max() <= 2 * tol * want_b.abs().max() + 1e-5


@pytest.mark.parametrize("case", [("gb", 128, 64, 3, 1, 16, 24), ("shared1x1", 72, 384, 1, 0, 8, 16),
                                  ("sliced", 40, 24, 3, 1, 8, 12)], ids=lambda c: c[0])
@pytest.mark.parametrize("dy_bf16", [False, True], ids=["dy_f32", "dy_bf16"])
def test_mixed_precision_bf16_stored_operands_are_bit_identical(case, dy_bf16):
    """Tensors that only matrix cores read may be STORED in bf16 by the mixed-precision plan (actv, the expanded
    label map, [dgamma|dbeta]).  The MMA operand bits are then the same as when the fp32 tensor is rounded while
    staged, so every consumer -- forward conv, data gradient (bf16 source / bf16 sign mask), weight gradient incl.
    the fused bias column (hrv_conv2d_wgrad_bf16mma_st_nhwc_f32, packed staging) -- gives BIT-IDENTICAL results."""
    ops, T = _mods()
    name, cin, cout, k, pad, H, W = case
    g = torch.Generator().manual_seed(cin + cout)
    N = 2
    # values already representable in bf16: the two storage forms hold the same numbers
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb(torch.randn(N, cin, H, W, generator=g)).cuda()
    dy = rb(torch.randn(N, cout, H, W, generator=g)).cuda()
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).cuda()
    mask = rb(torch.randn(N, cin, H, W, generator=g)).cuda()

    def as_act(t, bf, wide=False):
        if not wide:
            return ops.to_nhwc(t, bf16=bf)
        # a channel slice of a wider tensor (the block-wide actv / d(actv) tensors)
        C_ = t.shape[1]
        full = ops.alloc(t.shape[0], t.shape[2], t.shape[3], C_ + 16, t.device, bf16=bf)
        full.t.zero_()
        ops.to_nhwc(t, out=full.slice(8, C_))
        return full.slice(8, C_)

    wide = name == "sliced"
    outs = {}
    T.MMA_BF16[0] = True
    try:
        for bf in (False, True):
            xa = as_act(x, bf, wide)
            dya = as_act(dy, bf and dy_bf16, wide)
            fwd = T.conv_forward_dev(w, [(xa, 0)], 1, pad, name=name)
            dx = T.conv_dgrad(dya, w, H, W, 1, pad, act_mask=as_act(mask, bf, wide), slope=0.2, name=name + ".dgrad")
            dw = torch.zeros(cout, cin + 8, k, k, device="cuda")
            db = torch.zeros(cout, device="cuda")
            if dya.bf16 and not xa.bf16:
                continue
            T.conv_wgrad(dya, xa, 0, 8, cin + 8, k, k, 1, pad, dw, name=name + ".wgrad", dbias=db)
            outs[bf] = (ops.to_nchw(fwd), ops.to_nchw(dx), dw, db)
    finally:
        T.MMA_BF16[0] = False
    for what, a, b in zip(("forward", "dgrad", "wgrad", "dbias"), outs[False], outs[True]):
        assert torch.equal(a, b), (what, (a - b).abs().max().item())
    ref = torch.nn.grad.conv2d_weight(x.cpu(), (cout, cin, k, k), dy.cpu(), stride=1, padding=pad)
    got = outs[True][2][:, 8:8 + cin].cpu()
    assert (got - ref).abs().max() <= 3e-5 * ref.abs().max() + 1e-5
    assert (outs[True][3].cpu() - dy.cpu().sum((0, 2, 3))).abs().max() <= 1e-4 * dy.abs().sum((0, 2, 3)).max().item()


@pytest.mark.parametrize("case", [("gb_up4", 160, 1, 128, 256, False), ("gb_norm1", 64, 2, 136, 128, False),
                                  ("gb_up3_two_cout_tiles", 288, 1, 352, 96, False), ("sliced_partial_row", 128, 1, 260, 160, True)],
                         ids=lambda c: c[0])
def test_wgrad_tr_kernel_bf16_stored_operands(case, monkeypatch):
    """conv_wgrad_tr_kernel (wgrad_tr.hip: LDS-DMA staging + ds_read_b64_tr_b16 fragments, bf16-STORED dY and X, 128-channel
    source = the SPADE gamma|beta convolutions) vs torch's conv2d_weight on the same bf16-representable operands (fp32
    accumulation: only the summation order differs) and vs the register-transposing kernel it replaces (HRV_WGRAD_TR=0).
    Covers: one / two cout tiles, a row width that is not a multiple of the 64-pixel tile (masked tail, 96 and 160),
    image-border rows and columns (zero padding by out-of-range DMA offsets), channel slices of wider tensors, the bias
    gradient."""
    ops, T = _mods()
    name, cout, N, H, W, wide = case
    cin, k, pad = 128, 3, 1
    g = torch.Generator().manual_seed(cout + H)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb(torch.randn(N, cin, H, W, generator=g))
    dy = rb(torch.randn(N, cout, H, W, generator=g) * (torch.rand(N, cout, H, W, generator=g) > 0.3))

    def as_act(t):
        if not wide:
            return ops.to_nhwc(t.cuda(), bf16=True)
        C_ = t.shape[1]
        full = ops.alloc(t.shape[0], t.shape[2], t.shape[3], C_ + 24, "cuda", bf16=True)
        full.t.normal_()                                      # neighbours of the slice must not leak in
        ops.to_nhwc(t.cuda(), out=full.slice(16, C_))
        return full.slice(16, C_)

    res = {}
    T.MMA_BF16[0] = True
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("HRV_WGRAD_TR", mode)
            from hr_viton_amd import _lib as _hl; _hl.reload_env()
            dw = torch.full((cout, cin + 8, k, k), 7.0, device="cuda")
            db = torch.zeros(cout, device="cuda")
            T.conv_wgrad(as_act(dy), as_act(x), 0, 8, cin + 8, k, k, 1, pad, dw, name=name, dbias=db)
            torch.cuda.synchronize()
            res[mode] = (dw.cpu(), db.cpu())
    finally:
        T.MMA_BF16[0] = False
    ref = torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), dy, stride=1, padding=pad)
    scale = ref.abs().max().item()
    got = res["1"][0]
    assert torch.equal(got[:, :8], torch.full((cout, 8, k, k), 7.0)), "columns outside [ci_base, ci_base+C) must stay untouched"
    assert (got[:, 8:] - ref).abs().max().item() <= 3e-5 * scale + 1e-5, (got[:, 8:] - ref).abs().max().item() / scale
    assert (got[:, 8:] - res["0"][0][:, 8:]).abs().max().item() <= 6e-5 * scale
    dbr = dy.sum((0, 2, 3))
    assert (res["1"][1] - dbr).abs().max().item() <= 1e-4 * dy.abs().sum((0, 2, 3)).max().item()


@pytest.mark.parametrize("case", [("up3_conv0", 144, 64, 1, 136, 260, False), ("up3_conv1", 64, 64, 2, 128, 160, False),
                                  ("up2_conv0", 272, 128, 1, 132, 256, False), ("up1_conv1", 256, 256, 1, 130, 264, False),
                                  ("sliced_160", 160, 64, 1, 128, 288, True), ("up3_conv0_two_cout_blocks", 144, 128, 1, 128, 256, False)],
                         ids=lambda c: c[0])
def test_wgrad_tr_kernel_other_source_widths(case, monkeypatch):
    """conv_wgrad_tr_kernel's row-aligned classes (round 4): 3x3 weight gradients over 144 / 160 / 272 / 256-channel sources (a block =
    the three taps of one kernel row x every 32-channel group x 64 couts; the last group of 144 / 272 is half empty) and over 64
    channels (every tap in one block) -- SPADEResBlock.conv_0 / conv_1 of up_1..up_3 (network_generator.py:141-143) -- vs torch's
    conv2d_weight on the same bf16-representable operands and vs the register-transposing kernel (HRV_WGRAD_TR=0); the bias
    gradient rides along."""
    ops, T = _mods()
    name, cin, cout, N, H, W, wide = case
    k, pad = 3, 1
    g = torch.Generator().manual_seed(cout + cin + H)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb(torch.randn(N, cin, H, W, generator=g))
    dy = rb(torch.randn(N, cout, H, W, generator=g) * (torch.rand(N, cout, H, W, generator=g) > 0.3))

    def as_act(t):
        if not wide:
            return ops.to_nhwc(t.cuda(), bf16=True)
        C_ = t.shape[1]
        full = ops.alloc(t.shape[0], t.shape[2], t.shape[3], C_ + 24, "cuda", bf16=True)
        full.t.normal_()
        ops.to_nhwc(t.cuda(), out=full.slice(16, C_))
        return full.slice(16, C_)

    res = {}
    T.MMA_BF16[0] = True
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("HRV_WGRAD_TR", mode)
            from hr_viton_amd import _lib as _hl; _hl.reload_env()
            dw = torch.full((cout, cin + 8, k, k), 7.0, device="cuda")
            db = torch.zeros(cout, device="cuda")
            ops.profile_begin()
            T.conv_wgrad(as_act(dy), as_act(x), 0, 8, cin + 8, k, k, 1, pad, dw, name=name, dbias=db)
            ops.profile_end()
            torch.cuda.synchronize()
            res[mode] = (dw.cpu(), db.cpu())
    finally:
        T.MMA_BF16[0] = False
    ref = torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), dy, stride=1, padding=pad)
    scale = ref.abs().max().item()
    got = res["1"][0]
    assert torch.equal(got[:, :8], torch.full((cout, 8, k, k), 7.0)), "columns outside [ci_base, ci_base+C) must stay untouched"
    assert (got[:, 8:] - ref).abs().max().item() <= 3e-5 * scale + 1e-5, (got[:, 8:] - ref).abs().max().item() / scale
    assert (got[:, 8:] - res["0"][0][:, 8:]).abs().max().item() <= 6e-5 * scale
    dbr = dy.sum((0, 2, 3))
    assert (res["1"][1] - dbr).abs().max().item() <= 1e-4 * dy.abs().sum((0, 2, 3)).max().item()


@pytest.mark.parametrize("case", [("conv_0", 80, 32, 3, True, False, "none"), ("conv_1", 32, 32, 3, True, True, "lrelu"),
                                  ("conv_s", 80, 32, 1, False, False, "none"), ("conv_img", 32, 3, 3, False, False, "tanh"),
                                  ("vgg_features0", 3, 64, 3, False, False, "relu"), ("vgg_features2", 64, 64, 3, False, True, "relu")],
                         ids=lambda c: c[0])
def test_thin_conv_forward_and_data_gradient(case, monkeypatch):
    """thin_conv.hip (weights converted into LDS once per persistent block, LDS-DMA halo patches, 32 pixels x all columns
    per wave) for the <= 96-channel layers of the 1024x768 level: forward with bias / spectral 1/sigma / residual /
    activation / bf16 output, and the data gradient (transposed, flipped weights), vs torch on the same bf16-representable
    operands (fp32 accumulation) and vs the implicit-GEMM engine it replaces (HRV_THIN_CONV=0).  H and W overhang the 8x16
    tiles, the source is a channel slice of a wider tensor."""
    ops, T = _mods()
    name, cin, cout, k, spectral, with_res, actn = case
    N, H, W, pad = 2, 180, 200, k // 2
    g = torch.Generator().manual_seed(cin * 7 + cout)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb(torch.randn(N, cin, H, W, generator=g))
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.1).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    sigma = torch.tensor([1.7], device="cuda") if spectral else None
    res = torch.randn(N, cout, H, W, generator=g) if with_res else None
    dy = rb(torch.randn(N, cout, H, W, generator=g))
    act = {"none": ops.ACT_NONE, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH, "relu": ops.ACT_RELU}[actn]

    def sliced(t):
        C_ = t.shape[1]
        full = ops.alloc(t.shape[0], t.shape[2], t.shape[3], C_ + 16, "cuda", bf16=True)
        full.t.normal_()
        ops.to_nhwc(t.cuda(), out=full.slice(8, C_))
        return full.slice(8, C_)

    outs = {}
    T.MMA_BF16[0] = True
    try:
        for thin in ("1", "0"):
            monkeypatch.setenv("HRV_THIN_CONV", thin)
            xa = sliced(x)
            ra = ops.to_nhwc(res.cuda()) if with_res else None
            y = T.conv_forward_dev(w, [(xa, 0)], 1, pad, sigma=sigma, shift=b, residual=ra, act=act, slope=0.2, name=name,
                                   out_bf16=with_res)
            outs[thin] = [ops.to_nchw(y).float().cpu()]
            if cout % 8 == 0:
                dx = T.conv_dgrad(sliced(dy), w, H, W, 1, pad, sigma=sigma, name=name + ".dgrad")
                outs[thin].append(ops.to_nchw(dx).float().cpu())
            torch.cuda.synchronize()
    finally:
        T.MMA_BF16[0] = False
    ws = rb(w.cpu() / (1.7 if spectral else 1.0))
    ref = F.conv2d(x, ws, b.cpu(), padding=pad)
    if with_res:
        ref = ref + res
    ref = {"none": lambda t: t, "lrelu": lambda t: F.leaky_relu(t, 0.2), "tanh": torch.tanh, "relu": torch.relu}[actn](ref)
    tol = 1e-2 if with_res else 2e-4            # bf16 output: one bf16 rounding of the result
    assert (outs["1"][0] - ref).abs().max() <= tol * ref.abs().max(), (outs["1"][0] - ref).abs().max() / ref.abs().max()
    assert (outs["1"][0] - outs["0"][0]).abs().max() <= tol * ref.abs().max()
    if cout % 8 == 0:
        refd = torch.nn.grad.conv2d_input(x.shape, ws, dy, stride=1, padding=pad)
        assert (outs["1"][1] - refd).abs().max() <= 2e-4 * refd.abs().max(), (outs["1"][1] - refd).abs().max() / refd.abs().max()
        assert (outs["1"][1] - outs["0"][1]).abs().max() <= 2e-4 * refd.abs().max()


@pytest.mark.parametrize("hw", [(192, 208), (180, 200)], ids=["full_tiles", "overhanging_tiles"])
def test_thin_conv_conv_shared_as_one_384_column_layer(hw, monkeypatch):
    """thin_conv.hip, WIDE variant: a SPADEResBlock's three conv_shared (network_generator.py:99-102) as ONE 1x1 convolution
    over the tap-expanded one-hot label map (72 -> 384 columns, bias + ReLU, bf16 output; lanes l / l+32 exchange 4-channel
    groups so that a lane stores 16 bytes; full tiles wait for the next patch with a counted vmcnt).  vs torch on the same
    bf16-representable operands and vs the implicit-GEMM engine it replaces, for tiles inside the image and overhanging ones,
    over several persistent rounds (N*H*W/128 tiles > 2 x 256 CUs), the output a channel slice of a wider tensor."""
    ops, T = _mods()
    H, W = hw
    N, cin, cout = 3, 72, 384
    g = torch.Generator().manual_seed(H)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb((torch.rand(N, cin, H, W, generator=g) < 0.15).float() + 0.25 * torch.randn(N, cin, H, W, generator=g))
    w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    outs = {}
    T.MMA_BF16[0] = True
    try:
        for thin in ("1", "0"):
            monkeypatch.setenv("HRV_THIN_CONV", thin)
            xa = ops.to_nhwc(x.cuda(), bf16=True)
            full = ops.alloc(N, H, W, cout + 16, "cuda", bf16=True)
            full.t.fill_(3.0)
            y = T.conv_forward_dev(w, [(xa, 0)], 1, 0, shift=b, act=ops.ACT_RELU, out=full.slice(8, cout), out_bf16=True,
                                   name="up_4.conv_shared[x3 as 1x1 over taps]")
            torch.cuda.synchronize()
            outs[thin] = ops.to_nchw(y).float().cpu()
            assert torch.equal(full.t[..., :8].float().cpu(), torch.full((N, H, W, 8), 3.0)), "channels below the slice"
            assert torch.equal(full.t[..., 8 + cout:].float().cpu(), torch.full((N, H, W, 8), 3.0)), "channels above the slice"
    finally:
        T.MMA_BF16[0] = False
    ref = torch.relu(F.conv2d(x, rb(w.cpu()), b.cpu()))
    tol = 2.0 ** -8 * float(ref.abs().max())             # one bf16 rounding of the stored result
    assert (outs["1"] - ref).abs().max() <= tol, ((outs["1"] - ref).abs().max() / ref.abs().max()).item()
    assert (outs["1"] - outs["0"]).abs().max() <= tol


@pytest.mark.parametrize("case", [("tile17_256cols", 8, 128, 96, 128, 256, 17), ("tile18_192cols_split", 4, 128, 96, 128, 192, 18),
                                  ("tile17_2chunks", 8, 128, 96, 256, 128, 17), ("tile18_64cols", 8, 128, 96, 128, 64, 18)],
                         ids=lambda c: c[0])
def test_training_convs_on_patch_tiles_match_torch(case, monkeypatch):
    """conv_forward_dev / conv_dgrad (device-packed per-step weights) over a bf16-stored source at sizes where
    ops.patch_tile SELECTS the LDS-resident patch tiles (>= 512 tiles; the parity fixtures are too small for that):
    vs torch on the same bf16-representable operands and vs the gather tiles (HRV_CONV_PATCH=0).  Pins the device
    packer's row size for tiles 16-18 (it packed 32 k-values per row where the kernels read 64: wrong weights in every
    mixed-precision training convolution that picked a patch tile, which the 2e-2 bf16 image tolerance did not catch
    because gamma / beta are small at initialisation -- tools/diag/patch_train_check.py, patch_spade_check.py)."""
    ops, T = _mods()
    name, N, H, W, cin, cout, want_cfg = case
    g = torch.Generator().manual_seed(cin + cout)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb(torch.randn(N, cin, H, W, generator=g))
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    dy = rb(torch.randn(N, cout, H, W, generator=g))
    assert ops.patch_tile(True, 3, 3, 1, 1, 1, 0, cin, cout, N, H, W) == want_cfg
    outs = {}
    T.MMA_BF16[0] = True
    try:
        for env in ("1", "0"):
            monkeypatch.setenv("HRV_CONV_PATCH", env)
            xa = ops.to_nhwc(x.cuda(), bf16=True)
            y = T.conv_forward_dev(w, [(xa, 0)], 1, 1, shift=b, act=ops.ACT_LRELU, slope=0.2, name=name)
            outs[env] = [ops.to_nchw(y).float().cpu()]
            if cout % 128 == 0:            # the data gradient is a 'same' 3x3 convolution over dY: patch tiles too
                assert env == "0" or ops.patch_tile(True, 3, 3, 1, 1, 1, 0, cout, cin, N, H, W) in (17, 18)
                dx = T.conv_dgrad(ops.to_nhwc(dy.cuda(), bf16=True), w, H, W, 1, 1, name=name + ".dgrad")
                outs[env].append(ops.to_nchw(dx).float().cpu())
            torch.cuda.synchronize()
    finally:
        T.MMA_BF16[0] = False
    ref = F.leaky_relu(F.conv2d(x, rb(w.cpu()), b.cpu(), padding=1), 0.2)
    assert (outs["1"][0] - ref).abs().max() <= 2e-5 * ref.abs().max(), (outs["1"][0] - ref).abs().max() / ref.abs().max()
    assert (outs["1"][0] - outs["0"][0]).abs().max() <= 2e-5 * ref.abs().max()
    if cout % 128 == 0:
        dref = F.conv_transpose2d(dy, rb(w.cpu()), padding=1)
        assert (outs["1"][1] - dref).abs().max() <= 2e-5 * dref.abs().max()
        assert (outs["1"][1] - outs["0"][1]).abs().max() <= 2e-5 * dref.abs().max()


@pytest.mark.parametrize("case", [("C64_tile17", 2, 256, 192, 64, 17), ("C32_tile18", 4, 256, 192, 32, 18),
                                  ("C96_tile18_split", 2, 256, 192, 96, 18)], ids=lambda c: c[0])
def test_spade_training_layer_on_patch_tiles(case, monkeypatch):
    """gen_train.SpadeT (network_generator.py:93-118 in training mode: noise, instance-norm statistics, fused gamma|beta
    convolution + modulate + LeakyReLU, bf16 output) with the patch tiles selected vs the gather tiles (bit-identical
    weights, same summation per output) and vs torch on the same bf16-representable operands; actv is a channel slice of
    the block-wide tensor, as in the plan."""
    from argparse import Namespace
    ops, T = _mods()
    from hr_viton_amd.gen_train import SpadeT
    from hr_viton_amd.network_generator import SPADENorm
    name, N, H, W, Cc, want_cfg = case
    torch.manual_seed(Cc)
    norm = SPADENorm(Namespace(), "aliasinstance", Cc, 7).cuda()
    with torch.no_grad():
        for p in norm.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    rb = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    x = torch.randn(N, H, W, Cc, device="cuda")
    z = torch.randn(N, W, H, 1, device="cuda")
    actv_all = torch.relu(torch.randn(N, H, W, 384, device="cuda")).to(torch.bfloat16)
    a_nchw = actv_all[..., 128:256].float().permute(0, 3, 1, 2)
    gamma = F.conv2d(a_nchw, rb(norm.conv_gamma.weight), norm.conv_gamma.bias, padding=1)
    beta = F.conv2d(a_nchw, rb(norm.conv_beta.weight), norm.conv_beta.bias, padding=1)
    xn = F.instance_norm(x.permute(0, 3, 1, 2) + (z * norm.noise_scale).transpose(1, 3), eps=1e-5)
    ref = F.leaky_relu(xn * (1 + gamma) + beta, 0.2).permute(0, 2, 3, 1)
    st = SpadeT(norm, ops.ACT_LRELU, name)
    assert ops.patch_tile(True, 3, 3, 1, 1, 1, 0, 128, st.G * 64, N, H, W, wide=True) == want_cfg
    outs = {}
    T.MMA_BF16[0] = True
    monkeypatch.setenv("HRV_SPADE_GB", "0")      # this test pins the GENERIC patch tiles (tests/test_gpu_spade_gb.py: the dedicated kernel)
    try:
        for env in ("1", "0"):
            monkeypatch.setenv("HRV_CONV_PATCH", env)
            out, ctx = st.forward(ops.Act(x, Cc), ops.Act(actv_all, 128, 128), z, save=True)
            outs[env] = (out.t[..., :Cc].float(), ctx["g1p"].t[..., :Cc].float())
            assert out.bf16 == (Cc % 8 == 0)
    finally:
        T.MMA_BF16[0] = False
    tol = 2.0 ** -8 * float(ref.abs().max())             # one bf16 rounding of the stored result
    assert float((outs["1"][0] - ref).abs().max()) <= tol
    assert torch.equal(outs["1"][0], outs["0"][0])
    assert float((outs["1"][1] - (1 + gamma).permute(0, 2, 3, 1)).abs().max()) <= 2e-4 * float((1 + gamma).abs().max())


@pytest.mark.parametrize("iters", [1, 0], ids=["train", "eval"])
def test_batched_spectral_norm_matches_per_layer_and_torch(iters):
    """hrv_spectral_norm_batched_f32 (every spectral-normalised convolution of a network in four launches) vs the
    per-layer entry and vs torch's SpectralNorm.compute_weight formulas, incl. the kept (u, v) copies; the shared
    scratch starts EMPTY (a too-small scratch faulted in the first call of a fresh process)."""
    ops, T = _mods()
    ops._WS.clear()
    g = torch.Generator().manual_seed(3)
    shapes = [(16, 9, 3, 3), (1024, 64, 3, 3), (8, 8, 1, 1), (130, 24, 3, 3), (3, 64, 3, 3)]
    ws = [torch.randn(s, generator=g).cuda() for s in shapes]
    us = [F.normalize(torch.randn(s[0], generator=g), dim=0).cuda() for s in shapes]
    vs = [F.normalize(torch.randn(s[1] * s[2] * s[3], generator=g), dim=0).cuda() for s in shapes]
    u1, v1 = [u.clone() for u in us], [v.clone() for v in vs]
    sig1 = [T.spectral_sigma(w, u, v, iters) for w, u, v in zip(ws, u1, v1)]
    u2, v2 = [u.clone() for u in us], [v.clone() for v in vs]
    sig2, uk, vk = T.SpectralBatch(list(zip(ws, u2, v2))).run(iters)
    for j, w in enumerate(ws):
        Wm = w.reshape(w.shape[0], -1)
        u, v = us[j], vs[j]
        if iters:
            v = F.normalize(Wm.t() @ u, dim=0, eps=1e-12)
            u = F.normalize(Wm @ v, dim=0, eps=1e-12)
        sg = torch.dot(u, Wm @ v)
        for got_s, got_u, got_v in ((sig1[j], u1[j], v1[j]), (sig2[j], u2[j], v2[j]), (sig2[j], uk[j], vk[j])):
            assert abs(float(got_s) - float(sg)) <= 2e-5 * abs(float(sg)), (j, float(got_s), float(sg))
            assert torch.allclose(got_u, u, atol=2e-6) and torch.allclose(got_v, v, atol=2e-6), j


def test_bf16_stored_pool_and_loss_kernels_match_fp32_on_the_same_values():
    """Mixed-precision VGG19 keeps its activations in bf16: the 2x2 max pool (exact), its backward with the fused ReLU
    derivative, and the L1 tap loss over bf16-stored operands give what the fp32 kernels give on the same values."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(2, 20, 24, 64, generator=g)).to(torch.bfloat16).cuda()
    y = torch.relu(torch.randn(2, 20, 24, 64, generator=g)).to(torch.bfloat16).cuda()
    xa, xf = ops.Act(x, 64), ops.Act(x.float(), 64)
    p16, p32 = T.maxpool2x2(xa), T.maxpool2x2(xf)
    assert p16.bf16 and torch.equal(p16.t.float(), p32.t)
    # a ReLU written as max(v, v * 0) stores -0 (pattern 0x8000) for v < 0: the pool must not rank it above positive values
    xn = torch.where(x == 0, torch.full_like(x, -0.0), x)
    assert int((xn.view(torch.int16) == -32768).sum()) > 1000
    assert torch.equal(T.maxpool2x2(ops.Act(xn, 64)).t.float(), p32.t)
    dy = ops.Act(torch.randn(2, 10, 12, 64, generator=g).cuda(), 64)
    assert torch.equal(T.maxpool2x2_bwd(xa, dy, relu=True).t, T.maxpool2x2_bwd(xf, dy, relu=True).t)
    l16, l32 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    g16 = T.loss(x, y, T.LOSS_L1 | T.LOSS_RELU_MASK, 0.5, 0.25, l16, accumulate=False)
    g32 = T.loss(x.float(), y.float(), T.LOSS_L1 | T.LOSS_RELU_MASK, 0.5, 0.25, l32, accumulate=False)
    assert g16.dtype == torch.float32 and torch.equal(g16, g32) and torch.equal(l16, l32)


def test_bf16_stored_gradients_of_the_vgg_backward_match_fp32_on_the_same_values():
    """Mixed-precision VGG19 backward keeps its GRADIENT tensors in bf16 as well: the tap-loss gradient (mode | 32), the
    2x2 max-pool backward over bf16 x / dy / dx and the bf16 accumulation give the fp32 kernels' results on the same
    (bf16-representable) values, rounded once."""
    ops, T = _mods()
    g = torch.Generator().manual_seed(6)
    rb = lambda t: t.to(torch.bfloat16)     # noqa: E731
    x = torch.relu(torch.randn(2, 20, 24, 64, generator=g)).to(torch.bfloat16).cuda()
    y = torch.relu(torch.randn(2, 20, 24, 64, generator=g)).to(torch.bfloat16).cuda()
    l16, l32 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    g16 = T.loss(x, y, T.LOSS_L1 | T.LOSS_RELU_MASK, 0.5, 0.25, l16, accumulate=False, grad_bf16=True)
    g32 = T.loss(x, y, T.LOSS_L1 | T.LOSS_RELU_MASK, 0.5, 0.25, l32, accumulate=False)
    assert g16.dtype == torch.bfloat16 and torch.equal(g16, rb(g32)) and torch.equal(l16, l32)
    dy = torch.randn(2, 10, 12, 64, generator=g).to(torch.bfloat16).cuda()
    d16 = T.maxpool2x2_bwd(ops.Act(x, 64), ops.Act(dy, 64), relu=True)
    d32 = T.maxpool2x2_bwd(ops.Act(x, 64), ops.Act(dy.float(), 64), relu=True)
    assert d16.bf16 and torch.equal(d16.t.float(), d32.t)          # routing only: exact
    a = torch.randn(2, 20, 24, 64, generator=g).to(torch.bfloat16).cuda()
    acc = d16.t.clone()
    T.add_slice(ops.Act(a, 64), ops.Act(acc, 64), True)
    assert torch.equal(acc, rb(a.float() + d16.t.float()))


def test_space_to_depth_pair_and_patchgan_model0_as_a_2x2_convolution(monkeypatch):
    """hrv_space_to_depth2 / depth_to_space2 against torch indexing (source a channel slice), then gen_train.S2DConv --
    PatchGAN's 4x4 stride-2 pad-2 first convolution (network_generator.py NLayerDiscriminator model0, spectral-normalised,
    10 input channels) run as a 2x2 stride-1 pad-1 convolution over the space-to-depth tensor -- against torch autograd on
    the same weights: output, dW_orig through the spectral-norm backward, bias gradient, input gradient (fp32 engine:
    reassociation only), and the half-batch backward of the generator step."""
    import torch.nn as nn
    ops, T = _mods()
    from hr_viton_amd import gen_train
    g = torch.Generator().manual_seed(12)
    N, C_, H, W = 4, 10, 96, 80
    x = torch.randn(N, C_, H, W, generator=g)
    wide = ops.alloc(N, H, W, 24, "cuda")
    wide.t.normal_()
    a = wide.slice(8, 12)
    ops.to_nhwc(x.cuda(), out=wide.slice(8, C_))
    wide.t[..., 8 + C_:20] = 0
    a2 = T.space_to_depth2(Act_(ops, a.t, C_, 8))
    ref2 = torch.zeros(N, H // 2, W // 2, 4, 12)
    for dy in range(2):
        for dx in range(2):
            ref2[:, :, :, dy * 2 + dx, :C_] = x[:, :, dy::2, dx::2].permute(0, 2, 3, 1)
    assert a2.C == 48 and torch.equal(a2.t.cpu(), ref2.view(N, H // 2, W // 2, 48))
    back = T.depth_to_space2(a2, C_)
    assert torch.equal(ops.to_nchw(back).cpu(), x)

    conv = nn.utils.spectral_norm(nn.Conv2d(C_, 64, 4, stride=2, padding=2)).cuda().train()
    with torch.no_grad():
        conv.weight_orig.mul_(3.0)
    assert gen_train.S2DConv.fits(conv)
    tc = gen_train.S2DConv(conv, "model0")
    tc.refresh()
    tc.prepare(power_iteration=True)
    xa = ops.to_nhwc(x.cuda())
    y = tc.forward([(xa, 0)], act=ops.ACT_LRELU)
    w_sn = (conv.weight_orig / tc.sigma).detach().cpu().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = F.leaky_relu(F.conv2d(xr, w_sn, conv.bias.detach().cpu(), stride=2, padding=2), 0.2)
    assert tuple(y.t.shape) == (N, H // 2 + 1, W // 2 + 1, 64)
    assert (ops.to_nchw(y).cpu() - yr).abs().max() <= 2e-5 * yr.abs().max()
    dyt = torch.randn(yr.shape, generator=g)
    yr.backward(dyt)
    d = ops.to_nhwc(dyt.cuda())
    T.act_bwd_(d, y, ops.ACT_LRELU, 0.2)
    grads = {}
    dx = tc.backward(d, [(xa, 0)], grads, need_dx=True)
    assert (ops.to_nchw(dx).cpu() - xr.grad).abs().max() <= 2e-5 * xr.grad.abs().max()
    gw = grads[conv.weight_orig] if conv.weight_orig in grads else conv.weight_orig.grad
    # d/dW_orig of W_orig / sigma(W_orig) with (u, v) held constant, as torch's spectral_norm backward does
    u, v, sig = tc.u.cpu(), tc.v.cpu(), tc.sigma.cpu()
    Gm = w_sn.grad.reshape(64, -1)
    ref_gw = ((Gm - (Gm * w_sn.detach().reshape(64, -1)).sum() * torch.outer(u, v)) / sig).reshape(64, C_, 4, 4)
    assert (gw.cpu() - ref_gw).abs().max() <= 5e-5 * ref_gw.abs().max()
    gb = grads[conv.bias] if conv.bias in grads else conv.bias.grad
    assert (gb.cpu() - dyt.mul((yr > 0).float() + 0.2 * (yr <= 0).float()).sum((0, 2, 3))).abs().max() <= 1e-4 * dyt.abs().sum((0, 2, 3)).max()
    # half-batch backward (generator step: the fake half only)
    half = N // 2
    dxh = tc.backward(Act_(ops, d.t[:half], 64, 0), [(Act_(ops, xa.t[:half], C_, 0), 0)], {}, need_dx=True, need_w=False)
    assert (ops.to_nchw(dxh).cpu() - xr.grad[:half]).abs().max() <= 2e-5 * xr.grad.abs().max()


def Act_(ops, t, C_, coff):
    return ops.Act(t, C_, coff)


@pytest.mark.parametrize("shape", [(7, 3, 8), (5, 6, 8), (13, 3, 20)], ids=["patchgan_7_3", "generic_5_6", "sliced_13_3"])
def test_concat_of_an_nhwc_and_an_nchw_tensor(shape):
    """hrv_concat_nhwc_nchw_f32 (the PatchGAN input cat((parse, image), 1), train_generator.py:283-284) == torch.cat, pad
    channels zero: the vectorised 7 + 3 form and the generic form (a channel slice as the NHWC operand)."""
    ops, T = _mods()
    Ca, Cb, acs = shape
    N, H, W = 3, 37, 45
    g = torch.Generator().manual_seed(Ca)
    a = torch.randn(N, H, W, acs, generator=g).cuda()
    b = torch.randn(N, Cb, H, W, generator=g).cuda()
    coff = 0 if acs == 8 else 4
    cs = (Ca + Cb + 3) // 4 * 4
    out = ops.Act(torch.full((N, H, W, cs), 9.0, device="cuda"), Ca + Cb)
    T.concat_nhwc_nchw(ops.Act(a, Ca, coff), b, out)
    ref = torch.cat((a[..., coff:coff + Ca], b.permute(0, 2, 3, 1)), 3)
    assert torch.equal(out.t[..., :Ca + Cb], ref) and bool((out.t[..., Ca + Cb:] == 0).all())


@pytest.mark.parametrize("case", [("patchgan_last_4x4", 256, 4, 2, 3, 35, 27, False), ("wide_512ch_bf16_operands", 512, 4, 2, 2, 18, 19, True),
                                  ("k3_same_64ch", 64, 3, 1, 2, 20, 16, False)], ids=lambda c: c[0])
def test_one_output_channel_convolution_kernels(case, monkeypatch):
    """conv_cout1.hip -- PatchGAN's last layer Conv2d(nf, 1, 4, stride=1, padding=2) (network_generator.py
    NLayerDiscriminator) -- through conv_forward_dev / conv_dgrad (with the feature-matching addend) / conv_wgrad: against
    torch (fp32; mixed precision: on bf16-rounded operands, the engine's arithmetic) and against the implicit-GEMM engine
    it replaces (HRV_CONV_COUT1=0); the source is a channel slice of a wider tensor."""
    ops, T = _mods()
    name, Cin, K, pad, N, H, W, mixed = case
    g = torch.Generator().manual_seed(Cin + K)
    rb = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mixed else (lambda t: t)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(1, Cin, K, K, generator=g) * 0.05).cuda()
    b = torch.randn(1, generator=g).cuda()
    Ho, Wo = H + 2 * pad - K + 1, W + 2 * pad - K + 1
    dy = torch.randn(N, 1, Ho, Wo, generator=g)
    addt = torch.randn(N, Cin, H, W, generator=g)
    res = {}
    T.MMA_BF16[0] = mixed
    try:
        for mode in ("1", "0"):
            monkeypatch.setenv("HRV_CONV_COUT1", mode)
            wide = ops.alloc(N, H, W, Cin + 8, "cuda")
            wide.t.normal_()
            xa = wide.slice(4, Cin)
            ops.to_nhwc(x.cuda(), out=xa)
            y = T.conv_forward_dev(w, [(xa, 0)], 1, pad, shift=b, name=name)
            dya = ops.to_nhwc(dy.cuda())
            dx = T.conv_dgrad(dya, w, H, W, 1, pad, add=ops.to_nhwc(addt.cuda()), name=name + ".dgrad")
            dw = torch.full((1, Cin, K, K), 5.0, device="cuda")
            db = torch.full((1,), 5.0, device="cuda")
            T.conv_wgrad(dya, xa, 0, 0, Cin, K, K, 1, pad, dw, name=name + ".wgrad", dbias=db)
            torch.cuda.synchronize()
            assert bool((y.t[..., 1:] == 0).all()), "pad channels of the one-channel output stay zero"
            res[mode] = (ops.to_nchw(y).cpu(), ops.to_nchw(dx).cpu(), dw.cpu(), db.cpu())
    finally:
        T.MMA_BF16[0] = False
    xr, wr, dyr = rb(x), rb(w.cpu()), rb(dy)
    ref_y = F.conv2d(xr, wr, b.cpu(), padding=pad)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, wr, dyr, padding=pad) + addt
    ref_dw = torch.nn.grad.conv2d_weight(xr, w.shape, dyr, padding=pad)
    tol = 2e-5
    got = res["1"]
    assert (got[0] - ref_y).abs().max() <= tol * ref_y.abs().max()
    assert (got[1] - ref_dx).abs().max() <= tol * ref_dx.abs().max()
    assert (got[2] - ref_dw).abs().max() <= 5 * tol * ref_dw.abs().max()
    assert abs(got[3].item() - dy.sum().item()) <= 1e-4 * dy.abs().sum().item()
    # ... and the engine it replaces computes the same thing (same operand rounding in mixed precision)
    for a, e, s in zip(got[:3], res["0"][:3], (ref_y, ref_dx, ref_dw)):
        assert (a - e).abs().max() <= 1e-4 * s.abs().max()


def test_fused_adam_keeps_a_step_count_per_parameter_like_torch():
    """hr_viton_amd.optim.Adam against torch.optim.Adam when a parameter has NO gradient in some iterations (torch skips it:
    weight, moments and its step count stay) -- the bias correction of a parameter that fell behind uses its own count
    (one fused launch per run of equal counts), and state_dict() reports the per-parameter steps."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.optim import Adam
    g = torch.Generator().manual_seed(8)
    shapes = [(7, 5), (33,), (4, 3, 3, 3), (10,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ro = torch.optim.Adam(ref_p, lr=1e-2, betas=(0.5, 0.9), weight_decay=1e-3)
    ho = Adam(hip_p, lr=1e-2, betas=(0.5, 0.9), weight_decay=1e-3)
    present = [(1, 1, 1, 1), (1, 0, 1, 0), (1, 1, 0, 0), (0, 1, 1, 1), (1, 1, 1, 1)]
    for it, mask in enumerate(present):
        for p, q, on in zip(ref_p, hip_p, mask):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone() if on else None
            q.grad = gr.cuda() if on else None
        ro.step()
        ho.step()
        for i, (p, q) in enumerate(zip(ref_p, hip_p)):
            assert _rel(q.detach(), p.detach()) < 2e-6, (it, i)
    sd = ho.state_dict()["state"]
    assert [int(sd[i]["step"]) for i in range(4)] == [sum(m[i] for m in present) for i in range(4)] == [4, 4, 4, 3]


def test_wgrad_tr_kernel_below_its_default_size(monkeypatch):
    """HRV_WGRAD_TR_MIN_PIX lowers the smallest N*H*W conv_wgrad_tr_kernel takes (default 8192): the generator's 64x48 level
    (2 x 64 x 48 = 6144 pixels here, a 48-pixel row inside the 64-pixel row tile) against torch's conv2d_weight and against the
    register-transposing kernel that serves the level by default."""
    ops, T = _mods()
    cout, N, H, W, cin, k, pad = 160, 2, 64, 48, 128, 3, 1
    g = torch.Generator().manual_seed(12)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb(torch.randn(N, cin, H, W, generator=g))
    dy = rb(torch.randn(N, cout, H, W, generator=g) * (torch.rand(N, cout, H, W, generator=g) > 0.3))
    res = {}
    T.MMA_BF16[0] = True
    try:
        for minpix in ("4096", "32768"):      # (the default is 8192: this 6144-pixel case sits below it)
            monkeypatch.setenv("HRV_WGRAD_TR_MIN_PIX", minpix)
            from hr_viton_amd import _lib as _hl; _hl.reload_env()
            dw = torch.zeros((cout, cin, k, k), device="cuda")
            db = torch.zeros(cout, device="cuda")
            ops.profile_begin()
            T.conv_wgrad(ops.to_nhwc(dy.cuda(), bf16=True), ops.to_nhwc(x.cuda(), bf16=True), 0, 0, cin, k, k, 1, pad, dw, name="lowres", dbias=db)
            ops.profile_end()
            torch.cuda.synchronize()
            res[minpix] = (dw.cpu(), db.cpu())
    finally:
        T.MMA_BF16[0] = False
    ref = torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), dy, stride=1, padding=pad)
    scale = ref.abs().max().item()
    for key in res:
        assert (res[key][0] - ref).abs().max().item() <= 3e-5 * scale + 1e-5, key
        assert (res[key][1] - dy.sum((0, 2, 3))).abs().max().item() <= 1e-4 * dy.abs().sum((0, 2, 3)).max().item(), key
    assert not torch.equal(res["4096"][0], res["32768"][0]), "the two settings ran the same kernel (summation orders would differ)"
