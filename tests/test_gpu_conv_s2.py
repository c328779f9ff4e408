"""csrc/conv_s2.hip -- PatchGAN's 4x4 stride-2 pad-2 convolution (NLayerDiscriminator, network_generator.py:263-272) over a bf16-stored
NHWC feature map and its data gradient, two blocks per CU -- against plain torch on the same bf16-rounded operands: forward with bias
(+ LeakyReLU) at odd and even extents, the data gradient with the feature-matching tap gradient added and the LeakyReLU derivative of
the layer's input applied in the epilogue, and the 2x2 form over a space-to-depth image (model0)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("Cin,Cout,N,H,W,out_bf16,act", [(64, 128, 2, 129, 97, False, 0), (128, 256, 1, 65, 49, False, 0),
                                                          (64, 128, 1, 257, 193, True, 2), (32, 64, 2, 64, 48, True, 0),
                                                          (64, 192, 1, 40, 57, False, 0), (96, 128, 8, 130, 98, False, 0),
                                                          (64, 128, 4, 513, 385, True, 2)])      # 884 tiles: several units per resident block
def test_forward_matches_torch(Cin, Cout, N, H, W, out_bf16, act):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(Cin + Cout + H)
    xall = torch.randn(N, H, W, Cin + 8, generator=g).to(torch.bfloat16).cuda()          # a channel slice of a wider tensor
    x = ops.Act(xall, Cin, 8)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) * (2.0 / (16 * Cin)) ** 0.5).cuda()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    sigma = torch.tensor([1.7], device="cuda")
    Ho, Wo = H // 2 + 1, W // 2 + 1
    oall = torch.full((N, Ho, Wo, Cout + 16), 7.0, device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    out = ops.Act(oall, Cout, 8)
    pk = T.conv_s2_pack(T.S2_FWD, w, Cin, Cout, sigma=sigma)
    T.conv_s2(T.S2_FWD, x, pk, Cout, out, bias=b, act=act, slope=0.2, name="t")
    torch.cuda.synchronize()
    want = F.conv2d(xall[..., 8:].float().permute(0, 3, 1, 2), _bf(w * (1.0 / sigma)), b, stride=2, padding=2)      # (the packer's w * (wscale / sigma))
    if act == 2:
        want = F.leaky_relu(want, 0.2)
    want = want.permute(0, 2, 3, 1)
    assert tuple(want.shape[1:3]) == (Ho, Wo)
    got = oall[..., 8:8 + Cout].float()
    tol = (want.abs() * 2 ** -8 if out_bf16 else 0.0) + 2e-4 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())
    assert bool((oall[..., :8] == 7.0).all()) and bool((oall[..., 8 + Cout:] == 7.0).all())      # neighbours untouched


@pytest.mark.parametrize("Ck,Cph,N,H,W,out_bf16,extra", [(128, 64, 2, 129, 97, True, "both"), (256, 128, 1, 65, 49, True, "both"),
                                                          (128, 64, 1, 64, 48, False, "none"), (64, 32, 2, 33, 41, True, "mask"),
                                                          (128, 64, 1, 257, 193, True, "res32"), (256, 128, 4, 66, 50, True, "both"),
                                                          (128, 64, 4, 513, 385, True, "both")])      # 884 tiles x 2 passes
def test_data_gradient_matches_torch(Ck, Cph, N, H, W, out_bf16, extra):
    """dX = (conv^T(dY) [+ tap]) [* lrelu'(x)]: the forward layer maps Cph -> Ck channels over an H x W input."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(Ck * 3 + Cph + H)
    Hy, Wy = H // 2 + 1, W // 2 + 1
    dy_t = torch.randn(N, Hy, Wy, Ck, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(Ck, Cph, 4, 4, generator=g) * 0.05).cuda()
    xin = F.leaky_relu(torch.randn(N, H, W, Cph, generator=g), 0.2).to(torch.bfloat16).cuda()
    res_f32 = extra == "res32"
    tap = (torch.randn(N, H, W, Cph, generator=g) * 0.3).to(torch.float32 if res_f32 else torch.bfloat16).cuda()
    out = ops.alloc(N, H, W, Cph, "cuda", bf16=out_bf16)
    out.t.fill_(5.0)
    pk = T.conv_s2_pack(T.S2_DGRAD, w, Ck, 4 * Cph, Cph)
    T.conv_s2(T.S2_DGRAD, ops.Act(dy_t, Ck), pk, 4 * Cph, out, Cph=Cph,
              residual=ops.Act(tap, Cph) if extra in ("both", "res32") else None,
              mask=ops.Act(xin, Cph) if extra in ("both", "mask") else None, mask_slope=0.2, name="t")
    torch.cuda.synchronize()
    want = F.conv_transpose2d(dy_t.float().permute(0, 3, 1, 2), _bf(w), stride=2, padding=2,
                              output_padding=(H + 4 - 4 - 2 * (Hy - 1), W + 4 - 4 - 2 * (Wy - 1))).permute(0, 2, 3, 1)
    assert tuple(want.shape[1:3]) == (H, W)
    if extra in ("both", "res32"):
        want = want + tap.float()
    if extra in ("both", "mask"):
        want = want * torch.where(xin.float() > 0, 1.0, 0.2)
    got = out.t[..., :Cph].float()
    tol = (want.abs() * 2 ** -8 if out_bf16 else 0.0) + 3e-4 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())


@pytest.mark.parametrize("Cin,Cout,N,H,W", [(10, 64, 2, 128, 96), (10, 64, 1, 256, 192), (3, 64, 1, 66, 34)])
def test_cells_form_matches_torch(Cin, Cout, N, H, W):
    """model0: 4x4 stride-2 pad-2 over few channels as a 2x2 convolution over the space-to-depth image (even H, W)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(Cin + H)
    Cq = (Cin + 3) // 4 * 4
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1)
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    # space-to-depth: channel (dy*2+dx)*Cq + c of cell (cy, cx) = pixel (2cy+dy, 2cx+dx)
    xs = torch.zeros(N, H // 2, W // 2, 2, 2, Cq)
    xs[..., :Cin] = x.view(N, Cin, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1)
    xs = xs.view(N, H // 2, W // 2, 4 * Cq).to(torch.bfloat16).cuda()
    w2 = torch.zeros(Cout, 2, 2, Cq, 2, 2)
    w2[:, :, :, :Cin] = w.view(Cout, Cin, 2, 2, 2, 2).permute(0, 3, 5, 1, 2, 4)      # (co, c, ty, dy, tx, dx) -> (co, dy, dx, c, ty, tx)
    w2 = w2.view(Cout, 4 * Cq, 2, 2).contiguous().cuda()
    Ho, Wo = H // 2 + 1, W // 2 + 1
    out = ops.alloc(N, Ho, Wo, Cout, "cuda", bf16=True)
    pk = T.conv_s2_pack(T.S2_CELLS, w2, 4 * Cq, Cout)
    T.conv_s2(T.S2_CELLS, ops.Act(xs, 4 * Cq), pk, Cout, out, bias=b, act=2, slope=0.2, name="t")
    torch.cuda.synchronize()
    want = F.leaky_relu(F.conv2d(_bf(x).cuda(), _bf(w).cuda(), b, stride=2, padding=2), 0.2).permute(0, 2, 3, 1)
    got = out.t.float()
    tol = want.abs() * 2 ** -8 + 2e-4 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())


@pytest.mark.parametrize("mode,Cin,Cout,N,H,W", [("fwd", 64, 128, 2, 129, 97), ("fwd", 128, 256, 1, 65, 49), ("cells", 12, 64, 2, 128, 96)])
def test_split3_operands_reach_fp32_accuracy(mode, Cin, Cout, N, H, W):
    """HRV_S2_SPLIT3: the source as [hi | lo | hi] (T.split3), the weight packed [hi | hi | lo] -- hi*hi + lo*hi + hi*lo against the
    fp32 convolution of the UNROUNDED operands: ~2^-16 relative instead of the 2^-8 of plain bf16 operands."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(Cin + H)
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    if mode == "fwd":
        x = torch.randn(N, H, W, Cin, generator=g).cuda()
        w = (torch.randn(Cout, Cin, 4, 4, generator=g) * (2.0 / (16 * Cin)) ** 0.5).cuda()
        want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=2).permute(0, 2, 3, 1)
        src = T.split3(ops.Act(x, Cin))
        pk = T.conv_s2_pack(T.S2_FWD, w, 3 * Cin, Cout, split3=True)
        out = ops.alloc(N, H // 2 + 1, W // 2 + 1, Cout, "cuda")
        T.conv_s2(T.S2_FWD, src, pk, Cout, out, bias=b, name="t")
        plain = ops.alloc(N, H // 2 + 1, W // 2 + 1, Cout, "cuda")
        T.conv_s2(T.S2_FWD, ops.Act(x.to(torch.bfloat16), Cin), T.conv_s2_pack(T.S2_FWD, w, Cin, Cout), Cout, plain, bias=b, name="t")
    else:
        x = torch.randn(N, H // 2, W // 2, 4 * Cin, generator=g).cuda()          # a space-to-depth image
        w = (torch.randn(Cout, 4 * Cin, 2, 2, generator=g) * 0.1).cuda()
        want = F.conv2d(F.pad(x.permute(0, 3, 1, 2).double(), (1, 1, 1, 1)), w.double(), b.double()).permute(0, 2, 3, 1)
        src = T.split3(ops.Act(x, 4 * Cin))
        pk = T.conv_s2_pack(T.S2_CELLS, w, 12 * Cin, Cout, split3=True)
        out = ops.alloc(N, H // 2 + 1, W // 2 + 1, Cout, "cuda")
        T.conv_s2(T.S2_CELLS, src, pk, Cout, out, bias=b, name="t")
        plain = ops.alloc(N, H // 2 + 1, W // 2 + 1, Cout, "cuda")
        T.conv_s2(T.S2_CELLS, ops.Act(x.to(torch.bfloat16), 4 * Cin), T.conv_s2_pack(T.S2_CELLS, w, 4 * Cin, Cout), Cout, plain, bias=b, name="t")
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    err = float((out.t.double() - want).abs().max()) / scale
    err_plain = float((plain.t.double() - want).abs().max()) / scale
    assert err < 2e-5, (err, err_plain)
    assert err < err_plain / 50, (err, err_plain)


@pytest.mark.parametrize("N,H,W,Cq,Cout", [(2, 129, 97, 12, 64), (1, 257, 193, 12, 64), (2, 65, 77, 16, 128)])
def test_cells_weight_gradient_on_the_lds_dma_kernel(N, H, W, Cq, Cout):
    """model0's weight gradient: the space-to-depth image carries a one-cell zero border, so the layer is a 'same' 2x2 convolution (pad 1
    on top / left) and wgrad_tr.hip's 2x2 class serves it at any width -- against autograd on the same bf16-rounded operands, with the
    bias gradient."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    T.MMA_BF16[0] = True
    try:
        g = torch.Generator().manual_seed(N + H)
        K = 4 * Cq
        xs = torch.randn(N, H, W, K, generator=g)
        xs[:, -1] = 0
        xs[:, :, -1] = 0                                   # the zero border
        xs = xs.to(torch.bfloat16).cuda()
        dy = torch.randn(N, H, W, Cout, generator=g).to(torch.bfloat16).cuda()
        dw = torch.empty(Cout, K, 2, 2, device="cuda")
        db = torch.empty(Cout, device="cuda")
        ops.profile_begin()
        T.conv_wgrad(ops.Act(dy, Cout), ops.Act(xs, K), 0, 0, K, 2, 2, 1, 1, dw, name="m0.wgrad", dbias=db)
        recs = ops.profile_end()
        torch.cuda.synchronize()
        assert not any("pad_width" in r[1] for r in recs), [r[1] for r in recs]
        w = torch.zeros(Cout, K, 2, 2, device="cuda", requires_grad=True)
        b = torch.zeros(Cout, device="cuda", requires_grad=True)
        y = F.conv2d(F.pad(xs.float().permute(0, 3, 1, 2), (1, 0, 1, 0)), w, b)
        y.backward(dy.float().permute(0, 3, 1, 2))
        assert float((dw - w.grad).abs().max()) <= 2e-3 * float(w.grad.abs().max()), float((dw - w.grad).abs().max())
        assert float((db - b.grad).abs().max()) <= 2e-3 * float(b.grad.abs().max())
    finally:
        T.MMA_BF16[0] = False


def test_pack_multi_equals_single_packs():
    """hrv_conv_s2_pack_multi_dev: the weight streams of several layers from one launch are the streams of the single packs, bit for bit."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import train_ops as T
    g = torch.Generator().manual_seed(5)
    sig = torch.tensor([1.3], device="cuda")
    w0 = torch.randn(64, 48, 2, 2, generator=g).cuda()
    w1 = torch.randn(128, 64, 4, 4, generator=g).cuda()
    w2 = torch.randn(256, 128, 4, 4, generator=g).cuda()
    jobs = [(T.S2_CELLS, w0, 48, 64, 0, None, False), (T.S2_FWD, w1, 64, 128, 0, sig, False), (T.S2_FWD, w2, 384, 256, 0, sig, True),
            (T.S2_DGRAD, w1, 128, 256, 64, sig, False), (T.S2_DGRAD, w2, 256, 512, 128, None, False), (T.S2_CELLS, w0, 144, 64, 0, sig, True),
            (T.S2_FWD, w1, 192, 128, 0, None, True), (T.S2_FWD, w2, 128, 256, 0, None, False), (T.S2_FWD, w1, 64, 128, 0, None, False)]
    got = T.conv_s2_pack_multi(jobs)
    torch.cuda.synchronize()
    assert len(got) == len(jobs)
    for (mode, w, K, cols, Cph, s_, sp), buf in zip(jobs, got):
        want = T.conv_s2_pack(mode, w, K, cols, Cph, sigma=s_, split3=sp)
        torch.cuda.synchronize()
        assert torch.equal(buf.view(torch.int16), want.view(torch.int16)), (mode, K, cols)


@pytest.mark.parametrize("Cin,Cout,N,H,W,cs_mult", [(64, 128, 2, 193, 161, 1), (128, 256, 2, 129, 129, 1), (64, 128, 1, 257, 193, 3),
                                                     (128, 256, 1, 258, 194, 1), (64, 256, 1, 513, 385, 1)])
def test_stride2_weight_gradient_on_the_lds_dma_kernel(Cin, Cout, N, H, W, cs_mult):
    """csrc/wgrad_s2.hip: dW and db of the 4x4 stride-2 pad-2 layer over bf16-stored dY and X (X also as the hi third of a split
    tensor), odd and even extents, no width-padding copy of dY -- against autograd on the same bf16-rounded operands."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    T.MMA_BF16[0] = True
    try:
        g = torch.Generator().manual_seed(Cin + H)
        Ho, Wo = H // 2 + 1, W // 2 + 1
        xall = torch.randn(N, H, W, Cin * cs_mult, generator=g).to(torch.bfloat16).cuda()
        dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(torch.bfloat16).cuda()
        dw = torch.empty(Cout, Cin, 4, 4, device="cuda")
        db = torch.empty(Cout, device="cuda")
        ops.profile_begin()
        T.conv_wgrad(ops.Act(dy, Cout), ops.Act(xall, Cin, 0), 0, 0, Cin, 4, 4, 2, 2, dw, name="m1.wgrad", dbias=db)
        recs = ops.profile_end()
        torch.cuda.synchronize()
        assert not any("pad_width" in r[1] for r in recs), [r[1] for r in recs]
        w = torch.zeros(Cout, Cin, 4, 4, device="cuda", requires_grad=True)
        b = torch.zeros(Cout, device="cuda", requires_grad=True)
        y = F.conv2d(xall[..., :Cin].float().permute(0, 3, 1, 2), w, b, stride=2, padding=2)
        y.backward(dy.float().permute(0, 3, 1, 2))
        assert float((dw - w.grad).abs().max()) <= 2e-3 * float(w.grad.abs().max()), float((dw - w.grad).abs().max())
        assert float((db - b.grad).abs().max()) <= 2e-3 * float(b.grad.abs().max())
    finally:
        T.MMA_BF16[0] = False
