"""csrc/conv_p2.hip -- the two-blocks-per-CU 3x3 convolution over one bf16 source (32-channel chunks double-buffered in LDS) --
against plain torch on the same bf16-rounded operands: nn.Conv2d forward with bias + ReLU (VGG19, networks.py:201-233), the data
gradient of such a convolution with the ReLU mask of its input (mode 1), and the data gradient of the SPADE
(conv_gamma, conv_beta) pair over [dgamma | dbeta] (mode 2, network_generator.py:117-118); K of 32 .. 544 (1 .. 17 chunks),
128 / 256 / 64 / 192 columns (4-tile passes, a 2-tile pass, both) and column counts that end inside a 32-column tile (32, 36, 48, 80, 144,
272: 1- and 3-tile passes, bounded bias / mask / residual reads and stores), the residual of SPADEResBlock (x_s + dx, fp32 or bf16), extents that are not multiples of the 16x16 tile, more tiles
than resident blocks, bf16 and fp32 outputs, channel slices of wider tensors."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("Cin,Cout,N,H,W,out_bf16", [(64, 128, 1, 250, 270, True), (128, 256, 2, 40, 56, True), (256, 64, 1, 33, 47, False),
                                                      (96, 192, 1, 64, 48, True), (512, 128, 1, 24, 32, False), (144, 64, 1, 72, 88, False),
                                                      (272, 128, 1, 40, 48, False), (80, 64, 1, 33, 40, True),
                                                      (80, 32, 1, 64, 48, True), (144, 80, 1, 40, 56, True), (32, 36, 1, 33, 47, False),
                                                      (48, 48, 2, 32, 32, True), (64, 144, 1, 40, 40, False), (128, 272, 1, 24, 40, True),
                                                      (64, 64, 4, 256, 128, True), (64, 64, 1, 33, 47, True), (32, 64, 1, 40, 40, False)])
def test_forward_bias_relu_matches_torch(Cin, Cout, N, H, W, out_bf16):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(Cin + Cout)
    xall = torch.randn(N, H, W, Cin + 32, generator=g).to(torch.bfloat16).cuda()          # a channel slice of a wider tensor
    x = ops.Act(xall, Cin, 32)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    oall = torch.full((N, H, W, Cout + 16), 7.0, device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    out = ops.Act(oall, Cout, 8)
    T.conv_p2(x, T.conv_p2_pack(0, w, None, Cin, Cout), Cout, out, bias=b, act=ops.ACT_RELU, name="t")
    torch.cuda.synchronize()
    want = F.relu(F.conv2d(xall[..., 32:].float().permute(0, 3, 1, 2), _bf(w), b, padding=1)).permute(0, 2, 3, 1)
    got = oall[..., 8:8 + Cout].float()
    tol = (want.abs() * 2 ** -8 if out_bf16 else 0.0) + 2e-4 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())
    assert bool((oall[..., :8] == 7.0).all()) and bool((oall[..., 8 + Cout:] == 7.0).all())      # neighbours untouched
    if out_bf16:      # ReLU gives +0 (the bf16 max pool orders stored patterns)
        assert int((oall.view(torch.int16) == -32768).sum()) == 0


@pytest.mark.parametrize("Ck,Ccol,N,H,W,out_bf16,masked", [(128, 64, 1, 70, 50, True, True), (256, 128, 1, 48, 40, True, True),
                                                            (64, 128, 2, 32, 48, False, False), (128, 256, 1, 40, 24, True, True),
                                                            (64, 144, 1, 40, 56, True, True), (32, 80, 1, 48, 40, True, True),
                                                            (32, 48, 1, 33, 40, True, False), (128, 272, 1, 24, 32, True, True),
                                                            (64, 80, 2, 32, 32, False, True)])
def test_data_gradient_with_relu_mask_matches_torch(Ck, Ccol, N, H, W, out_bf16, masked):
    """dX = conv^T(dY) * relu'(x): the forward layer maps Ccol -> Ck channels (VGG19's backward, vgg.py)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(Ck * 3 + Ccol)
    dy_t = torch.randn(N, H, W, Ck, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(Ck, Ccol, 3, 3, generator=g) * 0.05).cuda()
    xin = torch.relu(torch.randn(N, H, W, Ccol, generator=g)).to(torch.bfloat16).cuda()
    out = ops.alloc(N, H, W, Ccol, "cuda", bf16=out_bf16)
    T.conv_p2(ops.Act(dy_t, Ck), T.conv_p2_pack(1, w, None, Ck, Ccol), Ccol, out, mask=ops.Act(xin, Ccol) if masked else None, mask_slope=0.0,
              name="t")
    torch.cuda.synchronize()
    want = F.conv_transpose2d(dy_t.float().permute(0, 3, 1, 2), _bf(w), padding=1).permute(0, 2, 3, 1)
    if masked:
        want = want * (xin.float() > 0)
    got = out.t[..., :Ccol].float()
    tol = (want.abs() * 2 ** -8 if out_bf16 else 0.0) + 3e-4 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())


@pytest.mark.parametrize("C_,N,H,W,cs_mult,out_bf16", [(80, 1, 250, 270, 3, True), (144, 1, 96, 112, 1, True), (32, 1, 40, 48, 1, False),
                                                         (272, 1, 24, 32, 2, True), (64, 2, 64, 80, 1, True)])
def test_pair_data_gradient_matches_torch(C_, N, H, W, cs_mult, out_bf16):
    """d(actv) = conv^T([dgamma | dbeta]) * relu'(actv) (the shapes of tests/test_gpu_spade_gb.py's data-gradient cases)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(C_)
    hid = 128
    actv_all = torch.relu(torch.randn(N, H, W, hid * cs_mult, generator=g)).to(torch.bfloat16).cuda()
    actv = ops.Act(actv_all, hid, hid * (cs_mult - 1))
    wg = (torch.randn(C_, hid, 3, 3, generator=g) * 0.03).cuda()
    wb = (torch.randn(C_, hid, 3, 3, generator=g) * 0.03).cuda()
    dgb_t = torch.randn(N, H, W, 2 * C_, generator=g).to(torch.bfloat16).cuda()
    dact_all = torch.full((N, H, W, hid * cs_mult), 7.0, device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    dact = ops.Act(dact_all, hid, hid * (cs_mult - 1))
    T.conv_p2(ops.Act(dgb_t, 2 * C_), T.conv_p2_pack(2, wg, wb, 2 * C_, hid), hid, dact, mask=actv, mask_slope=0.0, name="t")
    torch.cuda.synchronize()
    dy = dgb_t.float().permute(0, 3, 1, 2)
    want = (F.conv_transpose2d(dy[:, :C_], _bf(wg), padding=1) + F.conv_transpose2d(dy[:, C_:], _bf(wb), padding=1))
    want = (want * (actv.t[..., actv.coff:actv.coff + hid].float() > 0).permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    got = dact_all[..., dact.coff:dact.coff + hid].float()
    tol = (want.abs() * 2 ** -8 if out_bf16 else 0.0) + 3e-4 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())
    if cs_mult > 1:
        assert bool((dact_all[..., :dact.coff] == 7.0).all())


def test_three_channel_image_ends_match_torch():
    """VGG19 features.0 (networks.py:208: Conv2d(3, 64, 3, padding=1) + ReLU over the image) and its data gradient (64 -> 3): K / column
    counts far below a 32-wide tile -- the source stores its 3 channels padded to 8 (zeros), the gradient its 3 channels padded to 4
    (the pad lane receives 0)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(11)
    N, H, W = 2, 70, 90
    img = torch.randn(N, 3, H, W, generator=g).cuda()
    x = ops.to_nhwc(img, bf16=True)
    assert x.bf16 and x.C == 3 and x.cstride % 8 == 0
    w = (torch.randn(64, 3, 3, 3, generator=g) * 0.2).cuda()
    b = (torch.randn(64, generator=g) * 0.1).cuda()
    out = ops.alloc(N, H, W, 64, "cuda", bf16=True)
    T.conv_p2(x, T.conv_p2_pack(0, w, None, 3, 64), 64, out, bias=b, act=ops.ACT_RELU, name="t")
    torch.cuda.synchronize()
    want = F.relu(F.conv2d(_bf(img), _bf(w), b, padding=1)).permute(0, 2, 3, 1)
    got = out.t[..., :64].float()
    assert bool(((got - want).abs() <= want.abs() * 2 ** -8 + 2e-4 * float(want.abs().max())).all()), float((got - want).abs().max())
    dy = ops.Act(torch.randn(N, H, W, 64, generator=g).to(torch.bfloat16).cuda(), 64)
    dx = ops.Act(torch.full((N, H, W, 4), 7.0, device="cuda"), 3)
    T.conv_p2(dy, T.conv_p2_pack(1, w, None, 64, 3), 3, dx, name="t")
    torch.cuda.synchronize()
    wantd = F.conv_transpose2d(dy.t.float().permute(0, 3, 1, 2), _bf(w), padding=1).permute(0, 2, 3, 1)
    assert float((dx.t[..., :3] - wantd).abs().max()) <= 3e-4 * float(wantd.abs().max())
    assert bool((dx.t[..., 3] == 0).all())                  # the pad lane stays zero


def test_vgg_first_layer_routes_through_the_kernel(monkeypatch):
    """train_ops routes features.0 and its data gradient onto conv_p2 (HRV_CONV_P2_ODD=0: the thin kernel): same results to bf16 /
    accumulation-order noise."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    T.MMA_BF16[0] = True
    try:
        g = torch.Generator().manual_seed(12)
        N, H, W = 2, 512, 384
        img = torch.randn(N, 3, H, W, generator=g).cuda()
        w = (torch.randn(64, 3, 3, 3, generator=g) * 0.2).cuda()
        b = (torch.randn(64, generator=g) * 0.1).cuda()
        dy = ops.Act(torch.randn(N, H, W, 64, generator=g).to(torch.bfloat16).cuda(), 64)
        res = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("HRV_CONV_P2_ODD", flag)
            y = T.conv_forward_dev(w, [(ops.to_nhwc(img, bf16=True), 0)], 1, 1, shift=b, act=ops.ACT_RELU, out_bf16=True, name="vgg.features.0")
            dx = T.conv_dgrad(dy, w, H, W, 1, 1, out_bf16=True, name="vgg.features.0.dgrad")
            torch.cuda.synchronize()
            res[flag] = (y.t.float().clone(), dx.t[..., :3].float().clone())
        for a, b_ in zip(res["1"], res["0"]):
            assert float((a - b_).abs().max()) <= 2 ** -7 * float(b_.abs().max())
    finally:
        T.MMA_BF16[0] = False


@pytest.mark.parametrize("Cin,Cout,res_bf16,out_bf16,act", [(80, 64, False, False, 0), (48, 32, False, True, 2), (144, 128, True, False, 0),
                                                             (80, 80, False, True, 1)])
def test_forward_with_residual_matches_torch(Cin, Cout, res_bf16, out_bf16, act):
    """SPADEResBlock: out = act(x_s + conv_1(h) + bias) (network_generator.py:168-170; the last block's LeakyReLU rides along)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(7 * Cin + Cout)
    N, H, W = 2, 40, 56
    x = ops.Act(torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).cuda(), Cin)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).cuda()
    b = (torch.randn(Cout, generator=g) * 0.1).cuda()
    rall = torch.randn(N, H, W, Cout + 12, generator=g)
    rall = (rall.to(torch.bfloat16) if res_bf16 else rall).cuda()
    res = ops.Act(rall, Cout, 4)
    oall = torch.full((N, H, W, Cout + 16), 7.0, device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    out = ops.Act(oall, Cout, 8)
    T.conv_p2(x, T.conv_p2_pack(0, w, None, Cin, Cout), Cout, out, bias=b, act=act, slope=0.2, residual=res, name="t")
    torch.cuda.synchronize()
    want = F.conv2d(x.t.float().permute(0, 3, 1, 2), _bf(w), b, padding=1).permute(0, 2, 3, 1) + rall[..., 4:4 + Cout].float()
    want = F.relu(want) if act == 1 else (F.leaky_relu(want, 0.2) if act == 2 else want)
    got = oall[..., 8:8 + Cout].float()
    tol = (want.abs() * 2 ** -8 if out_bf16 else 0.0) + 2e-4 * float(want.abs().max())
    assert bool(((got - want).abs() <= tol).all()), float((got - want).abs().max())
    assert bool((oall[..., :8] == 7.0).all()) and bool((oall[..., 8 + Cout:] == 7.0).all())


def test_training_convs_route_through_the_kernel_and_match_the_generic_tiles(monkeypatch):
    """train_ops.conv_forward_dev / conv_dgrad pick the kernel for plain 3x3 bf16 layers with >= 2 tiles per CU (HRV_CONV_P2=0:
    the generic patch tiles); both agree to accumulation-order noise on a VGG-like layer."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    T.MMA_BF16[0] = True
    try:
        g = torch.Generator().manual_seed(5)
        N, H, W, Cin, Cout = 2, 256, 272, 128, 128                     # 2 x 16 x 17 = 544 tiles
        x = ops.Act(torch.relu(torch.randn(N, H, W, Cin, generator=g)).to(torch.bfloat16).cuda(), Cin)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03).cuda()
        b = (torch.randn(Cout, generator=g) * 0.1).cuda()
        dy = ops.Act(torch.randn(N, H, W, Cout, generator=g).to(torch.bfloat16).cuda(), Cout)
        res = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("HRV_CONV_P2", flag)
            y = T.conv_forward_dev(w, [(x, 0)], 1, 1, shift=b, act=ops.ACT_RELU, out_bf16=True, name="l")
            dx = T.conv_dgrad(dy, w, H, W, 1, 1, act_mask=x, slope=0.0, out_bf16=True, name="l.dgrad")
            torch.cuda.synchronize()
            res[flag] = (y.t.float().clone(), dx.t.float().clone())
        for a, b_ in zip(res["1"], res["0"]):
            assert float((a - b_).abs().max()) <= 2 ** -7 * float(b_.abs().max())
    finally:
        T.MMA_BF16[0] = False


def test_data_gradient_joins_a_second_gradient_behind_the_mask(monkeypatch):
    """conv_dgrad(add_after=...): dX = conv^T(dY) * relu'(x) + g (VGG19's tap gradients meet the gradient flowing down through the tap
    in the epilogue of the data gradient above it; HRV_DGRAD_ADD_AFTER=0: a separate add_slice pass over the bf16 result)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    T.MMA_BF16[0] = True
    try:
        g = torch.Generator().manual_seed(21)
        N, H, W, K, Ccol = 2, 256, 272, 128, 64
        dy = ops.Act(torch.randn(N, H, W, K, generator=g).to(torch.bfloat16).cuda(), K)
        w = (torch.randn(K, Ccol, 3, 3, generator=g) * 0.05).cuda()
        x = ops.Act(torch.relu(torch.randn(N, H, W, Ccol, generator=g)).to(torch.bfloat16).cuda(), Ccol)
        gi = ops.Act(torch.randn(N, H, W, Ccol, generator=g).to(torch.bfloat16).cuda(), Ccol)
        conv = F.conv_transpose2d(dy.t.float().permute(0, 3, 1, 2), _bf(w), padding=1).permute(0, 2, 3, 1) * (x.t.float() > 0)
        want = conv + gi.t.float()
        for flag in ("1", "0"):
            monkeypatch.setenv("HRV_DGRAD_ADD_AFTER", flag)
            d = T.conv_dgrad(dy, w, H, W, 1, 1, act_mask=x, slope=0.0, out_bf16=True, name="l.dgrad", add_after=gi)
            torch.cuda.synchronize()
            got = d.t[..., :Ccol].float()
            # one rounding of the sum (fused) or two (the data gradient stored in bf16, then the sum: the first one is relative to the
            # gradient, which can cancel against g)
            tol = (conv.abs() + want.abs()) * 2 ** -8 + 3e-4 * float(want.abs().max())
            assert bool(((got - want).abs() <= tol).all()), (flag, float((got - want).abs().max()))
    finally:
        T.MMA_BF16[0] = False
