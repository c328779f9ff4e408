"""The fused SPADENorm forward (csrc/spade_fused.hip; network_generator.py:93-121): conv_shared + ReLU computed INSIDE the
gamma|beta kernel from the 8-channel label patch, modulate epilogue, optional actv / (1 + gamma) side outputs -- against plain
torch on the same bf16-rounded operands (label map, the three weights; actv rounded to bf16 where the kernel rounds it), at
the channel counts of the generator's blocks: 80 (one 5-tile pass), 144 (4 + 5: two launches), 64 / 128 (4-tile passes), 32 /
96 (a 2-tile pass), 272 (4 x 4 + 2: three passes in one launch + ...), on extents that are not multiples of the 16x16 tile,
with more tiles than resident blocks (persistent loop), and with the label map at 1x / 2x / 4x the level's resolution
(F.interpolate(segmap, size, 'nearest') read in place)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(N, H, W, C_, shift, seed, onehot=True, label_nc=7):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    Hs, Ws = H << shift, W << shift
    if onehot:
        lab = torch.randint(0, label_nc, (N, Hs, Ws, 1), generator=g)
        seg = torch.zeros(N, Hs, Ws, 8).scatter_(3, lab, 1.0)
    else:
        seg = torch.randn(N, Hs, Ws, 8, generator=g)
        seg[..., label_nc:] = 0
    seg = seg.to(torch.bfloat16).to(dev)
    wsh = (torch.randn(128, label_nc, 3, 3, generator=g) * 0.3).to(dev)
    bsh = (torch.randn(128, generator=g) * 0.2).to(dev)
    wg = (torch.randn(C_, 128, 3, 3, generator=g) * 0.03).to(dev)
    wb = (torch.randn(C_, 128, 3, 3, generator=g) * 0.03).to(dev)
    return ops, ops.Act(seg, label_nc), wsh, bsh, wg, wb, g, dev


def _ref_actv(seg, shift, wsh, bsh, label_nc=7):
    s = seg.t[:, ::(1 << shift), ::(1 << shift), :label_nc].float().permute(0, 3, 1, 2)      # nearest, power-of-two ratio
    a = F.relu(F.conv2d(s, wsh.to(torch.bfloat16).float(), bsh, padding=1))
    return a.to(torch.bfloat16).float()                                                       # the kernel keeps actv in bf16


@pytest.mark.parametrize("C_,N,H,W,shift,save", [(80, 1, 250, 270, 0, True), (80, 2, 64, 48, 2, False), (144, 1, 96, 112, 1, True),
                                                   (64, 1, 128, 144, 0, True), (128, 1, 70, 50, 1, False), (272, 1, 40, 48, 2, True),
                                                   (32, 1, 64, 80, 0, True), (96, 1, 48, 64, 0, False)])
def test_fused_forward_matches_torch_on_bf16_rounded_operands(C_, N, H, W, shift, save):
    ops, seg, wsh, bsh, wg, wb, g, dev = _mk(N, H, W, C_, shift, 1)
    from hr_viton_amd import train_ops as T
    x = ops.Act(torch.randn(N, H, W, C_, generator=g).to(dev), C_)
    z = torch.randn(N, W, H, 1, generator=g).to(dev)
    ns = (torch.randn(C_, generator=g) * 0.1).to(dev)
    bg, bb = (torch.randn(C_, generator=g) * 0.1).to(dev), (torch.randn(C_, generator=g) * 0.1).to(dev)
    mean, rstd = torch.randn(N, C_, generator=g).to(dev) * 0.1, (torch.rand(N, C_, generator=g) + 0.5).to(dev)
    out = ops.alloc(N, H, W, C_, dev, bf16=True)
    g1p = torch.full((N, H, W, C_), 5.0, device=dev, dtype=torch.bfloat16)
    actv_all = torch.full((N, H, W, 384), 7.0, device=dev, dtype=torch.bfloat16)      # a slice of the block's side-by-side tensor
    actv = ops.Act(actv_all, 128, 128)
    pk = T.spade_fused_pack(wsh, bsh, wg, wb)
    T.spade_fused_forward(seg, shift, x, mean, rstd, z, ns, pk, bg, bb, ops.ACT_LRELU, 0.2, out, g1p if save else None,
                          actv if save else None, "t")
    torch.cuda.synchronize()
    a = _ref_actv(seg, shift, wsh, bsh)
    gam = F.conv2d(a, wg.to(torch.bfloat16).float(), bg, padding=1)
    bet = F.conv2d(a, wb.to(torch.bfloat16).float(), bb, padding=1)
    xn = x.t.permute(0, 3, 1, 2) + z.permute(0, 3, 2, 1) * ns.view(1, -1, 1, 1)
    xn = (xn - mean.view(N, C_, 1, 1)) * rstd.view(N, C_, 1, 1)
    want = F.leaky_relu(xn * (1 + gam) + bet, 0.2).permute(0, 2, 3, 1)
    got = out.t[..., :C_].float()
    # bf16 result: half an ulp of the stored value (2^-9 relative) + accumulation-order noise (incl. the rare actv element
    # whose bf16 rounding flips with the summation order of conv_shared)
    err = (got - want).abs()
    assert float((err / (want.abs() * 2 ** -8 + 2e-3)).max()) < 1.0, float(err.max())
    if save:
        g1w = (1 + gam).permute(0, 2, 3, 1)
        assert float(((g1p.float() - g1w).abs() / (g1w.abs() * 2 ** -8 + 1e-3)).max()) < 1.0
        # actv: conv_shared sums <= 9 exact products per output (one-hot labels) -> bit-identical up to the summation
        # order: at most one bf16 ulp, on a vanishing fraction of the elements
        aw = a.permute(0, 2, 3, 1)
        ag = actv_all[..., 128:256].float()
        d = (ag - aw).abs()
        assert float((d / (aw.abs() * 2 ** -7 + 1e-6)).max()) < 1.0 and float((d > 0).float().mean()) < 1e-3, (float(d.max()), float((d > 0).float().mean()))
        assert bool((actv_all[..., :128] == 7.0).all()) and bool((actv_all[..., 256:] == 7.0).all())      # neighbours untouched
    else:
        assert bool((g1p == 5.0).all()) and bool((actv_all == 7.0).all())


@pytest.mark.parametrize("x_bf16,noise,act", [(True, False, "none"), (False, True, "none"), (True, True, "lrelu")])
def test_fused_forward_input_types_and_general_label_values(x_bf16, noise, act):
    """x bf16-stored (inference) or fp32 (training), with / without the noise term, no activation (norm_s); the label map holds
    arbitrary bf16 values (the kernel does not rely on one-hot inputs)."""
    C_, N, H, W, shift = 80, 1, 48, 64, 0
    ops, seg, wsh, bsh, wg, wb, g, dev = _mk(N, H, W, C_, shift, 2, onehot=False)
    from hr_viton_amd import train_ops as T
    xb = torch.randn(N, H, W, C_, generator=g).to(dev)
    x = ops.Act(xb.to(torch.bfloat16) if x_bf16 else xb, C_)
    z = torch.randn(N, W, H, 1, generator=g).to(dev) if noise else None
    ns = (torch.randn(C_, generator=g) * 0.1).to(dev)
    bg, bb = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    mean, rstd = torch.zeros(N, C_, device=dev), torch.ones(N, C_, device=dev)
    out = ops.alloc(N, H, W, C_, dev, bf16=True)
    a_code = ops.ACT_LRELU if act == "lrelu" else ops.ACT_NONE
    T.spade_fused_forward(seg, shift, x, mean, rstd, z, ns if noise else None, T.spade_fused_pack(wsh, bsh, wg, wb), bg, bb, a_code, 0.2,
                          out, None, None, "t")
    torch.cuda.synchronize()
    a = _ref_actv(seg, shift, wsh, bsh)
    gam = F.conv2d(a, wg.to(torch.bfloat16).float(), padding=1)
    bet = F.conv2d(a, wb.to(torch.bfloat16).float(), padding=1)
    xn = x.t.float().permute(0, 3, 1, 2)
    if noise:
        xn = xn + z.permute(0, 3, 2, 1) * ns.view(1, -1, 1, 1)
    want = xn * (1 + gam) + bet
    if act == "lrelu":
        want = F.leaky_relu(want, 0.2)
    want = want.permute(0, 2, 3, 1)
    # (general label values: an actv element near a bf16 rounding boundary may round the other way -> 2^-8 of |actv| x one weight)
    assert float(((out.t[..., :C_].float() - want).abs() / (want.abs() * 2 ** -8 + 4e-3)).max()) < 1.0


def test_block_through_the_training_plan_fused_vs_unfused(monkeypatch):
    """BlockT.forward / backward with conv_shared inside the gamma|beta kernels (HRV_SPADE_FUSED=1, the default where the
    level has >= 2 tiles per CU) against the unfused plan (thin conv_shared launch + spade_gb.hip): outputs, saved actv and
    every parameter gradient of the block agree to accumulation-order noise; the fused forward launches no conv_shared
    convolution, the no_grad forward writes no actv."""
    from argparse import Namespace
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import gen_train, ops, train_ops as T
    from hr_viton_amd.network_generator import SPADEResBlock
    T.MMA_BF16[0] = True
    try:
        torch.manual_seed(0)
        N, H, W = 1, 384, 352                                  # 24 x 22 = 528 tiles >= 2 per CU
        opt = Namespace(norm_G="spectralaliasinstance", gen_semantic_nc=7)
        blk = SPADEResBlock(opt, 144, 64, use_mask_norm=False).cuda()      # up_3's widths: norm_s / norm_0 over 144, norm_1 over 64
        with torch.no_grad():
            for n_, p_ in blk.named_parameters():
                if n_.endswith("noise_scale"):
                    p_.normal_(0, 0.1)
                elif "conv_gamma.weight" in n_ or "conv_beta.weight" in n_:
                    p_.mul_(4.0)
        bt = gen_train.BlockT(blk, "up_3")
        lab = torch.randint(0, 7, (N, 2 * H, 2 * W, 1), device="cuda")
        seg = ops.Act(torch.zeros(N, 2 * H, 2 * W, 8, device="cuda").scatter_(3, lab, 1.0), 7)
        x = ops.Act(torch.randn(N, H, W, 144, device="cuda"), 144)
        zs = [torch.randn(N, W, H, 1, device="cuda") for _ in range(3)]
        dout = ops.Act(torch.randn(N, H, W, 64, device="cuda") * 0.1, 64)
        res = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("HRV_SPADE_FUSED", flag)
            for p_ in blk.parameters():
                p_.grad = None
            T.prepare_convs(bt, bt.convs() + [n_.shared for n_ in bt.norms()], False)      # (no power iteration: the same sigma in both runs)
            ops.profile_begin()
            o, ctx = bt.forward(x, seg, 1, zs, None, 0, ops.ACT_NONE, save=True)
            recs = ops.profile_end()
            grads = {}
            d_x = bt.backward(ctx, dout, grads)
            torch.cuda.synchronize()
            gd = {n_: grads[p_].detach().float().clone() if p_ in grads else None for n_, p_ in blk.named_parameters()}
            res[flag] = (o.t.float().clone(), d_x.t.float().clone(), gd, [r[1] for r in recs if r[0] == "conv"],
                         ctx["n0"]["actv"].t[..., ctx["n0"]["actv"].coff:ctx["n0"]["actv"].coff + 128].float().clone())
        (of, dxf, gf, names_f, af), (ou, dxu, gu, names_u, au) = res["1"], res["0"]
        assert any("conv_shared+gamma|beta" in n_ for n_ in names_f) and not any("as 1x1 over taps" in n_ for n_ in names_f), names_f
        assert any("as 1x1 over taps" in n_ for n_ in names_u), names_u
        assert float((af - au).abs().max()) <= 2 ** -7 * float(au.abs().max()) and float(((af - au).abs() > 0).float().mean()) < 1e-3
        assert float((of - ou).abs().max()) <= 2 ** -6 * float(ou.abs().max()), float((of - ou).abs().max())
        assert float((dxf - dxu).abs().max()) <= 2e-2 * float(dxu.abs().max())
        for n_ in gu:
            if gu[n_] is None:
                assert gf[n_] is None, n_
                continue
            cos = float(F.cosine_similarity(gf[n_].flatten(), gu[n_].flatten(), dim=0)) if gu[n_].numel() > 1 else 1.0
            assert cos > 0.9995, (n_, cos)
        # the no_grad forward (discriminator step): nothing saved, no actv tensor, same output
        monkeypatch.setenv("HRV_SPADE_FUSED", "1")
        o2, ctx2 = bt.forward(x, seg, 1, zs, None, 0, ops.ACT_NONE, save=False)
        torch.cuda.synchronize()
        assert ctx2["segx"] is None and ctx2["n0"]["actv"] is None
        assert float((o2.t.float() - of).abs().max()) == 0.0
    finally:
        T.MMA_BF16[0] = False


def test_block_reads_the_upsampled_input_in_place_bit_identically():
    """ops.ActUp: x = cat(nearest_up2(previous block's output), stem) is never materialised -- the one-pass statistics of norm_s /
    norm_0, the fused SPADE forward and the normalisation backward address the low-resolution tensor at (y >> 1, x >> 1)
    (network_generator.py:203,226-242).  Same values in the same order: the block's output, the gradient of its input and every
    parameter gradient are BIT-IDENTICAL to the run on the materialised 4-fold copy."""
    from argparse import Namespace
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import gen_train, ops, train_ops as T
    from hr_viton_amd.network_generator import SPADEResBlock
    T.MMA_BF16[0] = True
    try:
        torch.manual_seed(1)
        N, H, W = 1, 384, 352
        opt = Namespace(norm_G="spectralaliasinstance", gen_semantic_nc=7)
        blk = SPADEResBlock(opt, 144, 64, use_mask_norm=False).cuda()
        with torch.no_grad():
            for n_, p_ in blk.named_parameters():
                if n_.endswith("noise_scale"):
                    p_.normal_(0, 0.1)
                elif "conv_gamma.weight" in n_ or "conv_beta.weight" in n_:
                    p_.mul_(4.0)
        bt = gen_train.BlockT(blk, "up_3")
        lab = torch.randint(0, 7, (N, 2 * H, 2 * W, 1), device="cuda")
        seg = ops.Act(torch.zeros(N, 2 * H, 2 * W, 8, device="cuda").scatter_(3, lab, 1.0), 7)
        lo = ops.Act(torch.randn(N, H // 2, W // 2, 128, device="cuda"), 128)
        hi = ops.Act(torch.randn(N, H, W, 16, device="cuda"), 16)
        full = ops.Act(torch.cat([lo.t.repeat_interleave(2, 1).repeat_interleave(2, 2), hi.t], 3).contiguous(), 144)
        zs = [torch.randn(N, W, H, 1, device="cuda") for _ in range(3)]
        dout = ops.Act(torch.randn(N, H, W, 64, device="cuda") * 0.1, 64)
        assert bt.reads_upsampled_input(N, H, W, 144, seg, 1)
        res = []
        for x in (ops.ActUp(lo, hi), full):
            for p_ in blk.parameters():
                p_.grad = None
            T.prepare_convs(bt, bt.convs() + [n_.shared for n_ in bt.norms()], False)
            o, ctx = bt.forward(x, seg, 1, zs, None, 0, ops.ACT_NONE, save=True)
            grads = {}
            d_x = bt.backward(ctx, dout, grads)
            torch.cuda.synchronize()
            res.append((o.t.clone(), d_x.t.clone(), {n_: grads[p_].detach().clone() for n_, p_ in blk.named_parameters() if p_ in grads},
                        ctx["n0"]["mean"].clone(), ctx["ns"]["rstd"].clone()))
        a, b = res
        assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]), "statistics"
        assert torch.equal(a[0], b[0]), "block output"
        assert torch.equal(a[1], b[1]), "gradient of the block input"
        assert a[2].keys() == b[2].keys()
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), k
    finally:
        T.MMA_BF16[0] = False


@pytest.mark.parametrize("up", [False, True], ids=["materialised_x", "upsampled_in_place"])
def test_one_pass_backward_of_the_two_norms_over_the_block_input_is_bit_identical(monkeypatch, up):
    """hrv_spade_norm_bwd2_nhwc_f32: norm_0 and norm_s of a learned-shortcut block normalise the same x -- one pass per stage over
    it, dx = dx_0 + dx_s written once.  Same values, one commutative fp32 add: the gradient of the block input and every
    parameter gradient are bit-identical to the two sequential calls (HRV_NORM_BWD2=0, the default: the one-pass form measured
    slower, see gen_train.BlockT.backward)."""
    from argparse import Namespace
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import gen_train, ops, train_ops as T
    from hr_viton_amd.network_generator import SPADEResBlock
    T.MMA_BF16[0] = True
    try:
        torch.manual_seed(2)
        N, H, W = 1, 384, 352
        opt = Namespace(norm_G="spectralaliasinstance", gen_semantic_nc=7)
        blk = SPADEResBlock(opt, 80, 32, use_mask_norm=False).cuda()
        with torch.no_grad():
            for n_, p_ in blk.named_parameters():
                if n_.endswith("noise_scale"):
                    p_.normal_(0, 0.1)
        bt = gen_train.BlockT(blk, "up_4")
        lab = torch.randint(0, 7, (N, H, W, 1), device="cuda")
        seg = ops.Act(torch.zeros(N, H, W, 8, device="cuda").scatter_(3, lab, 1.0), 7)
        if up:
            x = ops.ActUp(ops.Act(torch.randn(N, H // 2, W // 2, 64, device="cuda"), 64), ops.Act(torch.randn(N, H, W, 16, device="cuda"), 16))
        else:
            x = ops.Act(torch.randn(N, H, W, 80, device="cuda"), 80)
        zs = [torch.randn(N, W, H, 1, device="cuda") for _ in range(3)]
        dout = ops.Act(torch.randn(N, H, W, 32, device="cuda") * 0.1, 32)
        res = []
        # (both arms with the data gradients in their own tensors: the in-place dbeta form of round 6 rounds dbeta once instead of
        #  twice and is switched off by HRV_NORM_BWD2=1 anyway -- this test compares the two normalisation-backward forms only)
        monkeypatch.setenv("HRV_DBETA_INPLACE", "0")
        for flag in ("1", "0"):
            monkeypatch.setenv("HRV_NORM_BWD2", flag)
            for p_ in blk.parameters():
                p_.grad = None
            T.prepare_convs(bt, bt.convs() + [n_.shared for n_ in bt.norms()], False)
            ops.profile_begin()
            o, ctx = bt.forward(x, seg, 0, zs, None, 0, ops.ACT_NONE, save=True)
            grads = {}
            d_x = bt.backward(ctx, dout, grads)
            recs = ops.profile_end()
            torch.cuda.synchronize()
            res.append((d_x.t.clone(), {n_: grads[p_].detach().clone() for n_, p_ in blk.named_parameters() if p_ in grads},
                        [r[1] for r in recs if r[0] == "norm_bwd"]))
        a, b = res
        assert a[2].count("spade_norm_bwd x2") == 1 and a[2].count("spade_norm_bwd") == 1 and b[2].count("spade_norm_bwd") == 3, (a[2], b[2])
        assert torch.equal(a[0], b[0]), "gradient of the block input"
        assert a[1].keys() == b[1].keys()
        for k in a[1]:
            assert torch.equal(a[1][k], b[1][k]), k
    finally:
        T.MMA_BF16[0] = False
