"""The dedicated SPADE gamma|beta kernel (csrc/spade_gb.hip; network_generator.py:117-121 + the modulate of :120-121 and
the LeakyReLU of :170-171) against plain torch on the same bf16-rounded operands: forward (SPADE epilogue, (1 + gamma)
side output) and data gradient (ReLU mask of actv), at the channel counts of the generator's blocks -- 80 (5 column tiles:
two pairs + the 16-channel tail), 144 (two launches: 4-tile passes + the 5-tile tail pass), 64 / 128 (pairs only), 32 / 96
(a 2-tile pass) -- on
extents that are not multiples of the 16x16 tile and with more tiles than CUs (persistent loop)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mk(N, H, W, C_, seed, cs_mult=1):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops
    g = torch.Generator().manual_seed(seed)
    hid = 128
    dev = "cuda"
    # actv as a channel slice of a wider tensor (the plan keeps the block's three actv tensors side by side)
    actv_all = torch.relu(torch.randn(N, H, W, hid * cs_mult, generator=g)).to(torch.bfloat16).to(dev)
    actv = ops.Act(actv_all, hid, hid * (cs_mult - 1))
    wg = (torch.randn(C_, hid, 3, 3, generator=g) * 0.03).to(dev)
    wb = (torch.randn(C_, hid, 3, 3, generator=g) * 0.03).to(dev)
    return ops, actv, wg, wb, g, dev, hid


@pytest.mark.parametrize("C_,N,H,W,cs_mult", [(80, 1, 250, 270, 3), (144, 2, 96, 112, 1), (64, 1, 128, 144, 2), (128, 1, 70, 50, 1),
                                                (272, 1, 40, 48, 1), (32, 1, 64, 80, 1), (96, 1, 48, 64, 2)])
def test_forward_matches_torch_on_bf16_rounded_operands(C_, N, H, W, cs_mult):
    ops, actv, wg, wb, g, dev, hid = _mk(N, H, W, C_, 1, cs_mult)
    from hr_viton_amd import train_ops as T
    x = ops.Act(torch.randn(N, H, W, C_, generator=g).to(dev), C_)
    z = torch.randn(N, W, H, 1, generator=g).to(dev)
    ns = (torch.randn(C_, generator=g) * 0.1).to(dev)
    bg, bb = (torch.randn(C_, generator=g) * 0.1).to(dev), (torch.randn(C_, generator=g) * 0.1).to(dev)
    mean, rstd = torch.randn(N, C_, generator=g).to(dev) * 0.1, (torch.rand(N, C_, generator=g) + 0.5).to(dev)
    out = ops.alloc(N, H, W, C_, dev, bf16=True)
    g1p = torch.empty(N, H, W, C_, device=dev, dtype=torch.bfloat16)     # (1 + gamma) is stored in bf16
    pk = T.spade_gb_pack(0, wg, wb)
    T.spade_gb_forward(actv, x, mean, rstd, z, ns, pk, bg, bb, ops.ACT_LRELU, 0.2, out, g1p, "t", 1.0, 1.0)
    torch.cuda.synchronize()
    a = actv.t[..., actv.coff:actv.coff + hid].float().permute(0, 3, 1, 2)
    gam = F.conv2d(a, wg.to(torch.bfloat16).float(), bg, padding=1)
    bet = F.conv2d(a, wb.to(torch.bfloat16).float(), bb, padding=1)
    xn = x.t.permute(0, 3, 1, 2) + z.permute(0, 3, 2, 1) * ns.view(1, -1, 1, 1)
    xn = (xn - mean.view(N, C_, 1, 1)) * rstd.view(N, C_, 1, 1)
    want = F.leaky_relu(xn * (1 + gam) + bet, 0.2).permute(0, 2, 3, 1)
    got = out.t[..., :C_].float()
    # bf16 result: half an ulp of the stored value (2^-9 relative) + accumulation-order noise
    err = (got - want).abs()
    assert float((err / (want.abs() * 2 ** -8 + 2e-3)).max()) < 1.0, float(err.max())
    g1w = (1 + gam).permute(0, 2, 3, 1)
    assert float(((g1p.float() - g1w).abs() / (g1w.abs() * 2 ** -8 + 1e-3)).max()) < 1.0


@pytest.mark.parametrize("x_bf16,noise,save", [(True, False, True), (False, True, False)])
def test_forward_input_types(x_bf16, noise, save):
    C_, N, H, W = 80, 1, 48, 64
    ops, actv, wg, wb, g, dev, hid = _mk(N, H, W, C_, 2)
    from hr_viton_amd import train_ops as T
    xb = torch.randn(N, H, W, C_, generator=g).to(dev)
    x = ops.Act(xb.to(torch.bfloat16) if x_bf16 else xb, C_)       # x may be bf16-stored (inference) or fp32 (training)
    z = torch.randn(N, W, H, 1, generator=g).to(dev) if noise else None
    ns = (torch.randn(C_, generator=g) * 0.1).to(dev)
    bg, bb = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
    mean, rstd = torch.zeros(N, C_, device=dev), torch.ones(N, C_, device=dev)
    out = ops.alloc(N, H, W, C_, dev, bf16=True)
    g1p = torch.full((N, H, W, C_), 5.0, device=dev, dtype=torch.bfloat16)
    T.spade_gb_forward(actv, x, mean, rstd, z, ns if noise else None, T.spade_gb_pack(0, wg, wb), bg, bb, ops.ACT_NONE, 0.2, out,
                       g1p if save else None, "t", 1.0, 1.0)
    torch.cuda.synchronize()
    a = actv.t.float().permute(0, 3, 1, 2)
    gam = F.conv2d(a, wg.to(torch.bfloat16).float(), padding=1)
    bet = F.conv2d(a, wb.to(torch.bfloat16).float(), padding=1)
    xn = x.t.float().permute(0, 3, 1, 2)
    if noise:
        xn = xn + z.permute(0, 3, 2, 1) * ns.view(1, -1, 1, 1)
    want = (xn * (1 + gam) + bet).permute(0, 2, 3, 1)
    assert float(((out.t[..., :C_].float() - want).abs() / (want.abs() * 2 ** -8 + 2e-3)).max()) < 1.0
    g1w = (1 + gam).permute(0, 2, 3, 1)
    if save:
        assert float(((g1p.float() - g1w).abs() / (g1w.abs() * 2 ** -8 + 1e-3)).max()) < 1.0
    else:
        assert bool((g1p == 5.0).all())


@pytest.mark.parametrize("out_bf16", [False, True], ids=["f32out", "bf16out"])
@pytest.mark.parametrize("C_,N,H,W,cs_mult", [(80, 1, 250, 270, 3), (144, 1, 96, 112, 1), (64, 1, 64, 80, 2), (32, 1, 40, 48, 1),
                                                (272, 1, 24, 32, 1)])
def test_data_gradient_matches_torch(C_, N, H, W, cs_mult, out_bf16):
    ops, actv, wg, wb, g, dev, hid = _mk(N, H, W, C_, 3, cs_mult)
    from hr_viton_amd import train_ops as T
    dgb_t = torch.randn(N, H, W, 2 * C_, generator=g).to(torch.bfloat16).to(dev)
    dgb = ops.Act(dgb_t, 2 * C_)
    dact_all = torch.full((N, H, W, hid * cs_mult), 7.0, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    dact = ops.Act(dact_all, hid, hid * (cs_mult - 1))
    T.spade_gb_dgrad(dgb, T.spade_gb_pack(1, wg, wb), C_, actv, 0.0, dact, "t")
    torch.cuda.synchronize()
    dy = dgb_t.float().permute(0, 3, 1, 2)
    want = (F.conv_transpose2d(dy[:, :C_], wg.to(torch.bfloat16).float(), padding=1) +
            F.conv_transpose2d(dy[:, C_:], wb.to(torch.bfloat16).float(), padding=1))
    m = (actv.t[..., actv.coff:actv.coff + hid].float() > 0).permute(0, 3, 1, 2)
    want = (want * m).permute(0, 2, 3, 1)
    got = dact_all[..., dact.coff:dact.coff + hid].float()
    if out_bf16:       # one bf16 rounding of the stored value
        assert float(((got - want).abs() / (want.abs() * 2 ** -8 + 3e-4 * float(want.abs().max()))).max()) < 1.0
    else:
        assert float((got - want).abs().max()) < 3e-4 * float(want.abs().max())
    if cs_mult > 1:        # the neighbouring slices are untouched
        assert bool((dact_all[..., :dact.coff] == 7.0).all())


def test_spade_layer_through_the_training_plan_uses_the_kernel_and_matches_the_generic_tiles(monkeypatch):
    """SpadeT.forward / backward route the 80-channel norm of up_4 through the dedicated kernel; same layer with
    HRV_SPADE_GB=0 (generic patch tiles) must agree to accumulation-order noise."""
    from argparse import Namespace
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops, train_ops as T
    from hr_viton_amd.gen_train import SpadeT
    from hr_viton_amd.network_generator import SPADENorm
    T.MMA_BF16[0] = True
    try:
        torch.manual_seed(0)
        N, H, W, Cc = 1, 256, 272, 80
        norm = SPADENorm(Namespace(), "aliasinstance", Cc, 7).cuda()
        with torch.no_grad():
            norm.noise_scale.normal_(0, 0.1)
        st = SpadeT(norm, ops.ACT_LRELU, "up_4.norm_0")
        x = ops.Act(torch.randn(N, H, W, Cc, device="cuda"), Cc)
        actv = ops.Act(torch.relu(torch.randn(N, H, W, 128, device="cuda")).to(torch.bfloat16), 128)
        z = torch.randn(N, W, H, 1, device="cuda")
        res = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("HRV_SPADE_GB", flag)
            ops.profile_begin()
            out, ctx = st.forward(x, actv, z, save=True)
            recs = ops.profile_end()
            res[flag] = (out.t.float().clone(), ctx["g1p"].t.clone(), [r[1] for r in recs if r[0] == "conv"])
        torch.cuda.synchronize()
        a, b = res["1"], res["0"]
        assert float((a[0] - b[0]).abs().max()) <= 2 ** -7 * float(b[0].abs().max())
        assert float((a[1].float() - b[1].float()).abs().max()) < 2 ** -7 * float(b[1].abs().max())
    finally:
        T.MMA_BF16[0] = False
