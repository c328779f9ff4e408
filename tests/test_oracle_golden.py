"""CPU: the oracle restatement vs golden vectors produced by the REAL reference
(oracle/make_golden.py imports /root/reference).  This is what pins the oracle."""
import torch

from conftest import load_golden
from oracle import hrviton_oracle as O


def _close(a, b, tol=2e-5):
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * max(ref, 1.0), f"max err {err} vs max |ref| {ref}"


def test_primitives():
    g = load_golden("primitives.pt")
    _close(O.grid_sample_bilinear_border(g["gs_in"], g["gs_grid"]), g["gs_out"], 1e-6)
    _close(O.resize_bilinear(g["bil_in"], scale_factor=2), g["bil_x2"], 1e-6)
    _close(O.resize_bilinear(g["bil_in"], size=(13, 9)), g["bil_size"], 1e-6)
    assert torch.equal(O.resize_nearest(g["bil_in"], (10, 12)), g["near_size"])
    # torch.linspace's vectorised CPU kernel differs from the scalar formula by <=1 ulp
    # (lane-strided base + lane*step); the restatement is held to 1 ulp of 1.0.
    assert (O.make_grid(2, 7, 5) - g["grid_7x5"]).abs().max() <= 1.2e-7
    assert (O.make_grid(1, 24, 18) - g["grid_24x18"]).abs().max() <= 1.2e-7
    _close(O.instance_norm(g["in_in"]), g["in_out"], 1e-5)


def test_tocg_matches_reference():
    g = load_golden("tocg_ngf8_96x64.pt")
    flow_list, seg, wc, wcm = O.tocg_forward(g["state_dict"], g["input1"], g["input2"])
    for a, b in zip(flow_list, g["flow_list"]):
        assert a.shape == b.shape
        _close(a, b)
    _close(seg, g["seg"])
    # white-noise images amplify 1e-6-px flow differences: |dI/dx| ~ 2 per px
    _close(wc, g["warped_c"], 1e-4)
    _close(wcm, g["warped_cm"], 1e-4)
    # the flows are non-trivial (otherwise the warp is not exercised)
    assert g["flow_list"][-1].abs().max() > 0.5


def test_tocg_encoder_conv_variant_matches_reference():
    """warp_feature='encoder' (the decoder reads the warped cloth-encoder feature, networks.py:46-54,142-144) + out_layer='conv'
    (ResBlock + Conv2d 1x1, networks.py:57-61): the restatement against vectors of the real reference."""
    g = load_golden("tocg_encoder_conv_ngf8_96x64.pt")
    flow_list, seg, wc, wcm = O.tocg_forward(g["state_dict"], g["input1"], g["input2"], g["warp_feature"], g["out_layer"])
    for a, b in zip(flow_list, g["flow_list"]):
        assert a.shape == b.shape
        _close(a, b)
    _close(seg, g["seg"])
    _close(wc, g["warped_c"], 1e-4)
    _close(wcm, g["warped_cm"], 1e-4)
    assert g["flow_list"][-1].abs().max() > 0.5 and (g["seg"] < 0).any()      # flows are exercised; logits are not ReLU outputs
    # forward(..., upsample='nearest') (networks.py:98,130-133,150) on the same weights
    n = g["nearest"]
    flow_list, seg, wc, wcm = O.tocg_forward(g["state_dict"], g["input1"], g["input2"], g["warp_feature"], g["out_layer"], "nearest")
    for a, b in zip(flow_list, n["flow_list"]):
        _close(a, b)
    _close(seg, n["seg"])
    _close(wc, n["warped_c"], 1e-4)
    _close(wcm, n["warped_cm"], 1e-4)
    assert (n["seg"] - g["seg"]).abs().max() > 1e-2                          # the mode changes the result


def test_spade_generator_matches_reference():
    g = load_golden("gen_ngf2_256x128.pt")
    out = O.spade_generator_forward(g["state_dict"], g["x"], g["seg"], 256, 128, "most", noise=g["noise"])
    assert out.shape == g["out"].shape
    _close(out, g["out"], 5e-5)
    assert g["out"].std() > 1e-2


def test_gen_discriminator_matches_reference():
    g = load_golden("gend_ndf8_128x64.pt")
    out = O.gen_discriminator_forward(g["state_dict"], g["input"])
    assert len(out) == len(g["out"]) == 2
    for a_s, b_s in zip(out, g["out"]):
        assert len(a_s) == len(b_s) == 4
        for a, b in zip(a_s, b_s):
            assert a.shape == b.shape
            _close(a, b, 5e-5)


def test_training_step_restatement_matches_reference():
    """The oracle's training-mode restatement (spectral-norm power iteration, hinge + feature
    matching, autograd through it) against summaries of one REAL-reference training step."""
    import types, sys
    from oracle.recipes import trainstep_build
    g = load_golden("trainstep_ngf8_256x128.pt")
    # the recipe needs module classes only to create identically-initialised parameters: use
    # torch containers via the product mirrors (CPU construction is allowed; forward is not)
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.network_generator import MultiscaleDiscriminator, SPADEGenerator
    opt, gen, dis, x, seg, real, noise = trainstep_build(SPADEGenerator, MultiscaleDiscriminator)
    sd_g = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith(("weight_u", "weight_v")))
            for k, v in gen.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith(("weight_u", "weight_v")))
            for k, v in dis.state_dict().items()}
    O.SN_TRAIN["on"], O.SN_TRAIN["uv"] = True, {}
    try:
        fake = O.spade_generator_forward(sd_g, x, seg, 256, 128, "most", noise=noise)
        pred = O.gen_discriminator_forward(sd_d, torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0))
    finally:
        O.SN_TRAIN["on"] = False
    _close(fake, g["out"], 5e-5)
    pf, pr = O.split_fake_real(pred)
    l_gan, l_feat = O.hinge_loss(pf, True, False), O.feat_match_loss(pf, pr, 10.0)
    assert abs(l_gan.item() - g["l_gan"].item()) < 1e-5 and abs(l_feat.item() - g["l_feat"].item()) < 1e-4
    (l_gan + l_feat).backward()
    _close(O.SN_TRAIN["uv"]["up_0.conv_0"][0], g["u_after"]["up_0.conv_0"], 1e-5)
    _close(O.SN_TRAIN["uv"]["discriminator_0.model1.0.0"][0], g["u_after"]["D.discriminator_0.model1.0.0"], 1e-5)
    gmax = max(v[0] for v in g["grad_summary"].values())
    for name, (gm, gs, gas) in g["grad_summary"].items():
        t = (sd_d[name[2:]] if name.startswith("D.") else sd_g[name]).grad
        assert t is not None, name
        # sign() in the L1 feature-matching gradient makes tiny gradients noisy: scale-aware tolerance
        assert abs(t.abs().max().item() - gm) < 2e-2 * max(gm, 1e-3 * gmax), (name, t.abs().max().item(), gm)
        assert abs(t.abs().sum().item() - gas) < 2e-2 * max(gas, 1e-3 * gmax * t.numel()), (name, t.abs().sum().item(), gas)
    for name, want in g["sample_grads"].items():
        t = (sd_d[name[2:]] if name.startswith("D.") else sd_g[name]).grad
        assert (t - want).abs().max() < 1e-2 * max(want.abs().max().item(), 1e-3 * gmax), name


def _condstep_oracle():
    """Shared by the CPU pin below and the GPU parity test: the oracle's train_condition.py step on the
    recipe's weights (built through the product's parameter containers on the CPU)."""
    from oracle.recipes import condstep_build
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import networks
    opt, tocg, D, batch = condstep_build(networks.ConditionGenerator, networks.define_D)
    sd_g = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running_" not in k)
            for k, v in tocg.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    res = O.condition_train_losses(sd_g, sd_d, None, batch)
    return tocg, D, batch, sd_g, sd_d, res


def test_condition_training_step_restatement_matches_reference():
    """Oracle restatement of one train_condition.py iteration (batch-stat BatchNorm, flow warps, TV / CE /
    LSGAN / interflow losses, autograd through all of it) vs summaries of the REAL reference's step."""
    g = load_golden("condstep_ngf8_128x96.pt")
    tocg, D, batch, sd_g, sd_d, r = _condstep_oracle()
    for k, want in g["losses"].items():
        got = r[{"loss_G": "loss_G", "loss_D": "loss_D", "l1": "l1", "tv": "tv", "ce": "ce", "g_gan": "g_gan",
                 "d_fake": "d_fake", "d_real": "d_real"}[k]].item()
        assert abs(got - want) < 2e-5 * max(1.0, abs(want)), (k, got, want)
    _close(r["fake_segmap"][:, :, ::4, ::4], g["fake_segmap"], 2e-5)
    _close(r["flow_list"][-1], g["flow_last"], 2e-5)
    _close(r["warped_cm"][:, :, ::2, ::2], g["warped_cm"], 1e-4)
    assert g["flow_last"].abs().max() > 0.3        # the warps are exercised
    # generator gradients (sign() of the L1 terms makes them noisy at the 1e-3 level: scale-aware 1e-2)
    r["loss_G"].backward(retain_graph=True)
    gmax = max(v[0] for v in g["grad_summary_G"].values())
    for name, (gm, gs, gas) in g["grad_summary_G"].items():
        t = sd_g[name].grad
        assert t is not None, name
        assert abs(t.abs().max().item() - gm) < 1e-2 * max(gm, 1e-3 * gmax), (name, t.abs().max().item(), gm)
        assert abs(t.abs().sum().item() - gas) < 1e-2 * max(gas, 1e-3 * gmax * t.numel()), (name, gas)
    for name, want in g["sample_grads_G"].items():
        assert (sd_g[name].grad - want).abs().max() < 1e-2 * max(want.abs().max().item(), 1e-3 * gmax), name
    # discriminator gradients (optimizer_D.zero_grad() precedes loss_D.backward(): train_condition.py:284-286)
    for v in sd_d.values():
        v.grad = None
    r["loss_D"].backward()
    dmax = max(v[0] for v in g["grad_summary_D"].values())
    for name, (gm, gs, gas) in g["grad_summary_D"].items():
        t = sd_d[name].grad
        assert abs(t.abs().max().item() - gm) < 1e-3 * max(gm, 1e-3 * dmax), (name, t.abs().max().item(), gm)
    for name, want in g["sample_grads_D"].items():
        assert (sd_d[name].grad - want).abs().max() < 1e-3 * max(want.abs().max().item(), 1e-3 * dmax), name
    # running statistics after the step: momentum 0.1 on the recorded batch statistics
    sd0 = tocg.state_dict()
    for k, want in g["bn_after"].items():
        if k.endswith("num_batches_tracked"):
            assert int(want) == 1
            continue
        p, which = k.rsplit(".", 1)
        mean, var_unb = r["bn_stats"][p]
        new = 0.9 * sd0[k] + 0.1 * (mean if which == "running_mean" else var_unb)
        _close(new, want, 1e-5)
