"""GPU: parity at BASELINE sizes (the small golden fixtures do not reach the tile paths the released configuration
uses: 1040/528/272-channel gamma|beta convolutions, 64-column patch tiles, sub-batch launches).

* SPADEGenerator forward, ngf=64 'most' 1024x768 (network_generator.py:221-245), fp32 engine vs the oracle
  (stated fp32 tolerance 1e-3 rel, north star) and bf16 engine vs the oracle with the same rounding points
  (oracle.QUANT) -- one CPU forward each (~10 s).
* The generator half of one train_generator.py iteration (:279-322) at 512x384 ngf=64 vs torch autograd over the
  oracle: image, loss terms incl. VGG, every parameter gradient (table -> gpurun_out/).
"""
import os
from argparse import Namespace

import pytest
import torch

from oracle import hrviton_oracle as O
from oracle import step_check

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _gen(fp16):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.network_generator import SPADEGenerator
    opt = Namespace(cuda=True, norm_G="spectralaliasinstance", gen_semantic_nc=7, ngf=64, num_upsampling_layers="most",
                    fine_height=1024, fine_width=768, fp16=fp16)
    torch.manual_seed(0)
    m = SPADEGenerator(opt, 9)
    m.init_weights("xavier", 0.02)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("noise_scale"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n_.endswith("weight") or n_.endswith("weight_orig"):
                p.mul_(8.0)          # xavier(0.02) alone leaves gamma/beta ~0: the modulation would not be exercised
            elif n_.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    x = torch.rand(1, 9, 1024, 768, generator=g) * 2 - 1
    lab = torch.randint(0, 7, (1, 1, 64, 48), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3)
    seg = torch.zeros(1, 7, 1024, 768).scatter_(1, lab, 1.0)
    noise = {}
    for j, name in enumerate(m._blocks()):
        h, w = m.sh << j, m.sw << j
        k = 3 if getattr(m, name).learned_shortcut else 2
        noise[name] = [torch.randn(1, w, h, 1, generator=g) for _ in range(k)]
    return opt, m, x, seg, noise


def test_generator_forward_1024x768_ngf64_fp32_and_bf16_vs_oracle():
    opt, m, x, seg, noise = _gen(False)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        want = O.spade_generator_forward(sd, x, seg, 1024, 768, "most", noise=noise)
        O.QUANT["fn"] = lambda t: t.to(torch.bfloat16).to(torch.float32)
        try:
            want_bf = O.spade_generator_forward(sd, x, seg, 1024, 768, "most", noise=noise)
            # calibration: the bf16-rounded oracle against ITSELF with the input nudged by 1e-6 (relative) before the
            # same rounding points -- a value that sits on a bf16 rounding boundary flips by 2^-8 relative, and eight
            # SPADE blocks of x8-scaled weights amplify that (measured: mean 3.4e-3, max 0.26, 5.7 % of the outputs
            # off by more than 2e-2).  No two bf16 evaluations of this network agree better than that.
            gq = torch.Generator().manual_seed(11)
            want_bf2 = O.spade_generator_forward(sd, x * (1 + 1e-6 * torch.randn(x.shape, generator=gq)), seg, 1024, 768,
                                                 "most", noise=noise)
        finally:
            O.QUANT["fn"] = None
    m.cuda().eval()
    got = m(x.cuda(), seg.cuda(), noise=noise).cpu()
    rel = float((got - want).abs().max() / want.abs().max())
    opt.fp16 = True
    got_bf = m(x.cuda(), seg.cuda(), noise=noise).cpu()
    err = (got_bf - want_bf).abs()
    dev = (got_bf - want).abs()
    self_err = (want_bf2 - want_bf).abs()
    frac = lambda e: float((e > 2e-2).float().mean())     # noqa: E731
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "fullsize_generator_parity.txt"), "w") as f:
        f.write(f"SPADEGenerator fwd 1x1024x768 ngf=64 'most', |want|max {float(want.abs().max()):.4f}\n")
        f.write(f"fp32 engine vs oracle: max-rel-err {rel:.3e} (tolerance 1e-3)\n")
        f.write(f"bf16 engine vs oracle with bf16 operand rounding (QUANT): max {float(err.max()):.3e} mean "
                f"{float(err.mean()):.3e} frac>2e-2 {frac(err):.3e}\n")
        f.write(f"bf16-rounded oracle vs itself, input nudged by 1e-6: max {float(self_err.max()):.3e} mean "
                f"{float(self_err.mean()):.3e} frac>2e-2 {frac(self_err):.3e}\n")
        f.write(f"bf16 engine vs fp32 oracle: max {float(dev.max()):.3e} mean {float(dev.mean()):.3e}\n")
        f.write(f"bf16-rounded oracle vs fp32 oracle: mean {float((want_bf - want).abs().mean()):.3e}\n")
    assert rel < 1e-3, rel
    # stated bf16 tolerance (outputs are tanh-bounded, |y| <= 1): against the oracle WITH THE SAME ROUNDING POINTS the
    # engine is at most twice as far away as the rounded oracle is from its own nudged re-evaluation, in the mean and in
    # the share of outputs off by more than 2e-2, and no further from the fp32 result than bf16 rounding itself is
    assert float(err.mean()) < 2.0 * float(self_err.mean()) + 1e-4, (float(err.mean()), float(self_err.mean()))
    assert frac(err) < 2.0 * frac(self_err) + 1e-3, (frac(err), frac(self_err))
    assert float(dev.mean()) < 1.5 * float((want_bf - want).abs().mean()) + 1e-4


def test_generator_step_512x384_ngf64_vs_oracle_autograd():
    """fp32 engine: reassociation only.  Mixed precision (--fp16: bf16 matrix-core operands) against the SAME fp32 oracle
    pass -- stated bf16 tolerance: image mean-abs 3e-3 (max 3e-2 of the range), loss terms 2e-3 relative, gradient
    cosine >= 0.99 on every sizeable parameter."""
    os.makedirs(OUT, exist_ok=True)
    reps = step_check.compare_generator_step(512, 384, 64, 64, 1, seed=0, wmul=8.0, mixed=(False, True), with_vgg=True,
                                             table_path=os.path.join(OUT, "grad_parity_gen_512x384_ngf64.txt"),
                                             cpu_threads=min(os.cpu_count() or 1, 32))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "step_parity_gen_512x384_ngf64.txt"), "w") as f:
        f.write(repr(reps) + "\n")
    rep = reps[False]
    assert rep["image_max_rel_err"] < 1e-3, rep
    assert all(v < 1e-3 for v in rep["loss_rel_err"].values()), rep
    # sign() of the L1 terms (feature matching, VGG) turns round-off into flipped gradient elements deep below the loss
    assert rep["grad_worst_rel_err"] < 2e-2 and rep["grad_median_rel_err"] < 2e-3, rep
    rep = reps[True]
    # measured 1.2e-3 / 1.1e-2 / 1e-4 / 0.9943 -- and 5.4e-3 / 4.7e-2 / 2e-4 / 0.985 while the device packer fed the patch
    # tiles of up_3 / up_4 wrong gamma|beta weights (gamma, beta are small at initialisation): the bounds sit between
    assert rep["image_mean_abs_err"] < 3e-3 and rep["image_max_rel_err"] < 3e-2, rep
    assert all(v < 2e-3 for v in rep["loss_rel_err"].values()), rep
    assert rep["grad_min_cosine"] > 0.99, rep


def test_generator_step_1024x768_batch2_vs_oracle_autograd():
    """The same comparison at the BENCH resolution with two images: the tile counts of the released configuration, so
    the size-gated kernels of the mixed-precision path are the ones compared -- patch tiles on up_1..up_4 (gamma|beta
    forward / data gradient, the 128-channel resblock convolutions), LDS-DMA weight gradients, thin convolutions of the
    1024x768 level, sub-batch launches.  (~13 GB per image for the CPU autograd pass.)"""
    os.makedirs(OUT, exist_ok=True)
    reps = step_check.compare_generator_step(1024, 768, 64, 64, 2, seed=1, wmul=8.0, mixed=(False, True), with_vgg=True,
                                             table_path=os.path.join(OUT, "grad_parity_gen_1024x768_n2.txt"),
                                             cpu_threads=min(os.cpu_count() or 1, 32))
    with open(os.path.join(OUT, "step_parity_gen_1024x768_n2.txt"), "w") as f:
        f.write(repr(reps) + "\n")
    rep = reps[False]
    assert rep["image_max_rel_err"] < 1e-3, rep
    assert all(v < 1e-3 for v in rep["loss_rel_err"].values()), rep
    assert rep["grad_worst_rel_err"] < 2e-2 and rep["grad_median_rel_err"] < 2e-3, rep
    rep = reps[True]
    assert rep["image_mean_abs_err"] < 3e-3 and rep["image_max_rel_err"] < 3e-2, rep
    assert all(v < 2e-3 for v in rep["loss_rel_err"].values()), rep
    assert rep["grad_min_cosine"] > 0.99, rep


def test_discriminator_step_1024x768_batch2_vs_oracle_autograd():
    """The D half of the headline iteration (train_generator.py:327-360) at the bench resolution, two images: hinge D losses,
    every PatchGAN gradient (odd extents 513x385 / 257x193: zero-padded-dY bf16 weight gradients, stride-2 phase data
    gradients), D's Adam step -- fp32 engine and --fp16 engine against torch autograd over the oracle."""
    os.makedirs(OUT, exist_ok=True)
    reps = step_check.compare_discriminator_step(1024, 768, 64, 64, 2, seed=1, mixed=(False, True),
                                                 cpu_threads=min(os.cpu_count() or 1, 32),
                                                 table_path=os.path.join(OUT, "grad_parity_dis_1024x768_n2.txt"))
    with open(os.path.join(OUT, "step_parity_dis_1024x768_n2.txt"), "w") as f:
        f.write(repr(reps) + "\n")
    rep = reps[False]
    assert all(v < 1e-4 for v in rep["loss_rel_err"].values()), rep
    assert rep["grad_worst_rel_err"] < 1e-2 and rep["grad_min_cosine"] > 0.9999, rep
    assert rep["post_step_weight_frac_off_by_more_than_lr_tenth"] < 1e-3, rep
    rep = reps[True]
    # bf16 operands (8 mantissa bits): loss terms 5e-3 relative; the gradient (hinge / LeakyReLU masks: discontinuous) against the
    # FP32 oracle: cosine >= 0.99 -- PatchGAN model1's forward keeps fp32 operands in the D step (gen_train._d_f32; measured 0.9916;
    # every convolution on bf16 operands: 0.980, the oracle's own bf16-operand evaluation: 0.983, the bound of rounds 3-4) -- and a
    # median error no worse than that evaluation's
    ref = rep["bf16_rounded_oracle_vs_fp32_oracle"]
    assert all(v < 5e-3 for v in rep["loss_rel_err"].values()), rep
    assert rep["grad_min_cosine"] > 0.99, rep
    assert rep["grad_median_rel_err"] < ref["grad_median_rel_err"] + 1e-3, rep


@pytest.mark.parametrize("seed", [2, 3])
def test_discriminator_step_bf16_cosine_holds_on_other_draws(seed):
    """the bf16 D-gradient bound on more than the one seed it was tuned on.  With round 5's default (model1's forward
    alone on fp32 operands) seed 2 gives 0.9825; the chain model0..model2 of the half-resolution scale gives 0.9975 / 0.9960 /
    0.9973 on seeds 1 / 2 / 3 (gen_train._d_f32, profiles/r06_d_f32_seeds.txt)."""
    reps = step_check.compare_discriminator_step(1024, 768, 64, 64, 2, seed=seed, mixed=(True,), cpu_threads=min(os.cpu_count() or 1, 32))
    rep = reps[True]
    assert all(v < 5e-3 for v in rep["loss_rel_err"].values()), rep
    assert rep["grad_min_cosine"] > 0.99, rep
