"""Checker for one train_generator.py iteration (train_generator.py:279-360): the product classes on the
MI355X against torch autograd over the CPU restatement (hrviton_oracle.py), at any size the CPU can afford.

TEST INFRASTRUCTURE ONLY -- imported by tests/ and by bench.py's parity / cpu_baseline legs, never by the
product package.  Entry points:

* ``build``            one deterministic (generator, PatchGAN, VGG criterion, batch, SPADE noise) recipe;
* ``compare_generator_step``  G-step losses, the generated image and EVERY parameter gradient, HIP vs oracle
                       (fp32: reassociation only; ``mixed=True``: bf16 matrix-core operands, stated tolerance);
* ``compare_discriminator_step``  the D half (no_grad G forward, D losses, every D gradient, post-step D weights);
* ``cpu_train_generator_step``  the whole iteration (G step + D step + Adam) on the oracle, for the CPU baseline.
"""
from __future__ import annotations

import time
from argparse import Namespace
from typing import Dict, List, Optional

import torch

from . import hrviton_oracle as O


def _fl(v):
    """python float of a loss value (a tensor that may still require grad, or a number)"""
    return float(v.detach()) if hasattr(v, "detach") else float(v)


def build(H: int, W: int, ngf: int, ndf: int, N: int, seed: int = 0, wmul: float = 8.0, layers: str = "most"):
    """SPADEGenerator(ngf, ``layers``) + MultiscaleDiscriminator(ndf) with the reference's xavier(0.02) init, the
    non-spectral weights scaled by ``wmul`` (a random-init generator is otherwise ~linear), random biases and
    noise_scale; one synthetic batch (x [N,9,H,W], one-hot 7-class blobs, real image) and the SPADE noise."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.network_generator import MultiscaleDiscriminator, SPADEGenerator
    from hr_viton_amd.vgg import VGGLoss
    opt = Namespace(cuda=True, norm_G="spectralaliasinstance", gen_semantic_nc=7, ngf=ngf, num_upsampling_layers=layers,
                    fine_height=H, fine_width=W, ndf=ndf, norm_D="spectralinstance", n_layers_D=3, num_D=2,
                    no_ganFeat_loss=False, lambda_feat=10.0, lambda_vgg=10.0, no_vgg_loss=False)
    torch.manual_seed(seed)
    gen = SPADEGenerator(opt, 9)
    gen.init_weights("xavier", 0.02)
    D = MultiscaleDiscriminator(opt)
    D.init_weights("xavier", 0.02)
    vgg = VGGLoss(Namespace(cuda=False))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n_, p in list(gen.named_parameters()) + list(D.named_parameters()):
            if n_.endswith("noise_scale"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif n_.endswith("weight") or n_.endswith("weight_orig"):
                p.mul_(wmul)
            elif n_.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    x = torch.rand(N, 9, H, W, generator=g) * 2 - 1
    b = 16 if H % 16 == 0 and W % 16 == 0 else 1
    lab = torch.randint(0, 7, (N, 1, H // b, W // b), generator=g).repeat_interleave(b, 2).repeat_interleave(b, 3)
    seg = torch.zeros(N, 7, H, W).scatter_(1, lab, 1.0)
    real = torch.rand(N, 3, H, W, generator=g) * 2 - 1
    noise = {}
    for j, name in enumerate(gen._blocks()):
        h, w = gen.sh << j, gen.sw << j
        k = 3 if getattr(gen, name).learned_shortcut else 2
        noise[name] = [torch.randn(N, w, h, 1, generator=g) for _ in range(k)]
    return opt, gen, D, vgg, x, seg, real, noise


def oracle_sd(mod) -> Dict[str, torch.Tensor]:
    return {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith(("weight_u", "weight_v")))
            for k, v in mod.state_dict().items()}


def oracle_generator_losses(opt, sd_g, sd_d, sd_vgg, x, seg, real, noise):
    """Generator half of the iteration (train_generator.py:279-314) on the oracle; spectral norm in training mode
    (one power iteration per forward, like torch)."""
    O.SN_TRAIN["on"], O.SN_TRAIN["uv"] = True, {}
    try:
        fake = O.spade_generator_forward(sd_g, x, seg, opt.fine_height, opt.fine_width, opt.num_upsampling_layers,
                                         noise=noise)
        pred = O.gen_discriminator_forward(sd_d, torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0))
    finally:
        O.SN_TRAIN["on"] = False
    pf, pr = O.split_fake_real(pred)
    losses = {"GAN": O.hinge_loss(pf, True, False), "GAN_Feat": O.feat_match_loss(pf, pr, opt.lambda_feat)}
    if sd_vgg is not None:
        losses["VGG"] = O.vgg_loss(sd_vgg, fake, real) * opt.lambda_vgg
    return fake, losses


def _grad_table(mod, sd):
    """[(rel_err, abs_err, |want|max, name)] per parameter; rel = abs / max(|want|max, 1e-3 x the module's largest
    gradient magnitude): analytically-zero gradients (a bias in front of an InstanceNorm) are pure round-off."""
    gmax = max((sd[n].grad.abs().max().item() for n, _ in mod.named_parameters() if sd[n].grad is not None), default=1.0)
    rows = []
    for name, p in mod.named_parameters():
        want = sd[name].grad
        if want is None or p.grad is None:
            continue
        got = p.grad.detach().float().cpu()
        aerr = (got - want).abs().max().item()
        cos = torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0).item() if want.numel() > 1 else 1.0
        rows.append((aerr / max(want.abs().max().item(), 1e-3 * gmax), aerr, want.abs().max().item(), cos, name))
    rows.sort(reverse=True)
    return rows, gmax


def compare_generator_step(H: int, W: int, ngf: int = 64, ndf: int = 64, N: int = 1, seed: int = 0, wmul: float = 8.0,
                           mixed=False, with_vgg: bool = True, table_path: Optional[str] = None,
                           cpu_threads: int = 0):
    """``mixed``: False / True, or a tuple of engines, e.g. (False, True): ONE oracle pass, one HIP pass per engine,
    returns {engine: report} (the CPU pass dominates the cost)."""
    if isinstance(mixed, (tuple, list)):
        return _compare(H, W, ngf, ndf, N, seed, wmul, tuple(mixed), with_vgg, table_path, cpu_threads)
    return _compare(H, W, ngf, ndf, N, seed, wmul, (mixed,), with_vgg, table_path, cpu_threads)[mixed]


def _compare(H, W, ngf, ndf, N, seed, wmul, engines, with_vgg, table_path, cpu_threads) -> dict:
    """Runs the generator half of one iteration on cuda:0 (product classes) and on the CPU oracle with identical
    weights, inputs and SPADE noise.  Returns max-rel errors of the image and the loss terms, the worst / median
    per-parameter gradient error and cosine, and optionally writes the per-parameter table."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import train_ops as T
    from hr_viton_amd.losses import GANLoss, L1Loss
    if cpu_threads:
        torch.set_num_threads(cpu_threads)
    opt, gen, D, vgg, x, seg, real, noise = build(H, W, ngf, ndf, N, seed, wmul)
    sd_g, sd_d = oracle_sd(gen), oracle_sd(D)
    sd_vgg = {k: v.detach().clone() for k, v in vgg.vgg.state_dict().items()} if with_vgg else None
    t0 = time.perf_counter()
    fake, losses = oracle_generator_losses(opt, sd_g, sd_d, sd_vgg, x, seg, real, noise)
    sum(losses.values()).backward()
    t_oracle = time.perf_counter() - t0
    gen.cuda().train()
    D.cuda().train()
    vgg.cuda()
    sd0_g = {k: v.detach().clone() for k, v in gen.state_dict().items()}       # u, v are advanced by every forward
    sd0_d = {k: v.detach().clone() for k, v in D.state_dict().items()}
    reports = {}
    for mixed in engines:
        gen.load_state_dict(sd0_g)
        D.load_state_dict(sd0_d)
        for p_ in list(gen.parameters()) + list(D.parameters()):
            p_.grad = None
        reports[mixed] = _hip_pass(opt, gen, D, vgg, x, seg, real, noise, mixed, with_vgg, losses, fake, sd_g, N, H, W, ngf,
                                   t_oracle, None if table_path is None else
                                   (table_path if len(engines) == 1 else table_path.replace(".txt", "_bf16.txt" if mixed else "_f32.txt")))
    return reports


def _hip_pass(opt, gen, D, vgg, x, seg, real, noise, mixed, with_vgg, losses, fake, sd_g, N, H, W, ngf, t_oracle, table_path):
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import train_ops as T
    from hr_viton_amd.losses import GANLoss, L1Loss
    T.MMA_BF16[0] = mixed
    try:
        xc, sc, rc = x.cuda(), seg.cuda(), real.cuda()
        out = gen(xc, sc, noise={k: [z.cuda() for z in v] for k, v in noise.items()})
        pf, pr = D(torch.cat([torch.cat([sc, out], 1), torch.cat([sc, rc], 1)], 0), split=True)
        got = {"GAN": GANLoss("hinge")(pf, True, for_discriminator=False)}
        feat = 0
        for i in range(len(pf)):
            for j in range(len(pf[i]) - 1):
                feat = feat + L1Loss()(pf[i][j], pr[i][j].detach()) * opt.lambda_feat / len(pf)
        got["GAN_Feat"] = feat
        if with_vgg:
            got["VGG"] = vgg(out, rc) * opt.lambda_vgg
        sum(got.values()).mean().backward()
        torch.cuda.synchronize()
    finally:
        T.MMA_BF16[0] = False
    rows_g, gmax = _grad_table(gen, sd_g)
    rel = lambda a, b: float(((a.detach().float().cpu() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-12)))  # noqa: E731
    rep = {"size": f"{N}x{H}x{W} ngf={ngf}", "mixed": mixed, "oracle_fwd_bwd_s": round(t_oracle, 2),
           "image_max_rel_err": rel(out, fake),
           "image_mean_abs_err": float((out.detach().cpu() - fake.detach()).abs().mean()),
           "loss_rel_err": {k: abs(_fl(got[k]) - _fl(losses[k])) / max(1.0, abs(_fl(losses[k]))) for k in losses},
           "losses_oracle": {k: _fl(v) for k, v in losses.items()},
           "grad_worst_rel_err": rows_g[0][0], "grad_worst_name": rows_g[0][4],
           "grad_median_rel_err": rows_g[len(rows_g) // 2][0],
           "grad_min_cosine": min(r[3] for r in rows_g if r[2] > 1e-2 * gmax and not r[4].endswith("noise_scale")),
           "n_params_compared": len(rows_g)}
    if table_path:
        with open(table_path, "w") as f:
            f.write(f"# generator step {rep['size']} mixed={mixed}: rel_err abs_err |want|max cosine name "
                    f"(module max grad {gmax:.3e})\n")
            for r in rows_g:
                f.write("%.3e %.3e %.3e %.6f %s\n" % r)
    return rep


def compare_discriminator_step(H: int, W: int, ngf: int = 64, ndf: int = 64, N: int = 1, seed: int = 0, wmul: float = 8.0,
                               mixed=(False,), cpu_threads: int = 0, table_path: Optional[str] = None) -> dict:
    """The DISCRIMINATOR half of the iteration (train_generator.py:327-360) on cuda:0 against the oracle, from identical
    weights: no_grad generator forward (its own noise draw), PatchGAN on [fake; real] (spectral norm in training mode),
    hinge D losses, backward, Adam(lr 4e-4, betas 0 / 0.9).  Compared: the two loss terms, EVERY discriminator parameter
    gradient (the odd-extent 4x4 stride-2 layers at 513x385 / 257x193 go through the zero-padded-dY weight gradients in
    mixed precision), and the post-step weights."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import train_ops as T
    from hr_viton_amd.losses import GANLoss
    from hr_viton_amd.optim import Adam
    if cpu_threads:
        torch.set_num_threads(cpu_threads)
    engines = tuple(mixed) if isinstance(mixed, (tuple, list)) else (mixed,)
    opt, gen, D, _vgg, x, seg, real, noise = build(H, W, ngf, ndf, N, seed, wmul)
    sd_g, sd_d = oracle_sd(gen), oracle_sd(D)
    t0 = time.perf_counter()
    O.SN_TRAIN["on"], O.SN_TRAIN["uv"] = True, {}
    try:
        with torch.no_grad():
            fake = O.spade_generator_forward(sd_g, x, seg, H, W, opt.num_upsampling_layers, noise=noise)
        pred = O.gen_discriminator_forward(sd_d, torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0))
    finally:
        O.SN_TRAIN["on"] = False
    pf, pr = O.split_fake_real(pred)
    want_l = {"D_Fake": O.hinge_loss(pf, False, True), "D_Real": O.hinge_loss(pr, True, True)}
    pd = [(k, v) for k, v in sd_d.items() if v.requires_grad]
    od = torch.optim.Adam([v for _, v in pd], lr=4e-4, betas=(0.0, 0.9))
    od.zero_grad()
    sum(want_l.values()).backward()
    want_g = {k: v.grad.detach().clone() for k, v in pd if v.grad is not None}
    wantq_g = None
    if any(engines):
        # the same half on the oracle WITH THE bf16 ENGINE'S ROUNDING POINTS (conv operands of G and D rounded, straight-
        # through backward): how far a bf16-operand evaluation of this (hinge / LeakyReLU / InstanceNorm) gradient sits
        # from the fp32 one, engine or not -- the yardstick of the mixed-precision comparison below
        od.zero_grad()
        O.QUANT["fn"] = lambda t: t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()
        O.SN_TRAIN["on"], O.SN_TRAIN["uv"] = True, {}
        try:
            with torch.no_grad():
                fq = O.spade_generator_forward(sd_g, x, seg, H, W, opt.num_upsampling_layers, noise=noise)
            predq = O.gen_discriminator_forward(sd_d, torch.cat([torch.cat([seg, fq], 1), torch.cat([seg, real], 1)], 0))
        finally:
            O.SN_TRAIN["on"] = False
            O.QUANT["fn"] = None
        pfq, prq = O.split_fake_real(predq)
        (O.hinge_loss(pfq, False, True) + O.hinge_loss(prq, True, True)).backward()
        wantq_g = {k: v.grad.detach().clone() for k, v in pd if v.grad is not None}
        for k, v in pd:                      # restore the fp32 gradients for the reference Adam step
            v.grad = want_g[k].clone() if k in want_g else None
    od.step()
    want_w = {k: v.detach().clone() for k, v in pd}
    t_oracle = time.perf_counter() - t0
    gen.cuda().train()
    D.cuda().train()
    sd0_g = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    sd0_d = {k: v.detach().clone() for k, v in D.state_dict().items()}
    reports = {}
    import os
    for mx in engines:
        # an engine entry is True / False, or (True, {environment switches}, label): the bf16 engine under those switches, reported
        # under ``label`` (tools/d_f32_layers.py: which PatchGAN layers keep fp32 operands) -- the oracle above is computed once
        env, label = {}, None
        if isinstance(mx, tuple):
            mx, env, label = mx
        gen.load_state_dict(sd0_g)
        D.load_state_dict(sd0_d)
        for p_ in D.parameters():
            p_.grad = None
        T.MMA_BF16[0] = bool(mx)
        old_env = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            xc, sc, rc = x.cuda(), seg.cuda(), real.cuda()
            with torch.no_grad():
                out = gen(xc, sc, noise={k: [z.cuda() for z in v] for k, v in noise.items()})
            pfh, prh = D(torch.cat([torch.cat([sc, out], 1), torch.cat([sc, rc], 1)], 0), split=True)
            crit = GANLoss("hinge")
            got_l = {"D_Fake": crit(pfh, False, for_discriminator=True), "D_Real": crit(prh, True, for_discriminator=True)}
            opt_d = Adam(D.parameters(), lr=4e-4, betas=(0.0, 0.9))
            opt_d.zero_grad()
            sum(got_l.values()).mean().backward()
            got_g = {n: p.grad.detach().float().cpu().clone() for n, p in D.named_parameters() if p.grad is not None}
            opt_d.step()
            torch.cuda.synchronize()
        finally:
            T.MMA_BF16[0] = False
            for k, v in old_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        gmax = max(float(w.abs().max()) for w in want_g.values())

        def table(got):
            rws = []
            for n, w in want_g.items():
                a = got[n]
                cos = float(torch.nn.functional.cosine_similarity(a.flatten(), w.flatten(), dim=0)) if w.numel() > 1 else 1.0
                rws.append((float((a - w).abs().max()) / max(float(w.abs().max()), 1e-3 * gmax), float((a - w).abs().max()),
                            float(w.abs().max()), cos, n))
            rws.sort(reverse=True)
            return rws
        rows = table(got_g)
        # post-step weights: Adam's first step is -lr * g / (|g| + eps): compared where the reference gradient is not ~0
        worst_w, bad_frac = 0.0, 0.0
        params = dict(D.named_parameters())
        for n, w in want_w.items():
            if n not in want_g:
                continue
            g = want_g[n]
            big = g.abs() > 1e-2 * g.abs().max()
            dw = (params[n].detach().float().cpu() - w)[big].abs()
            if dw.numel():
                worst_w = max(worst_w, float(dw.max()))
                bad_frac = max(bad_frac, float((dw > 4e-5).float().mean()))
        rep = {"size": f"{N}x{H}x{W} ndf={ndf}", "mixed": bool(mx), "oracle_s": round(t_oracle, 2),
               "loss_rel_err": {k: abs(float(got_l[k]) - float(want_l[k])) / max(1.0, abs(float(want_l[k]))) for k in want_l},
               "losses_oracle": {k: float(v) for k, v in want_l.items()},
               "grad_worst_rel_err": rows[0][0], "grad_worst_name": rows[0][4], "grad_median_rel_err": rows[len(rows) // 2][0],
               "grad_min_cosine": min(r_[3] for r_ in rows if r_[2] > 1e-2 * gmax),
               "post_step_weight_max_abs_diff": worst_w, "post_step_weight_frac_off_by_more_than_lr_tenth": bad_frac,
               "n_params_compared": len(rows)}
        if mx and wantq_g is not None:
            rq = table(wantq_g)
            rep["bf16_rounded_oracle_vs_fp32_oracle"] = {"grad_worst_rel_err": rq[0][0], "grad_median_rel_err": rq[len(rq) // 2][0],
                                                         "grad_min_cosine": min(r_[3] for r_ in rq if r_[2] > 1e-2 * gmax)}
        reports[bool(mx) if label is None else label] = rep
        if table_path:
            with open(table_path.replace(".txt", "_bf16.txt" if mx else "_f32.txt"), "w") as f:
                f.write(f"# discriminator step {rep['size']} mixed={bool(mx)}: rel_err abs_err |want|max cosine name\n")
                for r_ in rows:
                    f.write("%.3e %.3e %.3e %.6f %s\n" % r_)
    return reports


def _cond_step(mixed, opt, tocg, D, batch):
    """one condition_train_step on the HIP path; returns (losses, tocg grads, D grads) captured before the optimizer steps"""
    from hr_viton_amd import networks, pipeline, train_ops as T
    from hr_viton_amd.losses import L1Loss
    from hr_viton_amd.optim import Adam
    opt.fp16 = mixed
    T.MMA_BF16[0] = bool(mixed)
    try:
        og = Adam(tocg.parameters(), lr=0.0002, betas=(0.5, 0.999))
        od = Adam(D.parameters(), lr=0.0002, betas=(0.5, 0.999))
        gg, gd = {}, {}
        sg, sd_ = og.step, od.step

        def step_g():
            gg.update({n: p.grad.detach().float().cpu().clone() for n, p in tocg.named_parameters() if p.grad is not None})
            return sg()

        def step_d():
            gd.update({n: p.grad.detach().float().cpu().clone() for n, p in D.named_parameters() if p.grad is not None})
            return sd_()
        og.step, od.step = step_g, step_d
        losses = pipeline.condition_train_step(opt, tocg, D, L1Loss(), None, networks.GANLoss(use_lsgan=True), og, od,
                                               {k: v.cuda() for k, v in batch.items()})
        torch.cuda.synchronize()
        return {k: float(v.detach()) for k, v in losses.items() if torch.is_tensor(v) and v.numel() == 1}, gg, gd
    finally:
        T.MMA_BF16[0] = False


def compare_condition_step(H: int = 512, W: int = 384, ngf: int = 96, N: int = 1, engines=(False, True), cpu_threads: int = 0,
                           out_dir: Optional[str] = None) -> dict:
    """One train_condition.py iteration (train_condition.py:136-286, --Ddownx2 --lasttvonly --interflowloss) on cuda:0,
    fp32 engine and --fp16 engine, against torch autograd over the fp32 oracle -- and, as the yardstick of the bf16
    comparison, the oracle evaluated WITH THE ENGINE'S ROUNDING POINTS (bf16 conv operands, straight-through backward)."""
    import os
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import networks
    from .recipes import condstep_build
    if cpu_threads:
        torch.set_num_threads(cpu_threads)
    opt, tocg, D, batch = condstep_build(networks.ConditionGenerator, networks.define_D, ngf=ngf, N=N, H=H, W=W)
    opt.lasttvonly, opt.interflowloss, opt.occlusion, opt.clothmask_composition = True, True, False, "warp_grad"
    opt.edgeawaretv, opt.add_lasttv = "no_edge", False
    opt.tvlambda, opt.CElamda, opt.GANlambda, opt.no_GAN_loss = 2.0, 10.0, 1.0, False
    sd_g = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in tocg.state_dict().items()}
    sd_d = {k: v.detach().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    t0 = time.perf_counter()
    r = O.condition_train_losses(sd_g, sd_d, None, batch, occlusion=False, composition="warp_grad", edgeawaretv="no_edge",
                                 add_lasttv=False)
    r["loss_G"].backward(retain_graph=True)
    want_g = {k: v.grad.clone() for k, v in sd_g.items() if v.grad is not None}
    for v in sd_d.values():
        v.grad = None
    r["loss_D"].backward()
    want_d = {k: v.grad.clone() for k, v in sd_d.items() if v.grad is not None}
    t_oracle = time.perf_counter() - t0
    r = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in r.items()}      # (drop the autograd graph: GBs at 1024x768)
    # the same iteration on the oracle WITH THE ENGINE'S ROUNDING POINTS (bf16 conv operands in the forward, straight-through
    # in the backward): what a bf16-operand evaluation of this loss does to its (discontinuous) gradient, engine or not
    wantq_g = None
    if any(engines):
        for v in list(sd_g.values()) + list(sd_d.values()):
            v.grad = None
        O.QUANT["fn"] = lambda t: t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()
        try:
            rq = O.condition_train_losses(sd_g, sd_d, None, batch, occlusion=False, composition="warp_grad", edgeawaretv="no_edge",
                                          add_lasttv=False)
            rq["loss_G"].backward()
        finally:
            O.QUANT["fn"] = None
        wantq_g = {k: v.grad.clone() for k, v in sd_g.items() if v.grad is not None}
        del rq
    sd0_g = {k: v.detach().clone() for k, v in tocg.state_dict().items()}
    sd0_d = {k: v.detach().clone() for k, v in D.state_dict().items()}
    tocg.cuda().train()
    D.cuda().train()
    rep = {}
    for mixed in engines:
        tocg.load_state_dict(sd0_g)
        D.load_state_dict(sd0_d)
        for p_ in list(tocg.parameters()) + list(D.parameters()):
            p_.grad = None
        losses, gg, gd = _cond_step(mixed, opt, tocg, D, batch)
        lerr = {k: abs(losses[k] - float(r[k].detach())) / max(1.0, abs(float(r[k].detach())))
                for k in ("l1", "tv", "ce", "g_gan", "loss_G", "d_fake", "d_real", "loss_D")}

        def table(got, want):
            gmax = max(float(w.abs().max()) for w in want.values())
            rows = []
            for n, w in want.items():
                a = got[n]
                cos = float(torch.nn.functional.cosine_similarity(a.flatten(), w.flatten(), dim=0)) if w.numel() > 1 else 1.0
                rows.append((float((a - w).abs().max()) / max(float(w.abs().max()), 1e-3 * gmax), cos, float(w.abs().max()), n))
            rows.sort(reverse=True)
            sizeable = [x for x in rows if x[2] > 1e-2 * gmax]
            return rows, dict(worst_rel=rows[0][0], worst=rows[0][3], median_rel=rows[len(rows) // 2][0],
                              min_cosine=min(x[1] for x in sizeable), n=len(rows))
        rows_g, sum_g = table(gg, want_g)
        rows_d, sum_d = table(gd, want_d)
        rep[mixed] = dict(loss_rel_err=lerr, tocg=sum_g, D=sum_d)
        if mixed:
            rep["fp16_engine_vs_bf16_rounded_oracle"] = table(gg, wantq_g)[1]
            rep["bf16_rounded_oracle_vs_fp32_oracle"] = table(wantq_g, want_g)[1]
        if out_dir:
            with open(os.path.join(out_dir, "grad_parity_cond_%dx%d_ngf%d_%s.txt" % (H, W, ngf, "fp16" if mixed else "f32")), "w") as f:
                f.write(f"# train_condition iteration {N}x{H}x{W} ngf={ngf} engine={'bf16 MFMA' if mixed else 'fp32'}: {rep[mixed]}\n")
                for x in rows_g:
                    f.write("tocg %.3e %.6f %.3e %s\n" % x)
                for x in rows_d:
                    f.write("D    %.3e %.6f %.3e %s\n" % x)
    rep["size"] = f"{N}x{H}x{W} ngf={ngf}"
    rep["oracle_fwd_bwd_s"] = round(t_oracle, 2)
    return rep


def compare_tryon_step(opt, tocg, gen, inputs: Dict[str, torch.Tensor], mixed: bool, cpu_threads: int = 0) -> dict:
    """One image of the end-to-end test_generator.py step (test_generator.py:118-219: tocg at 256x192 -> parse glue -> high-
    resolution warp -> occlusion -> SPADE generator) on the HIP path against the oracle's composition of the same step.
    ``mixed``: the bf16 engines are compared with the oracle evaluated WITH THE SAME ROUNDING POINTS (oracle.QUANT), and the
    bound is that oracle's own re-evaluation with the inputs nudged by 1e-6 (two bf16 evaluations of this pipeline differ
    by flipped label pixels and by rounding-boundary flips amplified through eight SPADE blocks)."""
    import torch.nn.functional as F
    from hr_viton_amd.pipeline import tryon_step
    if cpu_threads:
        torch.set_num_threads(cpu_threads)
    H, W = opt.fine_height, opt.fine_width
    inp = {k: v[:1].detach().float().cpu() for k, v in inputs.items()}
    sd_t = {k: v.detach().cpu().clone() for k, v in tocg.state_dict().items()}
    sd_g = {k: v.detach().cpu().clone() for k, v in gen.state_dict().items()}
    layers = opt.num_upsampling_layers
    comp = getattr(opt, "clothmask_composition", "warp_grad")

    def oracle(nudge: float):
        g = torch.Generator().manual_seed(17)
        j = (lambda t: t * (1 + nudge * torch.randn(t.shape, generator=g))) if nudge else (lambda t: t)
        lo = (256, 192)
        cm = (inp["cloth_mask"] > 0.5).float()
        cloth, pose, agn = j(inp["cloth"]), j(inp["densepose"]), j(inp["agnostic"])
        i1 = torch.cat([O.resize_bilinear(cloth, size=lo), O.resize_nearest(cm, lo)], 1)
        i2 = torch.cat([O.resize_nearest(inp["parse_agnostic"], lo), O.resize_bilinear(pose, size=lo)], 1)
        flow_list, seg, _, wcm_p = O.tocg_forward(sd_t, i1, i2)
        gauss, lab_w, parse_w = O.parse_glue(seg, wcm_p, H, W, comp)
        wc, wm = O.hires_warp(flow_list[-1], cloth, cm)
        if getattr(opt, "occlusion", False):
            wm = O.remove_overlap(F.softmax(gauss, dim=1), wm)
            wc = wc * wm + torch.ones_like(wc) * (1 - wm)
        out = O.spade_generator_forward(sd_g, torch.cat((agn, pose, wc), 1), parse_w, H, W, layers)
        return out, lab_w, wc

    t0 = time.perf_counter()
    with torch.no_grad():
        O.QUANT["fn"] = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mixed else None
        try:
            want, lab_w, wc_w = oracle(0.0)
            want2, lab_w2, _ = oracle(1e-6) if mixed else (want, lab_w, None)
        finally:
            O.QUANT["fn"] = None
    t_oracle = time.perf_counter() - t0
    dev = next(gen.parameters()).device
    with torch.no_grad():
        res = tryon_step(opt, tocg, gen, {k: v.to(dev) for k, v in inp.items()})
    got, lab_g = res["output"].float().cpu(), res["fake_parse"].cpu()[:, 0]
    err, self_err = (got - want).abs(), (want2 - want).abs()
    return {"size": f"1x{H}x{W}", "mixed": bool(mixed), "oracle_s": round(t_oracle, 2),
            "oracle": "oracle composition of test_generator.py:118-219" + (" with bf16 operand rounding at the engine's rounding points "
                                                                            "(oracle.QUANT)" if mixed else ""),
            "label_map_mismatch_frac": float((lab_g != lab_w).float().mean()),
            "label_map_mismatch_frac_oracle_vs_nudged_oracle": float((lab_w2 != lab_w).float().mean()),
            "warped_cloth_max_abs_err": float((res["warped_cloth"].float().cpu() - wc_w).abs().max()),
            "image_mean_abs_err": float(err.mean()), "image_max_abs_err": float(err.max()),
            "image_frac_off_by_2e-2": float((err > 2e-2).float().mean()),
            "oracle_vs_nudged_oracle_mean_abs": float(self_err.mean()), "oracle_vs_nudged_oracle_max_abs": float(self_err.max()),
            "oracle_vs_nudged_oracle_frac_off_by_2e-2": float((self_err > 2e-2).float().mean())}


def cpu_train_generator_step(H: int = 256, W: int = 192, ngf: int = 64, ndf: int = 64, N: int = 1, layers: str = "more",
                             repeats: int = 3, warmup: int = 1, threads: int = 0) -> dict:
    """The whole train_generator.py iteration on the oracle (CPU): G forward, PatchGAN on [fake; real], hinge +
    feature-matching + VGG losses, backward, Adam; then the D half (no_grad G forward, PatchGAN, hinge, backward,
    Adam).  1 warm-up + ``repeats`` timed iterations, median -- BASELINE.md section 4."""
    if threads:
        torch.set_num_threads(threads)
    opt, gen, D, vgg, x, seg, real, noise = build(H, W, ngf, ndf, N, 0, 8.0, layers)
    sd_g, sd_d = oracle_sd(gen), oracle_sd(D)
    sd_vgg = {k: v.detach().clone() for k, v in vgg.vgg.state_dict().items()}
    pg = [v for v in sd_g.values() if v.requires_grad]
    pd = [v for v in sd_d.values() if v.requires_grad]
    og = torch.optim.Adam(pg, lr=1e-4, betas=(0.0, 0.9))
    od = torch.optim.Adam(pd, lr=4e-4, betas=(0.0, 0.9))
    times: List[float] = []
    for it in range(warmup + repeats):
        t0 = time.perf_counter()
        _, losses = oracle_generator_losses(opt, sd_g, sd_d, sd_vgg, x, seg, real, noise)
        og.zero_grad()
        od.zero_grad()
        sum(losses.values()).backward()
        og.step()
        O.SN_TRAIN["on"], O.SN_TRAIN["uv"] = True, {}
        try:
            with torch.no_grad():
                fake = O.spade_generator_forward(sd_g, x, seg, H, W, layers, noise=noise)
            pred = O.gen_discriminator_forward(sd_d, torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0))
        finally:
            O.SN_TRAIN["on"] = False
        pf, pr = O.split_fake_real(pred)
        ld = O.hinge_loss(pf, False, True) + O.hinge_loss(pr, True, True)
        od.zero_grad()
        ld.backward()
        od.step()
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"seconds_per_step_median": med, "images_per_s": N / med, "times": times, "size": (N, H, W), "layers": layers}
