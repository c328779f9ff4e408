#!/usr/bin/env python3
"""bench.py -- the hot path's throughput on MI355X (driver contract).

Default workload = the HEADLINE, BASELINE.json configs[3] (SURVEY.md 8d "config #4"): one iteration of
train_generator.py:279-360 at 1024x768, 4 images per GPU, mixed precision (bf16 matrix cores, fp32 accumulate):
frozen tocg@256x192 + parse glue, SPADE generator forward/backward, multi-scale PatchGAN on [fake; real] twice,
VGG19 + feature-matching + hinge losses, fused Adam on both networks; DP all-reduce of the gradients for N > 1.
A "step" is that whole iteration over one synthetic batch already resident in HBM.  N>1: one process per GPU
(torch.distributed.run, backend nccl = RCCL), weak scaling; value = images of all ranks / max-over-ranks time.

Output: the LAST stdout line is ONE compact JSON object under 4 KB (compact_line): the contract keys, `roofline`, `cpu_baseline`, the
worst `parity` numbers, value / ms / frac of the `extra` configurations.  The full result -- every table, note and sub-object below --
is written to gpurun_out/bench_detail.json (the line's `detail` key names it); progress goes to stderr.
  roofline     -- the DOMINANT kernel of the step: the device-kernel family with the largest summed HIP-event time in one iteration
                  (template instances of one kernel summed -- what heads a rocprofv3 kernel trace of the same command), its
                  algorithmic FLOPs / that time against the 2.5 PFLOP/s dense bf16 MFMA peak (an HBM-bound family: bytes against
                  8 TB/s), `traffic` = HBM bytes per launch of exactly that kernel name from the committed rocprofv3 PMC passes of
                  this command (profiles/, tools/profile_traffic.sh) next to `algorithmic_bytes_per_launch`; `kernels` = the next
                  families by time; `north_star_set_frac` = the north star's own aggregate: every 3x3 convolution launch
                  of the SPADE generator (forward, data gradient, weight gradient).  Detail file only: `whole_step_conv_family`,
                  `hbm_kinds` (the HBM-bound launch kinds in GB/s against 6.3 TB/s), `slowest_launches`.
  cpu_baseline -- the oracle's whole iteration (oracle/step_check.py) on this box's host cores: ONE image 1024x768 'most', one timed
                  iteration (and, detail file, 256x192 'more': 1 warm-up + 3 timed, median -- BASELINE.md section 4).
  parity       -- the generator and discriminator halves of the iteration at 2 x 1024x768 ngf=64 against torch autograd over the
                  oracle: image, loss terms, every parameter gradient (fp32 engine), and the same with the bf16 engine.
  extra        -- BASELINE configs[4] (tryon_infer bf16, 16 img/GPU, hipGraph replay), configs[1] (tocg inference fp32, with the
                  argmax index check), configs[2] (train_condition fp32, b=8) measured in the same run.
Other workloads: --workload tocg_infer | tryon_infer | train_condition (same JSON shape).
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 (the 5 PF headline figure includes 2:1 sparsity)
HBM_ACHIEVABLE_GBPS = 6300.0   # MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy (8.0 TB/s spec)
H, W = 1024, 768

# launches of the SPADE generator that are 3x3 convolutions (network_generator.py:98-99,117-121,141-143,184-186,201):
# conv_shared / gamma|beta / conv_0 / conv_1 of every block, the stems conv_0..7 and conv_img; conv_s is 1x1
_GEN = re.compile(r"^(head_0|G_middle_\d|up_\d|conv_\d+|conv_img)(\.|\[|$)")


_CONV_S = re.compile(r"\.conv_s(\.|\[| |$)")      # the learned shortcut (1x1); NOT "conv_shared..."


def is_spade_gen_3x3(kind, name):
    return kind in ("conv", "wgrad") and _GEN.match(name) is not None and _CONV_S.search(name) is None


_T0 = time.perf_counter()


def _log(msg):
    """progress on stderr (the JSON line is the only thing on stdout)"""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def summarize(recs, peak_tflops):
    """recs: [(kind, name, flops, bytes, ms, kernel)] of ONE step -> roofline pieces."""
    kinds, kern = {}, {}
    for k, n, fl, by, ms, kn in recs:
        for tab, key in ((kinds, k), (kern, kn)):
            a = tab.setdefault(key, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += fl
            a[3] += by
    mm = [r for r in recs if r[0] in ("conv", "wgrad")]
    sp = [r for r in mm if is_spade_gen_3x3(r[0], r[1])]
    gb = [r for r in mm if r[1].endswith("[spade_gb]")]      # the SPADE gamma|beta family: fused forward, pair data gradient
    gf = [r for r in gb if "conv_shared+gamma|beta" in r[1]]  # ... of which hrv::spade_fused_kernel (the dominant kernel)

    def agg(rows):
        ms = sum(r[4] for r in rows)
        fl = sum(r[2] for r in rows)
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"launches": len(rows), "ms_per_step": round(ms, 3), "flops_per_step": fl, "achieved": round(ach, 2),
                "frac": round(ach / peak_tflops, 4)}
    hbm = {}
    for k, (n, ms, fl, by) in sorted(kinds.items()):
        if k in ("conv", "wgrad") or by <= 0 or ms <= 0:
            continue
        gbps = by / (ms * 1e-3) / 1e9
        hbm[k] = {"launches": n, "ms": round(ms, 3), "GBps": round(gbps, 1), "frac_of_6.3TBps": round(gbps / HBM_ACHIEVABLE_GBPS, 3)}
    conv_bytes = sum(r[3] for r in mm)
    gba = agg(gb)
    gba["algorithmic_bytes_per_launch"] = sum(r[3] for r in gb) / max(1, len(gb))
    gfa = agg(gf)
    gfa["algorithmic_bytes_per_launch"] = sum(r[3] for r in gf) / max(1, len(gf))
    # per device-kernel family, largest share of the step first: [(kernel, launches, ms, flops, bytes)]
    by_kernel = sorted(((kn, a[0], a[1], a[2], a[3]) for kn, a in kern.items()), key=lambda r: -r[2])
    return {"kinds": kinds, "by_kernel": by_kernel, "spade": agg(sp), "all": agg(mm), "hbm": hbm, "gb": gba, "gf": gfa,
            "conv_alg_bytes_per_launch": conv_bytes / max(1, len(mm)), "conv_launches": len(mm),
            "top": sorted(mm, key=lambda r: -r[4])[:6]}


def dump_launches(path, recs):
    with open(path, "w") as f:
        for k, n, fl, by, ms, kn in recs:
            f.write(f"{k:8s} {n:52s} {ms:9.4f} ms  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.2f} TFLOP/s  "
                    f"{by / (ms * 1e-3) / 1e9 if ms > 0 else 0:9.1f} GB/s  {kn}\n")


# launch tag of hr_viton_amd.ops._Timed -> the kernel-family key of tools/traffic_json.py
_TRAFFIC_FAMILY = {"conv_wgrad_tr_kernel": "conv_wgrad_tr", "conv_wgrad_kernel": "conv_wgrad", "thin_conv_kernel": "thin_conv",
                   "stats": "instnorm"}


def load_traffic(tag, family=None):
    """HBM bytes per launch from the committed PMC passes of this command (cannot be read in-process): of one kernel
    family (``family``, e.g. "spade_gb_kernel") or averaged over every convolution launch."""
    family = _TRAFFIC_FAMILY.get(family, family)
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        tp = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{tag}.json")
        if os.path.exists(tp):
            with open(tp) as f:
                tj = json.load(f)
            src = f"profiles/{os.path.basename(tp)} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
            if family is not None:
                fam = (tj.get("per_kernel_family") or {}).get(family)
                if not fam or not fam.get("hbm_bytes_per_step"):
                    return None, None
                # bytes of the family per STEP: the caller divides by its own launch count (a layer may take two dispatches)
                return fam["hbm_bytes_per_step"], src + f", dispatches of hrv::{family}, per step"
            return tj.get("hbm_bytes_per_launch"), src
    return None, None


# ------------------------------------------------------------------------------------------------- workloads
def build_tocg(torch, nn, mixed=False, ngf=96):
    from argparse import Namespace
    from hr_viton_amd.networks import ConditionGenerator
    opt = Namespace(cuda=True, warp_feature="T1", out_layer="relu", fp16=mixed)
    torch.manual_seed(0)
    m = ConditionGenerator(opt, 4, 16, 13, ngf=ngf, norm_layer=nn.BatchNorm2d)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.2)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                mod.weight.copy_(1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
        for fc in m.flow_conv:
            fc.weight.mul_(4.0)
    m.eval()
    return opt, m


def tocg_inputs(torch, n, seed, device):
    g = torch.Generator().manual_seed(seed)
    input1 = torch.cat([torch.rand(n, 3, H, W, generator=g) * 2 - 1,
                        (torch.rand(n, 1, H, W, generator=g) > 0.5).float()], 1)
    lab = torch.randint(0, 13, (n, 1, H, W), generator=g)
    input2 = torch.cat([torch.zeros(n, 13, H, W).scatter_(1, lab, 1.0),
                        torch.rand(n, 3, H, W, generator=g) * 2 - 1], 1)
    return input1.to(device), input2.to(device)


def wl_tocg_infer(ctx, mixed, B):
    """BASELINE configs[1]: ConditionGenerator inference 1024x768 b=4 fp32 (flow + seg + grid_sample)."""
    torch, nn = ctx["torch"], ctx["nn"]
    B = B or 4
    opt, model = build_tocg(torch, nn, mixed)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(ctx["dev"])
    i1, i2 = tocg_inputs(torch, B, ctx["hdist"].shard_seed(1234, ctx["rank"]), ctx["dev"])

    def step(_i):
        model(opt, i1, i2)

    def parity():
        """configs[1] outputs vs the oracle at 1024x768 (1 image): max-rel errors and the argmax label map -- every
        mismatching pixel's top-2 logit margin is reported in ulps of the logit."""
        from oracle import hrviton_oracle as O
        c1, c2 = i1[:1].cpu(), i2[:1].cpu()
        torch.set_num_threads(ctx["cpu_threads"])
        t0 = time.perf_counter()
        with torch.no_grad():
            want = O.tocg_forward(sd_cpu, c1, c2)
        cdt = time.perf_counter() - t0
        got = model(opt, i1[:1], i2[:1])

        def rel(a, b):
            return float(((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-12)).item())
        seg_g, seg_w = got[1].cpu(), want[1]
        lab_g, lab_w = seg_g.argmax(1), seg_w.argmax(1)
        bad = (lab_g != lab_w)
        top2 = seg_w.topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])[bad]
        ulp = torch.abs(top2[:, 0][bad]).clamp_min(1e-30)
        ulps = (margin / (ulp * 2.0 ** -23)).tolist()
        return {"oracle_forward_s": round(cdt, 2), "flow_last_max_rel_err": rel(got[0][-1], want[0][-1]),
                "seg_max_rel_err": rel(seg_g, seg_w), "warped_cloth_max_rel_err": rel(got[2], want[2]),
                "argmax_mismatch_pixels": int(bad.sum().item()), "pixels": int(lab_w.numel()),
                "mismatch_top2_margin_ulps_of_logit": [round(u, 1) for u in ulps[:16]],
                "mismatch_top2_margin_max_abs": float(margin.max().item()) if margin.numel() else 0.0}
    return dict(step=step, B=B, train=False, parity=parity, flops_per_img=1.468e12,
                metric="1024x768 try-on images/sec (ConditionGenerator inference: flow+seg+grid_sample)",
                workload="BASELINE configs[1]: ConditionGenerator inference 1024x768 fp32, ngf=96, random-init weights",
                traffic_tag="tocg_infer")


def wl_train_condition(ctx, mixed, B):
    """BASELINE configs[2]: train_condition.py 1024x768 b=8 fp32 --Ddownx2 --lasttvonly --interflowloss."""
    torch, nn, dev, world, rank = ctx["torch"], ctx["nn"], ctx["dev"], ctx["world"], ctx["rank"]
    import train_condition as tc
    from hr_viton_amd.gen_train import attach_grad_sync
    from hr_viton_amd.losses import L1Loss
    from hr_viton_amd.networks import ConditionGenerator, GANLoss, VGGLoss, define_D
    from hr_viton_amd.optim import Adam
    from hr_viton_amd.parallel import broadcast_module
    from hr_viton_amd.pipeline import condition_train_step
    B = B or 8
    opt = tc.get_opt(["--name", "bench", "--synthetic", "-b", str(B * world), "--fine_height", "1024", "--fine_width",
                      "768", "--Ddownx2", "--lasttvonly", "--interflowloss"] + (["--fp16"] if mixed else []))
    from hr_viton_amd import train_ops as _T
    _T.MMA_BF16[0] = bool(mixed)
    torch.manual_seed(0)
    tocg = ConditionGenerator(opt, 4, 16, 13, ngf=96, norm_layer=nn.BatchNorm2d).to(dev).train()
    D = define_D(input_nc=4 + 16 + 13, Ddownx2=True, Ddropout=False, n_layers_D=3, spectral=False, num_D=2).to(dev).train()
    crit_vgg = VGGLoss(opt).to(dev)
    for m in (tocg, D, crit_vgg):
        broadcast_module(m)
    og = Adam(tocg.parameters(), lr=opt.G_lr, betas=(0.5, 0.999))
    od = Adam(D.parameters(), lr=opt.D_lr, betas=(0.5, 0.999))
    sg = og.make_grad_sync() if world > 1 else None
    sd = od.make_grad_sync() if world > 1 else None
    for s_ in (sg, sd):
        if s_ is not None:
            attach_grad_sync(s_)
    l1, gan = L1Loss(), GANLoss(use_lsgan=True)
    batch = tc.synthetic_batch(opt, B, ctx["hdist"].shard_seed(4321, rank), dev)

    def step(_i):
        _T.MMA_BF16[0] = bool(mixed)     # (another workload of this process may have switched it)
        condition_train_step(opt, tocg, D, l1, crit_vgg, gan, og, od, batch, sg, sd)

    def parity():
        """fp32 (BASELINE configs[2]): ONE image at the timed resolution, 1x1024x768 ngf=96, against torch autograd over the
        oracle (tests/test_gpu_fullsize_tocg.py holds the same comparison with its bounds, plus the 8-image
        self-consistency run of the bench's own batch).  The bf16-operand mode: 1x512x384, judged against the oracle's own
        bf16-operand evaluation -- NOT a configs[2] result (see `note`)."""
        from oracle import step_check
        try:
            if mixed:
                return step_check.compare_condition_step(512, 384, 96, 1, engines=(False, True), cpu_threads=ctx["cpu_threads"])
            return step_check.compare_condition_step(1024, 768, 96, 1, engines=(False,), cpu_threads=ctx["cpu_threads"])
        finally:
            _T.MMA_BF16[0] = bool(mixed)
    return dict(step=step, B=B, train=True, parity=parity, flops_per_img=13.2e12,
                metric="1024x768 images/sec (train_condition.py step: tocg fwd/bwd with batch-stat BN, 5 VGG pairs, LSGAN D, Adam)",
                workload="BASELINE configs[2]: train_condition.py 1024x768 --Ddownx2 --lasttvonly --interflowloss, ngf=96, "
                         "random-init weights", traffic_tag="train_condition")


def wl_generator(ctx, mixed, B, train):
    """train=True: BASELINE configs[3] (headline); False: configs[4] (end-to-end test_generator.py step)."""
    torch, nn, dev, world, rank = ctx["torch"], ctx["nn"], ctx["dev"], ctx["world"], ctx["rank"]
    import train_generator as tg
    from hr_viton_amd.gen_train import attach_grad_sync
    from hr_viton_amd.losses import GANLoss, L1Loss
    from hr_viton_amd.network_generator import MultiscaleDiscriminator, SPADEGenerator
    from hr_viton_amd.networks import ConditionGenerator
    from hr_viton_amd.optim import Adam
    from hr_viton_amd.parallel import broadcast_module
    from hr_viton_amd.pipeline import generator_train_step, make_generator_inputs, tryon_step
    from hr_viton_amd.vgg import VGGLoss
    from hr_viton_amd import train_ops as _T
    B = B or (4 if train else 16)
    opt = tg.get_opt(["--name", "bench", "--synthetic", "-b", str(B * world)] + (["--fp16"] if mixed else []))
    torch.manual_seed(0)
    tocg = ConditionGenerator(opt, 4, 16, 13, ngf=96, norm_layer=nn.BatchNorm2d).to(dev).eval()
    gen = SPADEGenerator(opt, 9)
    gen.init_weights("xavier", 0.02)
    gen.to(dev)
    batch = tg.synthetic_batch(opt, B, ctx["hdist"].shard_seed(1234, rank), dev)
    if train:
        dis = MultiscaleDiscriminator(opt)
        dis.init_weights("xavier", 0.02)
        dis.to(dev).train()
        gen.train()
        crit_vgg = VGGLoss(opt).to(dev)
        for m in (gen, dis, crit_vgg):
            broadcast_module(m)
        graph = bool(ctx["args"].graph)      # --graph: the whole iteration replayed as one hipGraph (N > 1: three graph segments,
        #                                       the two gradient all-reduces between them -- hr_viton_amd.graph.GraphedIteration)
        og = Adam(gen.parameters(), lr=opt.G_lr, betas=(0.0, 0.9), device_step=graph)
        od = Adam(dis.parameters(), lr=opt.D_lr, betas=(0.0, 0.9), device_step=graph)
        fake = int(os.environ.get("HRV_FAKE_ALLREDUCE", "0") or 0) > 0     # 1-GPU overlap trace (tools/dp_overlap.sh)
        sg = og.make_grad_sync(graph=graph) if (world > 1 or fake) else None
        sd = od.make_grad_sync(graph=graph) if (world > 1 or fake) else None
        for s_ in (sg, sd):
            if s_ is not None:
                attach_grad_sync(s_)
        cg, cf = GANLoss("hinge"), L1Loss()

        def step(_i):
            _T.MMA_BF16[0] = bool(mixed)     # (another workload of this process may have switched it)
            x, parse7 = make_generator_inputs(opt, tocg, batch)
            generator_train_step(opt, gen, dis, cg, cf, crit_vgg, og, od, x, parse7, batch["image"], sg, sd)

        def parity():
            from oracle import step_check
            gp = os.path.join(ROOT, "gpurun_out")
            os.makedirs(gp, exist_ok=True)
            engines = (False, True) if mixed else (False,)
            # at the bench resolution with two images: the size-gated kernels of the timed path (patch tiles, LDS-DMA weight
            # gradients, thin convolutions, sub-batch launches) are the ones compared (~30 s of CPU autograd)
            reps = step_check.compare_generator_step(1024, 768, 64, 64, 2, seed=1, mixed=engines, cpu_threads=ctx["cpu_threads"],
                                                     table_path=os.path.join(gp, "bench_grad_parity_gen.txt"))
            out = {"fp32_engine_vs_oracle": reps[False],
                   "tolerance_fp32": "image / losses 1e-3 rel (north star); per-parameter gradients 2e-2 of max(|g|, 1e-3 module max)",
                   "tolerance_short": "fp32 engine: image, losses 1e-3 rel, grads 2e-2 of max|g|"}
            if mixed:
                out["bf16_engine_vs_oracle"] = reps[True]
                out["tolerance_short"] += ("; bf16 engine, against the fp32 oracle: image mean-abs 3e-3, losses 2e-3 (G) / 5e-3 (D), grad cosine >= "
                                           "0.99 for G and for D (the half-resolution PatchGAN's model0..2 forwards keep fp32 operands in the D step: 0.9975 / 0.9960 / "
                                           "0.9973 on seeds 1 / 2 / 3, 0.980 without); extras: configs[2] at b=8 is self-consistency (oracle "
                                           "at 1x1024x768 and 2x512x384); argmax: integer stage bit-exact, end to end the listed pixels at "
                                           "<= 286 ulps of margin")
                out["tolerance_bf16"] = ("operands carry 8 mantissa bits: image mean-abs 3e-3 (max 3e-2 of the range), loss terms "
                                         "2e-3 rel, gradient cosine >= 0.99 on every sizeable parameter")
            # the discriminator half of the same iteration (train_generator.py:327-360): D losses, every D gradient, D's Adam step
            dreps = step_check.compare_discriminator_step(1024, 768, 64, 64, 2, seed=1, mixed=engines, cpu_threads=ctx["cpu_threads"],
                                                          table_path=os.path.join(gp, "bench_grad_parity_dis.txt"))
            out["discriminator_half_fp32_engine_vs_oracle"] = dreps[False]
            if mixed:
                out["discriminator_half_bf16_engine_vs_oracle"] = dreps[True]
                out["tolerance_bf16_discriminator"] = ("loss terms 5e-3 rel; gradient cosine against the fp32 oracle >= 0.99 (round 5: PatchGAN model1's "
                                                       "forward on fp32 operands in the D step -- every convolution in bf16 gives 0.980, the oracle's own "
                                                       "bf16-operand evaluation 0.983: bf16_rounded_oracle_vs_fp32_oracle)")
            return out

        def cpu_baseline():
            from oracle import step_check
            r = step_check.cpu_train_generator_step(256, 192, 64, 64, 1, "more", repeats=3, warmup=1, threads=ctx["cpu_threads"])
            # ... and ONE iteration at the metric's own resolution: 1 image 1024x768 'most', no warm-up
            rf = step_check.cpu_train_generator_step(1024, 768, 64, 64, 1, "most", repeats=1, warmup=0, threads=ctx["cpu_threads"])
            return {"value": round(rf["images_per_s"], 5), "unit": "images/s", "cores": ctx["cpu_threads"], "kind": "port",
                    "sample": "oracle/step_check.cpu_train_generator_step: the whole train_generator.py iteration (G fwd/bwd, "
                              "PatchGAN x2, VGG + feat + hinge, Adam x2) on torch-CPU fp32, ONE image 1024x768 'most' ngf=64, one "
                              "timed iteration (the metric's resolution and generator depth)",
                    "seconds_per_step": round(rf["seconds_per_step_median"], 2),
                    "at_256x192_more": {"images_per_s": round(r["images_per_s"], 4),
                                        "seconds_per_step_median": round(r["seconds_per_step_median"], 3),
                                        "sample": "1 image 256x192 'more' (BASELINE.md section 4), 1 warm-up + 3 timed, median",
                                        "scaled_to_1024x768_images_per_s": round(r["images_per_s"] / 16.0, 5)}}
        timed = None
        if graph:
            from hr_viton_amd.graph import GraphedIteration
            timed = (lambda gi: (lambda _i: gi()))(GraphedIteration(lambda: step(0), (og, od), warmup=3))
        return dict(step=step, timed=timed, B=B, train=True, parity=parity, cpu_baseline=cpu_baseline, flops_per_img=8.8e12,
                    metric="1024x768 try-on images/sec (train_generator.py step: tocg+glue, G fwd/bwd, D fwd/bwd x2, VGG, Adam)",
                    workload="BASELINE configs[3] (SURVEY 8d config #4, headline): train_generator.py 1024x768, "
                             f"{B} img/GPU, " + ("--fp16 (bf16 MFMA operands, fp32 accumulate)" if mixed else "fp32") +
                             ", SPADE ngf=64 'most' + multiscale-D + VGG/feat-match, random-init weights" +
                             (" [whole iteration replayed as " + ("one hipGraph]" if world == 1 else "three hipGraph segments, the gradient "
                                                                "all-reduces between them]") if graph else ""),
                    traffic_tag="train_generator")
    gen.eval()

    def step(_i):
        tryon_step(opt, tocg, gen, batch)
    timed, gr = None, None
    if ctx["args"].graph or ctx.get("force_graph"):
        from hr_viton_amd.graph import graphed_tryon
        gr = graphed_tryon(opt, tocg, gen, batch)
        timed = (lambda g: (lambda _i: g(batch)))(gr)

    def parity():
        """configs[4]: (a) the hipGraph replay at the TIMED batch against the eager step on the same inputs -- bit for bit;
        (b) image 0 of the batch end to end (tocg@256x192 -> glue -> warp -> generator at 1024x768) against the oracle's
        composition with the engine's rounding points."""
        from oracle import step_check
        out = {}
        if gr is not None:
            with torch.no_grad():
                want = tryon_step(opt, tocg, gen, batch)
                want = {k: want[k].clone() for k in ("output", "warped_cloth", "fake_parse", "fake_segmap")}
                got = gr(batch)
                torch.cuda.synchronize()
            out["hipgraph_replay_vs_eager"] = {"batch": B, "size": f"{B}x{H}x{W}",
                                               "bit_identical": {k: bool(torch.equal(got[k], want[k])) for k in want}}
            del want
        out["one_image_vs_oracle"] = step_check.compare_tryon_step(opt, tocg, gen, batch, mixed, cpu_threads=ctx["cpu_threads"])
        out["tolerance"] = ("bf16 engines: image mean-abs error against the rounded oracle <= 2x that oracle's own nudged "
                            "re-evaluation + 1e-4; label map mismatch <= 2x + 1e-3" if mixed else "fp32: 1e-3 rel (north star)")
        return out
    return dict(step=step, timed=timed, B=B, train=False, parity=parity, flops_per_img=1.73e12,
                metric="1024x768 try-on images/sec (test_generator.py step: tocg@256x192 + glue + SPADE generator)",
                workload=f"BASELINE configs[4]: end-to-end test_generator.py step 1024x768, {B} img/GPU, ngf=64, random-init "
                         "weights" + (" (bf16 storage in the generator; tocg + glue f32)" if mixed else " fp32") +
                         (" [hipGraph replay]" if timed else ""), traffic_tag="tryon_infer")


def make_workload(ctx, name, mixed, B):
    if name == "tocg_infer":
        return wl_tocg_infer(ctx, mixed, B)
    if name == "train_condition":
        return wl_train_condition(ctx, mixed, B)
    return wl_generator(ctx, mixed, B, train=(name == "train_generator"))


def measure(ctx, wl, steps, warmup, mixed, dump=None):
    """Timed region (driver contract) + one extra profiled step with per-launch HIP events."""
    torch, hdist, ops = ctx["torch"], ctx["hdist"], ctx["ops"]
    dt = hdist.timed_steps(wl.get("timed") or wl["step"], steps, warmup, torch.cuda.synchronize)
    ops.profile_begin()
    wl["step"](0)
    recs = ops.profile_end(kernels=True)
    if dump and ctx["rank"] == 0:
        dump_launches(dump, recs)
    peak = PEAK_BF16_MFMA_TFLOPS if mixed else PEAK_F32_MFMA_TFLOPS
    s = summarize(recs, peak)
    B, world = wl["B"], ctx["world"]
    res = {"value": round(B * world * steps / dt, 3), "ms_per_step": round(dt / steps * 1e3, 3), "peak": peak,
           "summary": s, "dt": dt}
    return res


HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s (spec); 6.3 TB/s is the measured copy rate (hbm_kinds)


# Sustained ceilings of THIS part, measured (round 6): tools/probes/mfma_roof.hip, profiles/r06_mfma_roof.txt -- a register-resident
# v_mfma_f32_32x32x16_bf16 loop on all 256 CUs holds 2 433 TFLOP/s at 2.38 GHz on all-zero operands (the datasheet condition) and
# 1 866-1 872 TFLOP/s at 1.86-1.87 GHz / 1.32-1.36 kW of the 1.4 kW cap on random operands: the power limit, not the pipes, sets the
# rate real data can reach.  HBM: 6.29 TB/s measured copy rate (MI355X_MICROARCH.md) against the 8 TB/s specification.
ACHIEVABLE_BF16_TFLOPS = 1870.0
RIDGE_FLOP_PER_BYTE = 2500e12 / (HBM_PEAK_GBPS * 1e9)      # 312: below it a kernel is priced against HBM, whatever its MFMA count


def kernel_row(row, peak_tflops):
    """(kernel, launches, ms, flops, bytes) -> the roofline fields of ONE device-kernel family of the step.  Bound by arithmetic
    intensity (algorithmic FLOPs / algorithmic bytes against the ridge of the two peaks), not by "has FLOPs"."""
    kn, n, ms, fl, by = row
    mfma = fl > 0 and (by <= 0 or fl / by >= RIDGE_FLOP_PER_BYTE * (peak_tflops / 2500.0))
    ach = (fl / (ms * 1e-3) / 1e12) if mfma else (by / (ms * 1e-3) / 1e9)
    peak = peak_tflops if mfma else HBM_PEAK_GBPS
    ach_peak = (ACHIEVABLE_BF16_TFLOPS if peak_tflops >= 2000.0 else peak_tflops) if mfma else HBM_ACHIEVABLE_GBPS
    return {"kernel": "hrv::" + kn, "bound": "mfma" if mfma else "hbm", "achieved": round(ach, 2), "peak": peak,
            "unit": "TFLOP/s" if mfma else "GB/s", "frac": round(ach / peak, 4), "achievable_peak": ach_peak,
            "frac_of_achievable": round(ach / ach_peak, 4), "launches_per_step": n, "ms_per_step": round(ms, 3),
            "algorithmic_flops_per_launch": fl / max(1, n), "algorithmic_bytes_per_launch": round(by / max(1, n), 1)}


def roofline_obj(wl, res, north_star):
    """The object describes the DOMINANT kernel of the step: the device-kernel family with the largest summed HIP-event time over one
    iteration (what heads the per-kernel table of a rocprofv3 trace of the same command, template instances of one kernel summed) --
    achieved / frac from the durations of exactly those launches and their algorithmic FLOPs (or bytes, for an HBM-bound kernel), HBM
    traffic of exactly that kernel name from the PMC passes of this build.  ``kernels``: the same fields for the next families by
    time (spade_fused_kernel among them); ``north_star_set_*``: every 3x3 convolution launch of the SPADE generator (forward, data and
    weight gradients) -- the aggregate BASELINE.json's north star prices."""
    s = res["summary"]
    rows = [kernel_row(r, res["peak"]) for r in s["by_kernel"] if r[2] > 0]
    out = dict(rows[0]) if rows else {"kernel": None, "bound": "mfma", "achieved": 0.0, "peak": res["peak"], "unit": "TFLOP/s", "frac": 0.0}
    fam = out["kernel"][5:] if out["kernel"] else None
    per_step, src = load_traffic(wl["traffic_tag"], fam) if fam else (None, None)
    traffic = per_step / out["launches_per_step"] if per_step else None
    alg = out.get("algorithmic_bytes_per_launch")
    out.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch of this kernel (PMC FETCH_SIZE x2 + WRITE_SIZE over its dispatches / its launches)",
                "traffic_source": src, "wasted_traffic_ratio": round(traffic / alg, 3) if (traffic and alg) else None,
                "launch_unit": "one call of the kernel's C entry point (HIP events around it on the launch stream); a layer wider than one "
                               "column pass takes several dispatches per call, so compare TOTAL kernel time per iteration with a trace",
                "kernels": rows[1:8]})
    if north_star:
        out.update({"north_star_set_frac": s["spade"]["frac"], "north_star_set_achieved": s["spade"]["achieved"],
                    "north_star_set_ms": s["spade"]["ms_per_step"], "north_star_set_launches": s["spade"]["launches"],
                    "spade_gamma_beta_family": dict(s["gb"], note="fused forwards (hrv::spade_fused_kernel) + the pair data gradients "
                                                                  "(hrv::conv_p2_kernel) of the levels with >= 2 tiles per CU"),
                    "spade_3x3_set": dict(s["spade"], note="every 3x3 convolution launch of the SPADE generator (conv_shared / gamma|beta / "
                                                           "conv_0 / conv_1 / stems / conv_img: forward, data and weight gradients)")})
    out.update({"whole_step_conv_family": s["all"],
                "end_to_end_TFLOPs_vs_survey_work": round(wl["B"] * wl["flops_per_img"] / (res["dt"] / res["steps"]) / 1e12, 2),
                "hbm_kinds": s["hbm"],
                "slowest_launches": [{"name": r[1], "ms": round(r[4], 3), "TFLOPs": round(r[2] / (r[4] * 1e-3) / 1e12, 1)}
                                     for r in s["top"]]})
    return out


# ------------------------------------------------------------------------------------------------- the line the driver parses
LINE_LIMIT = 3800          # (round 4's 21.8 KB line could not be parsed by the driver; the contract test holds the line under 4 KB)


def _sig(v, n=4):
    if isinstance(v, float):
        return float(f"{v:.{n}g}")
    return v


def _short(sv, n):
    sv = str(sv)
    return sv if len(sv) <= n else sv[:n - 3] + "..."


def _worst(d, out=None, path=""):
    """every numeric leaf of a parity block whose name says it is an error / mismatch / cosine figure -> {dotted.name: value}"""
    out = {} if out is None else out
    if isinstance(d, dict):
        for k, v in d.items():
            if "rounded_oracle_vs" in str(k) or "oracle_vs_nudged" in str(k):
                continue        # (the oracle's OWN bf16-operand / nudged re-evaluation: a yardstick in the detail file, not an engine error)
            ks = str(k).lower() if isinstance(k, bool) else str(k)      # (engine keys False / True: as json.dump writes them)
            _worst(v, out, f"{path}.{ks}" if path else ks)
    elif isinstance(d, bool):
        if "bit_identical" in path:
            out[path] = d
    elif isinstance(d, (int, float)):
        leaf = path.rsplit(".", 1)[-1]
        if any(t in leaf for t in ("err", "cosine", "mismatch_pixels", "mismatch_frac", "worst_rel", "median_rel")) or "loss_rel_err" in path:
            out[path] = _sig(float(d)) if isinstance(d, float) else d
    return out


def _parity_summary(p):
    """a parity block -> its worst numbers only: max over the error-like leaves, min over the cosines, AND of the bit-identity flags"""
    if not p:
        return None
    leaves = _worst(p)
    cos = [v for k, v in leaves.items() if "cosine" in k and not isinstance(v, bool)]
    errs = {k: v for k, v in leaves.items() if "cosine" not in k and "median" not in k and not isinstance(v, bool)}
    bits = [v for v in leaves.values() if isinstance(v, bool)]
    out = {}
    groups = {}
    for k, v in errs.items():
        top = k.split(".", 1)[0]
        leaf = k.rsplit(".", 1)[-1] if "loss_rel_err" not in k else "loss_rel_err"
        if "." not in k:                      # a number at the top level of the block
            out[k] = v
            continue
        g = groups.setdefault(top, {})
        g[leaf] = max(g.get(leaf, 0), v)
    for k, v in leaves.items():
        if "cosine" in k and not isinstance(v, bool) and "." in k:
            g = groups.setdefault(k.split(".", 1)[0], {})
            g["min_cosine"] = min(g.get("min_cosine", 1.0), v)
    out.update(groups)
    if cos:
        out["min_cosine"] = min(cos)
    if bits:
        out["all_bit_identical"] = all(bits)
    return out


def _roof_compact(r):
    if not r:
        return None
    keys = ("bound", "kernel", "achieved", "peak", "unit", "frac", "achievable_peak", "frac_of_achievable", "launches_per_step", "ms_per_step", "algorithmic_flops_per_launch",
            "algorithmic_bytes_per_launch", "traffic", "wasted_traffic_ratio", "north_star_set_frac", "north_star_set_achieved",
            "north_star_set_ms")
    out = {k: _sig(r[k], 6) for k in keys if k in r}
    if r.get("traffic_source"):
        out["traffic_source"] = _short(r["traffic_source"].split(" ")[0], 80)
    if r.get("whole_step_conv_family"):
        out["whole_step_conv_frac"] = r["whole_step_conv_family"]["frac"]
    out["kernels"] = {k["kernel"][5:]: {"ms": k["ms_per_step"], "frac": k["frac"], "bound": k["bound"]} for k in (r.get("kernels") or [])[:5]}
    return out


def compact_line(full, detail_path=None):
    """The full result of a run (every table, note and sub-object: written to ``detail_path``) -> the ONE line the driver parses:
    the contract keys, the dominant kernel's roofline, the CPU baseline, the worst parity numbers, value / ms / frac of the extra
    configurations.  Always under LINE_LIMIT bytes (sheds optional parts until it is)."""
    cfg = dict(full.get("config") or {})
    cfg["workload"] = _short(cfg.get("workload", ""), 150)
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data")}
    line["metric"] = _short(line["metric"], 120)
    line["config"] = cfg
    line["roofline"] = _roof_compact(full.get("roofline"))
    c = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not c else {k: (_short(v, 120) if k == "sample" else v) for k, v in c.items()
                                               if k in ("value", "unit", "cores", "kind", "sample", "seconds_per_step")}
    line["parity"] = _parity_summary(full.get("parity"))
    if line["parity"] and (full.get("parity") or {}).get("tolerance_short"):
        line["parity"]["tolerance"] = _short(full["parity"]["tolerance_short"], 300)
    if full.get("extra"):
        line["extra"] = {}
        for k, e in full["extra"].items():
            r = e.get("roofline") or {}
            line["extra"][k] = {"value": e.get("value"), "ms_per_step": e.get("ms_per_step"), "batch": e.get("batch"),
                                "kernel": (r.get("kernel") or "")[5:], "frac": r.get("frac")}
            if not k.startswith("experimental_"):      # (a mode outside BASELINE's configs: value only, its parity block is in the detail file)
                line["extra"][k]["parity"] = _parity_summary(e.get("parity"))
    if detail_path:
        line["detail"] = detail_path
    # shed optional parts (least important first) until the line fits
    for drop in (None, ("extra", "parity"), ("roofline", "kernels"), ("parity", None), ("extra", None)):
        if drop is not None:
            a_, b_ = drop
            if b_ is None:
                line[a_] = None if a_ == "parity" else {k: {"value": v.get("value"), "ms_per_step": v.get("ms_per_step")} for k, v in (line.get(a_) or {}).items()}
            elif a_ == "extra":
                for v in (line.get("extra") or {}).values():
                    v.pop(b_, None)
            elif isinstance(line.get(a_), dict):
                line[a_].pop(b_, None)
        txt = json.dumps(line, separators=(",", ":"), allow_nan=False)
        if len(txt) < LINE_LIMIT:
            return txt
    raise AssertionError(f"bench line does not fit {LINE_LIMIT} bytes: {len(txt)}")


def emit(full):
    """Write the full result next to the run (gpurun_out/bench_detail.json; stderr names it) and print the compact line LAST on stdout."""
    rel = os.path.join("gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(full, f, indent=1)
        _log(f"full result (every table / note / parity object): {rel}")
    except OSError as e:      # a read-only tree: the line still goes out
        _log(f"could not write {rel}: {e}")
        rel = None
    sys.stderr.flush()
    print(compact_line(full, rel), flush=True)


def self_launch(n):
    """``python bench.py --gpus N`` started WITHOUT a launcher (the way the driver starts the N=1 run): re-exec this
    command line as N ranks under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1, a free port) and
    hand its exit code back.  Rank 0 of the children prints the JSON line; this parent prints nothing."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    _log(f"--gpus {n} without a launcher: re-exec as {n} ranks: {' '.join(cmd[1:8])} ...")
    return subprocess.call(cmd, env=env)


def dist_fields(tdist):
    """config fields that say what carried the N > 1 run: ``rccl_ranks`` is the world size ONLY when the process group's
    backend is nccl (= RCCL on ROCm) -- a gloo run (CPU tests, ranks time-slicing one GPU) reports 0 there."""
    if not tdist.is_initialized():
        return {"dist_backend": "none (single process)", "world_size": 1, "rccl_ranks": 0}
    be = str(tdist.get_backend())
    ws = int(tdist.get_world_size())
    return {"dist_backend": be, "world_size": ws, "rccl_ranks": ws if be == "nccl" else 0}


def wl_stub(ctx, B):
    """CPU-only stand-in workload (tests/test_bench_selflaunch.py: the launcher / collective / JSON plumbing of
    ``bench.py --gpus N`` on a box without GPUs, HRV_DIST_BACKEND=gloo).  Never a measurement."""
    torch = ctx["torch"]
    import torch.distributed as tdist
    B = B or 2
    g = torch.Generator().manual_seed(ctx["hdist"].shard_seed(7, ctx["rank"]))
    w = torch.randn(64, 64, generator=g)

    def step(_i):
        y = (w @ w).sum().reshape(1)
        if tdist.is_initialized():
            tdist.all_reduce(y)        # the data-path collective of the training workloads (gradient all-reduce)
    return dict(step=step, B=B, train=True, parity=None, flops_per_img=0.0, metric="stub (plumbing test, not a measurement)",
                workload="stub: CPU matmul + all-reduce", traffic_tag="stub")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (cpu_baseline, parity)")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[4] / configs[1] sub-measurements")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU legs (0 = min(cores, 32))")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch table of one step to this file")
    ap.add_argument("--workload", default="train_generator",
                    choices=["train_generator", "tryon_infer", "tocg_infer", "train_condition", "stub"],
                    help="train_generator = BASELINE configs[3], the headline (default); tryon_infer = configs[4]; "
                         "tocg_infer = configs[1]; train_condition = configs[2]")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0: the config's own)")
    ap.add_argument("--bf16", action="store_true", help="bf16 matrix cores (default for train_generator / tryon_infer)")
    ap.add_argument("--fp32", action="store_true", help="fp32 engine (default for tocg_infer / train_condition)")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="CUs the persistent kernels leave free (hrv_set_reserved_cus) for kernels that run concurrently with them: "
                         "RCCL's collectives under data parallelism.  Default: HRV_RESERVE_CUS or 0")
    ap.add_argument("--graph", action="store_true", help="tryon_infer / train_generator: replay the step as one captured hipGraph (train_generator on N > 1 GPUs: graph "
                         "segments with the gradient all-reduces between them)")
    args = ap.parse_args()
    mixed = (args.workload in ("train_generator", "tryon_infer") or args.bf16) and not args.fp32
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.nn as nn
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import dist as hdist
    from hr_viton_amd import ops

    if args.reserve_cus >= 0 and args.workload != "stub":
        from hr_viton_amd import _lib as _hl
        _hl.check(_hl.load().hrv_set_reserved_cus(args.reserve_cus), "hrv_set_reserved_cus")
    rank, local_rank, world = hdist.init_from_env()      # nccl (= RCCL) unless HRV_DIST_BACKEND overrides it
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    import torch.distributed as tdist
    if args.workload == "stub":
        ctx = dict(torch=torch, nn=nn, hdist=hdist, ops=ops, rank=rank, world=world, dev=torch.device("cpu"), args=args)
        wl = wl_stub(ctx, args.batch)
        dt = hdist.timed_steps(wl["step"], args.steps, args.warmup, None)
        if rank == 0:
            print(json.dumps({"metric": wl["metric"], "value": round(wl["B"] * world * args.steps / dt, 3), "unit": "images/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": wl["workload"], "global_batch": wl["B"] * world,
                                         "parallelism": f"dp{world}-allreduce",
                                         **dist_fields(tdist)},
                              "roofline": None, "cpu_baseline": None}), flush=True)
        if tdist.is_initialized():
            tdist.barrier()
            tdist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    local_dev = local_rank % ndev        # == local_rank on a real node; lets a 1-GPU box smoke-test the N>1 logic
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    ctx = dict(torch=torch, nn=nn, hdist=hdist, ops=ops, rank=rank, world=world, dev=dev, args=args,
               cpu_threads=args.cpu_threads or min(os.cpu_count() or 1, 32))

    _log(f"building workload {args.workload} ({'bf16' if mixed else 'fp32'})")
    wl = make_workload(ctx, args.workload, mixed, args.batch)
    _log("timed region")
    res = measure(ctx, wl, args.steps, args.warmup, mixed, args.dump_launches)
    _log(f"{res['value']} images/s, {res['ms_per_step']} ms/step")
    res["steps"] = args.steps
    s = res["summary"]
    line = None
    if rank == 0:
        line = {"metric": wl["metric"], "value": res["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                "dtype": "bf16" if mixed else "f32",
                "dtype_note": "bf16 MFMA operands (matrix-core-only tensors stored bf16), f32 accumulate / norms / optimizer" if mixed else "f32",
                "data": "synthetic",
                "config": {"workload": wl["workload"], "global_batch": wl["B"] * world, "height": H, "width": W,
                           "parallelism": f"dp{world}" + ("-allreduce" if wl["train"] else "-replicas"),
                           "persistent_grid_cus": int(ops._lib.load().hrv_persistent_cus()),
                           **dist_fields(tdist)},
                "roofline": roofline_obj(wl, res, north_star=(args.workload in ("train_generator", "tryon_infer"))),
                "per_kind_ms": {k: {"launches": v[0], "ms": round(v[1], 2)} for k, v in sorted(s["kinds"].items())},
                "cpu_baseline": None, "parity": None}
    cpu_legs = rank == 0 and world == 1 and not args.no_cpu_baseline
    if cpu_legs:
        torch.set_num_threads(ctx["cpu_threads"])
        if wl.get("cpu_baseline"):
            _log("cpu_baseline leg")
            line["cpu_baseline"] = wl["cpu_baseline"]()
        if wl.get("parity"):
            _log("parity leg")
            line["parity"] = wl["parity"]()
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "train_generator":
        # the other two measured configurations of BASELINE.json, same process, after the headline's timed region
        extra = {}
        del wl
        torch.cuda.empty_cache()
        for key, name, mx, st in (("config5_tryon_infer_bf16_b16", "tryon_infer", True, 5),
                                  ("config2_tocg_infer_f32_b4", "tocg_infer", False, 10),
                                  ("config3_train_condition_f32_b8", "train_condition", False, 2),
                                  ("experimental_train_condition_bf16_operands_b8", "train_condition", True, 3)):
            from hr_viton_amd import train_ops as _T
            _T.MMA_BF16[0] = False
            _log(f"extra: {key}")
            ctx["force_graph"] = name == "tryon_infer"      # configs[4] names the hipGraph-captured decode: time the replay
            w2 = make_workload(ctx, name, mx, 0)
            r2 = measure(ctx, w2, st, 2, mx)
            ctx["force_graph"] = False
            r2["steps"] = st
            e = {"metric": w2["metric"], "value": r2["value"], "unit": "images/s", "ms_per_step": r2["ms_per_step"],
                 "steps": st, "warmup": 2, "workload": w2["workload"], "batch": w2["B"],
                 "roofline": roofline_obj(w2, r2, north_star=(name == "tryon_infer"))}
            if cpu_legs and w2.get("parity"):
                e["parity"] = w2["parity"]()
            if key.startswith("experimental_"):
                e["note"] = ("NOT a BASELINE configs[2] result (configs[2] is fp32: the entry above).  bf16 conv operands turn the "
                             "discontinuous tocg loss (floor() of five warps, L1 sign(), ReLU masks) into a gradient whose cosine "
                             "against the fp32 oracle is ~0.88 -- for this engine AND for the oracle's own bf16-operand evaluation "
                             "(parity.bf16_rounded_oracle_vs_fp32_oracle).  Kept as a measured mode of train_condition.py --fp16, "
                             "outside the north-star tolerance.")
            extra[key] = e
            del w2
            torch.cuda.empty_cache()
        line["extra"] = extra
    elif cpu_legs and args.workload == "tocg_infer" and line["parity"] is not None:
        p = line["parity"]
        line["cpu_baseline"] = {"value": round(1.0 / p["oracle_forward_s"], 4), "unit": "images/s", "cores": ctx["cpu_threads"],
                                "kind": "port", "sample": "1 image 1024x768, one forward of oracle.tocg_forward"}
    if rank == 0:
        emit(line)
    if tdist.is_available() and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
