#!/usr/bin/env python3
"""bench.py -- the hot path's throughput on MI355X (driver contract).

Workload (BASELINE.json configs[1]): ConditionGenerator inference, 1024x768,
batch 4 per GPU, fp32, random-init ngf=96 weights (+ randomised BatchNorm
statistics), synthetic VITON-HD-shaped inputs already resident in HBM.  A "step"
is one forward pass of the hot path (flow pyramid + seg logits + grid_sample
warps) over one batch.  N>1: one process per GPU (torch.distributed.run), each
rank owns its own batch (independent images => no data-path collective, "weak"
scaling); value = images all ranks processed / max-over-ranks time.

Extra objects on the JSON line:
  roofline     -- the dominant kernel family (implicit-GEMM fp32 MFMA conv):
                  algorithmic conv FLOPs of one step / summed per-launch HIP-event
                  durations of the conv launches of one step, vs 157.3 TFLOP/s.
  cpu_baseline -- the oracle (CPU restatement of the reference) timed on this
                  box's host cores on a bounded sample (1 image, 1 forward).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
H, W, BATCH, NGF = 1024, 768, 4, 96


def make_inputs(torch, n, seed, device):
    g = torch.Generator().manual_seed(seed)
    input1 = torch.cat([torch.rand(n, 3, H, W, generator=g) * 2 - 1,
                        (torch.rand(n, 1, H, W, generator=g) > 0.5).float()], 1)
    lab = torch.randint(0, 13, (n, 1, H, W), generator=g)
    input2 = torch.cat([torch.zeros(n, 13, H, W).scatter_(1, lab, 1.0),
                        torch.rand(n, 3, H, W, generator=g) * 2 - 1], 1)
    return input1.to(device), input2.to(device)


def build_model(torch, nn, mixed=False):
    from argparse import Namespace
    from hr_viton_amd.networks import ConditionGenerator
    opt = Namespace(cuda=True, warp_feature="T1", out_layer="relu", fp16=mixed)
    torch.manual_seed(0)
    m = ConditionGenerator(opt, 4, 16, 13, ngf=NGF, norm_layer=nn.BatchNorm2d)
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


def _emit(args, torch, hdist, ops, rank, world, B, step, metric, workload, flops_per_img, train):
    mixed = bool(getattr(args, "bf16", False))
    peak = 2500.0 if mixed else PEAK_F32_MFMA_TFLOPS
    dt = hdist.timed_steps(getattr(args, "_timed", None) or step, args.steps, args.warmup, torch.cuda.synchronize)
    ops.profile_begin()
    step(0)
    recs = ops.profile_end()
    kinds = {}
    for k, n, fl, by, ms in recs:
        a = kinds.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += fl
    if args.dump_launches and rank == 0:
        with open(args.dump_launches, "w") as f:
            for k, n, fl, by, ms in recs:
                f.write(f"{k:8s} {n:52s} {ms:9.4f} ms  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.2f} TFLOP/s\n")
    mf_ms = sum(v[1] for k, v in kinds.items() if k in ("conv", "wgrad"))
    mf_fl = sum(v[2] for k, v in kinds.items() if k in ("conv", "wgrad"))
    if rank == 0:
        ach = mf_fl / (mf_ms * 1e-3) / 1e12 if mf_ms > 0 else 0.0
        line = {"metric": metric, "value": round(B * world * args.steps / dt, 3), "unit": "images/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16 MFMA operands, f32 storage/accumulate" if mixed else "f32", "data": "synthetic",
                "config": {"workload": workload + (" [hipGraph replay]" if getattr(args, "_timed", None) else ""),
                           "global_batch": B * world,
                           "parallelism": f"dp{world}" + ("-allreduce" if train else "-replicas")},
                "roofline": {"bound": "mfma", "kernel": "hrv::conv_f32_mfma_kernel + hrv::conv_wgrad_mfma_kernel",
                             "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                             "frac": round(ach / peak, 4), "traffic": None,
                             "end_to_end_TFLOPs_vs_survey_work": round(B * flops_per_img / (dt / args.steps) / 1e12, 2)},
                "per_kind_ms": {k: {"launches": v[0], "ms": round(v[1], 2)} for k, v in sorted(kinds.items())},
                "cpu_baseline": None}
        print(json.dumps(line), flush=True)
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()


def cond_workload(args, torch, nn, hdist, ops, rank, local_rank, world, dev):
    """BASELINE configs[2]: train_condition.py 1024x768 b=8 fp32 --Ddownx2 --lasttvonly --interflowloss."""
    import train_condition as tc
    from hr_viton_amd.gen_train import attach_grad_sync
    from hr_viton_amd.losses import L1Loss
    from hr_viton_amd.networks import ConditionGenerator, GANLoss, VGGLoss, define_D
    from hr_viton_amd.optim import Adam
    from hr_viton_amd.parallel import broadcast_module
    from hr_viton_amd.pipeline import condition_train_step
    B = args.batch or 8
    opt = tc.get_opt(["--name", "bench", "--synthetic", "-b", str(B * world), "--fine_height", "1024", "--fine_width",
                      "768", "--Ddownx2", "--lasttvonly", "--interflowloss"] + (["--fp16"] if args.bf16 else []))
    if args.bf16:
        from hr_viton_amd import train_ops as _T
        _T.MMA_BF16[0] = True
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
    batch = tc.synthetic_batch(opt, B, hdist.shard_seed(4321, rank), dev)

    def step(_i):
        condition_train_step(opt, tocg, D, l1, crit_vgg, gan, og, od, batch, sg, sd)
    _emit(args, torch, hdist, ops, rank, world, B, step,
          "1024x768 images/sec (train_condition.py step: tocg fwd/bwd with batch-stat BN, 5 VGG pairs, LSGAN D, Adam)",
          "BASELINE configs[2]: train_condition.py 1024x768 fp32 --Ddownx2 --lasttvonly --interflowloss, ngf=96, "
          "random-init weights", 13e12, True)


def other_workloads(args, torch, nn, hdist, ops, rank, local_rank, world, dev):
    """Secondary workloads (not the driver's default bench): same timing contract, same JSON shape."""
    from argparse import Namespace
    import train_generator as tg
    from hr_viton_amd.gen_train import attach_grad_sync
    from hr_viton_amd.losses import GANLoss, L1Loss
    from hr_viton_amd.network_generator import MultiscaleDiscriminator, SPADEGenerator
    from hr_viton_amd.networks import ConditionGenerator
    from hr_viton_amd.optim import Adam
    from hr_viton_amd.parallel import broadcast_module
    from hr_viton_amd.pipeline import generator_train_step, make_generator_inputs, tryon_step
    from hr_viton_amd.vgg import VGGLoss
    if args.workload == "train_condition":
        return cond_workload(args, torch, nn, hdist, ops, rank, local_rank, world, dev)
    train = args.workload == "train_generator"
    B = args.batch or 4
    opt = tg.get_opt(["--name", "bench", "--synthetic", "-b", str(B * world)] + (["--fp16"] if args.bf16 else []))
    torch.manual_seed(0)
    tocg = ConditionGenerator(opt, 4, 16, 13, ngf=96, norm_layer=nn.BatchNorm2d).to(dev).eval()
    gen = SPADEGenerator(opt, 9)
    gen.init_weights("xavier", 0.02)
    gen.to(dev)
    batch = tg.synthetic_batch(opt, B, hdist.shard_seed(1234, rank), dev)
    if train:
        if args.bf16:
            from hr_viton_amd import train_ops as _T
            _T.MMA_BF16[0] = True      # mixed precision: bf16 matrix cores over fp32 tensors
        dis = MultiscaleDiscriminator(opt)
        dis.init_weights("xavier", 0.02)
        dis.to(dev).train()
        gen.train()
        crit_vgg = VGGLoss(opt).to(dev)
        for m in (gen, dis, crit_vgg):
            broadcast_module(m)
        og = Adam(gen.parameters(), lr=opt.G_lr, betas=(0.0, 0.9))
        od = Adam(dis.parameters(), lr=opt.D_lr, betas=(0.0, 0.9))
        sg = og.make_grad_sync() if world > 1 else None
        sd = od.make_grad_sync() if world > 1 else None
        for s_ in (sg, sd):
            if s_ is not None:
                attach_grad_sync(s_)
        cg, cf = GANLoss("hinge"), L1Loss()

        def step(_i):
            x, parse7 = make_generator_inputs(opt, tocg, batch)
            generator_train_step(opt, gen, dis, cg, cf, crit_vgg, og, od, x, parse7, batch["image"], sg, sd)
        metric = "1024x768 try-on images/sec (train_generator.py step: tocg+glue, G fwd/bwd, D fwd/bwd x2, VGG, Adam)"
        flops_per_img = 8.8e12
    else:
        gen.eval()

        def step(_i):
            tryon_step(opt, tocg, gen, batch)
        if args.graph:
            from hr_viton_amd.graph import graphed_tryon
            args._timed = (lambda g: (lambda _i: g(batch)))(graphed_tryon(opt, tocg, gen, batch))
        metric = "1024x768 try-on images/sec (test_generator.py step: tocg@256x192 + glue + SPADE generator)"
        flops_per_img = 1.73e12
    if train:
        _emit(args, torch, hdist, ops, rank, world, B, step, metric, "train_generator 1024x768 ngf=64, random-init weights",
              flops_per_img, True)
    else:
        _emit(args, torch, hdist, ops, rank, world, B, step, metric, "tryon_infer 1024x768 ngf=64, random-init weights"
              + (" (bf16 storage in the generator; tocg + glue f32)" if args.bf16 else ""), flops_per_img, False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = min(cores, 32))")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch table of one step to this file")
    ap.add_argument("--workload", default="tocg_infer", choices=["tocg_infer", "train_generator", "tryon_infer", "train_condition"],
                    help="tocg_infer = BASELINE configs[1] (default, the driver's bench); train_generator = configs[3] "
                         "shape in fp32 (4 img/GPU, G+D step incl. VGG, DP all-reduce); tryon_infer = end-to-end "
                         "test_generator.py step (configs[4] shape, fp32)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch for the non-default workloads")
    ap.add_argument("--bf16", action="store_true", help="tryon_infer: run the SPADE generator on the bf16 engine")
    ap.add_argument("--graph", action="store_true", help="inference workloads: replay the step as one captured hipGraph "
                                                         "(hr_viton_amd.graph); the per-launch roofline leg stays eager")
    args = ap.parse_args()

    import torch
    import torch.nn as nn
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import dist as hdist
    from hr_viton_amd import ops

    rank, local_rank, world = hdist.init_from_env()      # nccl (= RCCL) unless HRV_DIST_BACKEND overrides it
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    ndev = torch.cuda.device_count()
    local_dev = local_rank % ndev        # == local_rank on a real node; lets a 1-GPU box smoke-test the N>1 logic
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if args.workload != "tocg_infer":
        return other_workloads(args, torch, nn, hdist, ops, rank, local_rank, world, dev)

    opt, model = build_model(torch, nn, mixed=args.bf16)   # --bf16: bf16 MFMA operands over fp32 tensors (not the default)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev)
    i1, i2 = make_inputs(torch, BATCH, hdist.shard_seed(1234, rank), dev)
    torch.cuda.synchronize()

    def step(_i):
        model(opt, i1, i2)

    timed = step
    if args.graph:
        from hr_viton_amd.graph import graphed_condition
        g = graphed_condition(opt, model, i1, i2)
        feed = {"input1": i1, "input2": i2}

        def timed(_i):
            g(feed)
    dt = hdist.timed_steps(timed, args.steps, args.warmup, torch.cuda.synchronize)
    images = BATCH * world * args.steps
    value = images / dt

    # ---- roofline leg: per-launch HIP events (torch's current stream IS the launch stream)
    ops.profile_begin()
    step(0)
    recs = ops.profile_end()
    if args.dump_launches and rank == 0:
        with open(args.dump_launches, "w") as f:
            for k, n, fl, by, ms in recs:
                f.write(f"{k:8s} {n:44s} {ms:9.4f} ms  {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.2f} TFLOP/s  "
                        f"{by / (ms * 1e-3) / 1e9 if ms > 0 else 0:9.1f} GB/s\n")
    conv = [r for r in recs if r[0] == "conv"]
    conv_flops = sum(r[2] for r in conv)
    conv_ms = sum(r[4] for r in conv)
    other_ms = sum(r[4] for r in recs if r[0] != "conv")
    other_bytes = sum(r[3] for r in recs if r[0] != "conv")
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    top = sorted(conv, key=lambda r: -r[4])[:5]
    # HBM traffic cannot be read from inside the process: it comes from the committed rocprofv3 PMC
    # passes of this same command (profiles/, see tools/profile_pmc.sh), per conv launch.
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tp):
        with open(tp) as f:
            tj = json.load(f)
        traffic, traffic_src = tj.get("hbm_bytes_per_launch"), "profiles/r01_pmc_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)"
    peak = 2500.0 if args.bf16 else PEAK_F32_MFMA_TFLOPS
    if args.bf16:
        traffic, traffic_src = None, None      # the committed PMC passes are of the fp32 default
    roofline = {"bound": "mfma", "kernel": "hrv::conv_mfma_kernel (all tile configs)",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per conv launch (average over the step's launches)", "traffic_source": traffic_src,
                "algorithmic_flops_per_launch": conv_flops / max(1, len(conv)),
                "launches_per_step": len(conv), "flops_per_step": conv_flops,
                "conv_ms_per_step": round(conv_ms, 3),
                "hbm_kernels_ms_per_step": round(other_ms, 3),
                "hbm_kernels_GBps": round(other_bytes / (other_ms * 1e-3) / 1e9, 1) if other_ms > 0 else None,
                "slowest_launches": [{"name": r[1], "ms": round(r[4], 3),
                                      "TFLOPs": round(r[2] / (r[4] * 1e-3) / 1e12, 1)} for r in top]}

    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import hrviton_oracle as O
        c1, c2 = i1[:1].cpu(), i2[:1].cpu()
        # all 256 host threads of the GPU box are slower than 32 on this shape (oversubscribed oneDNN)
        torch.set_num_threads(args.cpu_threads or min(os.cpu_count() or 1, 32))
        t0 = time.perf_counter()
        with torch.no_grad():
            want = O.tocg_forward(sd_cpu, c1, c2)
        cdt = time.perf_counter() - t0
        cpu_baseline = {"value": round(1.0 / cdt, 4), "unit": "images/s", "cores": torch.get_num_threads(),
                        "kind": "port", "sample": "1 image 1024x768, one forward of oracle.tocg_forward "
                        "(torch CPU fp32 restatement of networks.py:98-159), no warm-up"}
        got = model(opt, i1[:1], i2[:1])

        def rel(a, b):
            return float(((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-12)).item())

        lab_g, lab_w = got[1].cpu().argmax(1), want[1].argmax(1)
        parity = {"flow_last_max_rel_err": rel(got[0][-1], want[0][-1]), "seg_max_rel_err": rel(got[1], want[1]),
                  "warped_cloth_max_rel_err": rel(got[2], want[2]),
                  "argmax_mismatch_pixels": int((lab_g != lab_w).sum().item()), "pixels": int(lab_w.numel())}

    if rank == 0:
        line = {"metric": "1024x768 try-on images/sec (ConditionGenerator inference: flow+seg+grid_sample)",
                "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16 MFMA operands, f32 storage/accumulate" if args.bf16 else "f32", "data": "synthetic",
                "config": {"workload": "BASELINE configs[1]: ConditionGenerator inference 1024x768 batch=4/GPU "
                                       "fp32, ngf=96, random-init weights" + (" [hipGraph replay]" if args.graph else ""),
                           "global_batch": BATCH * world, "height": H, "width": W, "parallelism": f"dp{world}-replicas"},
                "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity}
        print(json.dumps(line), flush=True)

    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
