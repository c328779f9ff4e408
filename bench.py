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


def build_model(torch, nn):
    from argparse import Namespace
    from hr_viton_amd.networks import ConditionGenerator
    opt = Namespace(cuda=True, warp_feature="T1", out_layer="relu")
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = min(cores, 32))")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch table of one step to this file")
    args = ap.parse_args()

    import torch
    import torch.nn as nn
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import dist as hdist
    from hr_viton_amd import ops

    rank, local_rank, world = hdist.init_from_env("nccl" if args.gpus > 1 else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    opt, model = build_model(torch, nn)
    sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev)
    i1, i2 = make_inputs(torch, BATCH, hdist.shard_seed(1234, rank), dev)
    torch.cuda.synchronize()

    def step(_i):
        model(opt, i1, i2)

    dt = hdist.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize)
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
    roofline = {"bound": "mfma", "kernel": "hrv::conv_f32_mfma_kernel (all tile configs)",
                "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
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
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "BASELINE configs[1]: ConditionGenerator inference 1024x768 batch=4/GPU "
                                       "fp32, ngf=96, random-init weights",
                           "global_batch": BATCH * world, "height": H, "width": W, "parallelism": f"dp{world}-replicas"},
                "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity}
        print(json.dumps(line), flush=True)

    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
