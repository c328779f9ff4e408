#!/usr/bin/env python3
"""csrc/conv_s2.hip at the PatchGAN's shapes (train_generator.py iteration, 4 x 1024x768 per GPU: D sees [fake ; real] = 8 images in the
forward, 4 or 8 in the backward) next to the generic engine on the same layers: forward of model1 / model2 of both scales, their data
gradients (one launch vs four phase launches), model0 in its 2x2 cells form.      python tools/s2_bench.py [rounds]      (via gpurun)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import ops, train_ops as T  # noqa: E402


def timed(fn, rounds):
    ts = []
    for rd in range(rounds + 2):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        if rd >= 2:
            ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


LAYERS = [("discriminator_0.model1  64->128 in 513x385", 64, 128, 513, 385), ("discriminator_0.model2 128->256 in 257x193", 128, 256, 257, 193),
          ("discriminator_1.model1  64->128 in 257x193", 64, 128, 257, 193), ("discriminator_1.model2 128->256 in 129x97", 128, 256, 129, 97)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    T.MMA_BF16[0] = True
    torch.manual_seed(0)
    for N in (8, 4):
        for name, cin, cout, H, W in LAYERS:
            Ho, Wo = H // 2 + 1, W // 2 + 1
            xb = ops.Act(torch.randn(N, H, W, cin, device="cuda").to(torch.bfloat16), cin)
            xf = ops.Act(xb.t.float(), cin)
            w = torch.randn(cout, cin, 4, 4, device="cuda") * 0.03
            b = torch.zeros(cout, device="cuda")
            fl = 2.0 * N * Ho * Wo * cin * cout * 16
            out = ops.alloc(N, Ho, Wo, cout, "cuda")
            pk = T.conv_s2_pack(T.S2_FWD, w, cin, cout)
            t_new = timed(lambda: T.conv_s2(T.S2_FWD, xb, pk, cout, out, bias=b, name="l"), rounds)
            t_pack = timed(lambda: T.conv_s2_pack(T.S2_FWD, w, cin, cout), rounds)
            t_old = timed(lambda: T.conv_forward_dev(w, [(xf, 0)], 2, 2, shift=b, name="l"), rounds)
            dyb = ops.Act(torch.randn(N, Ho, Wo, cout, device="cuda").to(torch.bfloat16), cout)
            dyf = ops.Act(dyb.t.float(), cout)
            dx = ops.alloc(N, H, W, cin, "cuda", bf16=True)
            tap = ops.Act(torch.randn(N, H, W, cin, device="cuda").to(torch.bfloat16), cin)
            pkd = T.conv_s2_pack(T.S2_DGRAD, w, cout, 4 * cin, cin)
            d_new = timed(lambda: T.conv_s2(T.S2_DGRAD, dyb, pkd, 4 * cin, dx, Cph=cin, name="l.dgrad"), rounds)
            d_new2 = timed(lambda: T.conv_s2(T.S2_DGRAD, dyb, pkd, 4 * cin, dx, Cph=cin, residual=tap, mask=xb, mask_slope=0.2, name="l.dgrad"), rounds)
            d_old = timed(lambda: T.conv_dgrad(dyf, w, H, W, 2, 2, name="l.dgrad"), rounds)
            print(f"N={N} {name}: fwd s2 {t_new * 1e3:.0f} us {fl / t_new / 1e9:7.1f} TF/s (pack {t_pack * 1e3:.0f} us) | generic fp32-stored {t_old * 1e3:.0f} us "
                  f"{fl / t_old / 1e9:7.1f}    dgrad s2 {d_new * 1e3:.0f} us {fl / d_new / 1e9:7.1f} (+tap, mask: {d_new2 * 1e3:.0f} us) | generic 4 phases "
                  f"{d_old * 1e3:.0f} us {fl / d_old / 1e9:7.1f}", flush=True)
    # model0 as cells
    for N, H, W in ((8, 1024, 768), (8, 512, 384)):
        Cq = 12
        xs = ops.Act(torch.randn(N, H // 2, W // 2, 4 * Cq, device="cuda").to(torch.bfloat16), 4 * Cq)
        w2 = torch.randn(64, 4 * Cq, 2, 2, device="cuda") * 0.05
        b = torch.zeros(64, device="cuda")
        out = ops.alloc(N, H // 2 + 1, W // 2 + 1, 64, "cuda", bf16=True)
        pk = T.conv_s2_pack(T.S2_CELLS, w2, 4 * Cq, 64)
        t = timed(lambda: T.conv_s2(T.S2_CELLS, xs, pk, 64, out, bias=b, act=ops.ACT_LRELU, name="m0"), rounds)
        gb = (xs.t.numel() * 2 + out.t.numel() * 2) / 1e9
        print(f"model0 cells N={N} {H}x{W}: {t * 1e3:.0f} us  {gb / t * 1e3:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
