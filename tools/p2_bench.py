#!/usr/bin/env python3
"""A/B of csrc/conv_p2.hip (two blocks per CU, HRV_CONV_P2=1) against what served the same layers before (HRV_CONV_P2=0: the
generic patch tiles; for the SPADE pair data gradient csrc/spade_gb.hip) at the bench sizes, interleaved rounds in ONE process,
median.      python tools/p2_bench.py [rounds]      (via gpurun)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import ops, train_ops as T  # noqa: E402

VGG = [("vgg.5   64->128 @512x384", 64, 128, 512, 384), ("vgg.7  128->128 @512x384", 128, 128, 512, 384),
       ("vgg.10 128->256 @256x192", 128, 256, 256, 192), ("vgg.12 256->256 @256x192", 256, 256, 256, 192),
       ("vgg.19 256->512 @128x96", 256, 512, 128, 96), ("vgg.21 512->512 @128x96", 512, 512, 128, 96)]
GB = [("up_4.norm_0.gb.dgrad", 80, 1024, 768), ("up_4.norm_1.gb.dgrad", 32, 1024, 768), ("up_3.norm_0.gb.dgrad", 144, 512, 384),
      ("up_3.norm_1.gb.dgrad", 64, 512, 384), ("up_2.norm_0.gb.dgrad", 272, 256, 192)]


def bench(fn, rounds):
    ts = {"1": [], "0": []}
    for rd in range(rounds + 2):
        for flag in ("1", "0"):
            os.environ["HRV_CONV_P2"] = flag
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            if rd >= 2:
                ts[flag].append(s.elapsed_time(e))
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


# round 6: the generator's coarse levels (fewer tiles than resident blocks: the passes of a tile run on different CUs)
COARSE = [("up_1.conv_0 528->256 @128x96", 528, 256, 128, 96), ("up_1.conv_1 256->256 @128x96", 256, 256, 128, 96),
          ("up_0.conv_0 1040->512 @64x48", 1040, 512, 64, 48), ("up_0.conv_1 512->512 @64x48", 512, 512, 64, 48),
          ("G_middle_1.conv_0 1040->1024 @32x24", 1040, 1024, 32, 24), ("G_middle_1.conv_1 1024->1024 @32x24", 1024, 1024, 32, 24),
          ("G_middle_0.conv_1 1024->1024 @16x12", 1024, 1024, 16, 12)]


def coarse(rounds):
    for q4 in ("3", "1"):
        os.environ["HRV_CONV_P2_MIN_TILES_X4"] = q4
        from hr_viton_amd import _lib
        _lib.load().hrv_diag_reload_env()
        for name, cin, cout, H, W in COARSE:
            N = 4
            x = ops.Act(torch.randn(N, H, W, (cin + 7) // 8 * 8, device="cuda").to(torch.bfloat16), cin)
            w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.02
            b = torch.zeros(cout, device="cuda")
            dy = ops.Act(torch.randn(N, H, W, cout, device="cuda").to(torch.bfloat16), cout)
            fl = 2.0 * N * H * W * cin * cout * 9
            f = bench(lambda: T.conv_forward_dev(w, [(x, 0)], 1, 1, shift=b, act=ops.ACT_NONE, out_bf16=False, name="l"), rounds)
            d = bench(lambda: T.conv_dgrad(dy, w, H, W, 1, 1, out_bf16=True, name="l.dgrad"), rounds)
            print(f"[min quarter-units per CU {q4}] {name}: fwd p2 {f['1'] * 1e3:.0f} us {fl / f['1'] / 1e9:7.1f} TF/s | generic {f['0'] * 1e3:.0f} us {fl / f['0'] / 1e9:7.1f}   "
                  f"dgrad p2 {d['1'] * 1e3:.0f} us {fl / d['1'] / 1e9:7.1f} | generic {d['0'] * 1e3:.0f} us {fl / d['0'] / 1e9:7.1f}   (incl. pack)", flush=True)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    T.MMA_BF16[0] = True
    torch.manual_seed(0)
    if len(sys.argv) > 2 and sys.argv[2] == "coarse":
        return coarse(rounds)
    for N in (4, 8):
        for name, cin, cout, H, W in VGG:
            x = ops.Act(torch.relu(torch.randn(N, H, W, cin, device="cuda")).to(torch.bfloat16), cin)
            w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.03
            b = torch.zeros(cout, device="cuda")
            dy = ops.Act(torch.randn(N, H, W, cout, device="cuda").to(torch.bfloat16), cout)
            fl = 2.0 * N * H * W * cin * cout * 9
            f = bench(lambda: T.conv_forward_dev(w, [(x, 0)], 1, 1, shift=b, act=ops.ACT_RELU, out_bf16=True, name="l"), rounds)
            d = bench(lambda: T.conv_dgrad(dy, w, H, W, 1, 1, act_mask=x, slope=0.0, out_bf16=True, name="l.dgrad"), rounds)
            print(f"N={N} {name}: fwd p2 {f['1']:.3f} ms {fl / f['1'] / 1e9:7.1f} TF/s | generic {f['0']:.3f} ms {fl / f['0'] / 1e9:7.1f} TF/s   "
                  f"dgrad p2 {d['1']:.3f} ms {fl / d['1'] / 1e9:7.1f} | generic {d['0']:.3f} ms {fl / d['0'] / 1e9:7.1f}   (incl. pack)", flush=True)
    from argparse import Namespace
    from hr_viton_amd.network_generator import SPADENorm
    for name, Cc, H, W in GB:
        N = 4
        norm = SPADENorm(Namespace(), "aliasinstance", Cc, 7).cuda()
        wg, wb = norm.conv_gamma.weight.data, norm.conv_beta.weight.data
        actv = ops.Act(torch.relu(torch.randn(N, H, W, 128, device="cuda")).to(torch.bfloat16), 128)
        dgb = ops.Act(torch.randn(N, H, W, 2 * Cc, device="cuda").to(torch.bfloat16), 2 * Cc)
        dact = ops.Act(torch.empty(N, H, W, 128, device="cuda", dtype=torch.bfloat16), 128)
        fl = 2.0 * N * H * W * 2 * Cc * 128 * 9

        def run():
            if os.environ["HRV_CONV_P2"] == "1":
                T.conv_p2(dgb, T.conv_p2_pack(2, wg, wb, 2 * Cc, 128), 128, dact, mask=actv, mask_slope=0.0, name=name)
            else:
                T.spade_gb_dgrad(dgb, T.spade_gb_pack(1, wg, wb), Cc, actv, 0.0, dact, name)
        r = bench(run, rounds)
        print(f"{name} C={Cc} {H}x{W}: p2 {r['1']:.3f} ms {fl / r['1'] / 1e9:7.1f} TF/s | spade_gb {r['0']:.3f} ms {fl / r['0'] / 1e9:7.1f} TF/s   (incl. pack)",
              flush=True)


if __name__ == "__main__":
    main()
