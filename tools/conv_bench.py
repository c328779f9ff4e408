#!/usr/bin/env python3
"""A/B micro-benchmark of the conv engine (tile config x kernel variant) on
representative layers of the path.  Interleaved rounds in ONE process, median."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import ops  # noqa: E402

LAYERS = [
    # name, [Cin...], Cout, k, stride, pad, N, H, W, residual
    ("seg4.block3 96->96 3x3 @1024x768", [96], 96, 3, 1, 1, 4, 1024, 768, True),
    ("enc1.block 192->192 3x3 @256x192", [192], 192, 3, 1, 1, 4, 256, 192, True),
    ("enc2.block 384->384 3x3 @128x96", [384], 384, 3, 1, 1, 4, 128, 96, False),
    ("conv.block 768->768 3x3 @32x24", [768], 768, 3, 1, 1, 4, 32, 24, False),
    ("seg4.scale 1x1 cat(96,96,384)->96 @512x384", [96, 96, 384], 96, 1, 1, 0, 4, 512, 384, False),
    ("enc0.scale 16->96 3x3 s2 @1024x768", [16], 96, 3, 2, 1, 4, 1024, 768, False),
    ("bott3 96->384 3x3 @512x384", [96], 384, 3, 1, 1, 4, 512, 384, False),
]


BF_LAYERS = [
    # SPADE-generator shapes of the bf16 engine
    ("spade.gb 128->2x64 3x3 @1024x768", [128], 128, 3, 1, 1, 1, 1024, 768, False),
    ("spade.gb 128->2x128 3x3 @512x384", [128], 256, 3, 1, 1, 1, 512, 384, False),
    ("blk conv 256->256 3x3 @256x192", [256], 256, 3, 1, 1, 1, 256, 192, False),
    ("blk conv 512->512 3x3 @128x96", [512], 512, 3, 1, 1, 1, 128, 96, False),
    ("blk conv 1024->1024 3x3 @64x48", [1024], 1024, 3, 1, 1, 1, 64, 48, False),
    ("blk conv 64->64 3x3 @1024x768", [64], 64, 3, 1, 1, 1, 1024, 768, False),
    ("conv_shared x3 as 1x1 over taps 72->384 @1024x768", [72], 384, 1, 1, 0, 1, 1024, 768, False),
]


def main():
    bf = os.environ.get("BF16", "0") == "1"
    mixed = os.environ.get("MIXED", "0") == "1"      # bf16 MFMA operands over fp32 tensors (training / --fp16 tocg)
    if bf or mixed:
        LAYERS[:] = BF_LAYERS
    combos = [(int(c), int(v)) for c, v in (x.split(":") for x in os.environ.get(
        "COMBOS", "1:0,1:1,1:3,2:0,2:1,2:3,0:1,7:1").split(","))]
    rounds = int(os.environ.get("ROUNDS", "5"))
    g = torch.Generator().manual_seed(0)
    sel = os.environ.get("LAYER_IDX")
    layers = LAYERS if sel is None else [LAYERS[int(i)] for i in sel.split(",")]
    for name, cins, cout, k, stride, pad, N, H, W, res in layers:
        xs = [ops.to_nhwc(torch.randn(N, c, H, W, generator=g).cuda(), bf16=bf) for c in cins]
        w = torch.randn(cout, sum(cins), k, k, generator=g) * 0.05
        sc = torch.rand(cout, generator=g) + 0.5
        sh = torch.randn(cout, generator=g)
        layer = ops.ConvLayer(w, cins, "cuda", scale=sc, shift=sh, stride=stride, pad=pad, act=ops.ACT_RELU, name=name,
                               bf16=bf, mma_bf16=mixed)
        Ho, Wo = layer.out_hw(H, W)
        out = ops.alloc(N, Ho, Wo, cout, "cuda", bf16=bf)
        r = ops.alloc(N, Ho, Wo, cout, "cuda", bf16=bf) if res else None
        if r is not None:
            r.t.normal_()
        flops = layer.flops(N, Ho, Wo)
        times = {c: [] for c in combos}
        for rd in range(rounds + 1):
            for (cfg, var) in combos:
                if ops._lib.load().hrv_conv2d_tile_bn(cfg) < 0:
                    continue
                os.environ["HRV_CONV_TILE"] = str(cfg)
                os.environ["HRV_CONV_VARIANT"] = str(var)
                ops._lib.reload_env()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                layer(xs, out=out, residual=r, cfg=cfg if mixed else None)
                e.record()
                torch.cuda.synchronize()
                if rd > 0:
                    times[(cfg, var)].append(s.elapsed_time(e))
        print(f"{name}  ({flops / 1e9:.1f} GFLOP)")
        for (cfg, var), ts in times.items():
            if ts:
                ts.sort()
                med = ts[len(ts) // 2]
                print(f"   cfg{cfg} var{var}: median {med:8.3f} ms  min {ts[0]:8.3f} ms  {flops / (med * 1e-3) / 1e12:7.1f} TFLOP/s")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
