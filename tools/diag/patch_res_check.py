import sys, os
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
import hr_viton_amd
from hr_viton_amd import ops
g = torch.Generator().manual_seed(1)
rb = lambda t: t.to(torch.bfloat16).float()
for (cin, cout, H, W, N) in ((128, 256, 20, 24, 2), (128, 128, 32, 48, 2), (128, 256, 64, 64, 2)):
    x = rb(torch.randn(N, cin, H, W, generator=g)); w = rb(torch.randn(cout, cin, 3, 3, generator=g) * 0.05)
    b = torch.randn(cout, generator=g); res = rb(torch.randn(N, cout, H, W, generator=g))
    ref = F.relu(F.conv2d(x, w, b, padding=1) + res)
    layer = ops.ConvLayer(w, [cin], "cuda", shift=b, stride=1, pad=1, act=ops.ACT_RELU, name="t", bf16=True)
    xa = ops.to_nhwc(x.cuda(), bf16=True); ra = ops.to_nhwc(res.cuda(), bf16=True)
    for f32out in (True, False):
        for cfg in (8, 17, 18, 16):
            out = ops.alloc(N, H, W, cout, "cuda", bf16=not f32out)
            o = layer([(xa, 0, ops.ACT_NONE)], residual=ra, cfg=cfg, out=out)
            e = (ops.to_nchw(o).float().cpu() - ref).abs().max().item()
            print(f"{cin}->{cout} {H}x{W} f32out={f32out} cfg{cfg}: max err {e:.3e}")
