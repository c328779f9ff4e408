"""Diagnostic (GPU): ONE training-mode SPADE norm (gen_train.SpadeT: fused gamma|beta conv + modulate) on the patch
tiles vs the gather tiles vs a torch reference on the same bf16-rounded operands."""
import os, sys
from argparse import Namespace
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hr_viton_amd  # noqa
from hr_viton_amd import ops, train_ops as T
from hr_viton_amd.gen_train import SpadeT
from hr_viton_amd.network_generator import SPADENorm
from hr_viton_amd.ops import ACT_LRELU

T.MMA_BF16[0] = True
torch.manual_seed(0)
for (N, H, W, Cc) in ((2, 256, 192, 64), (4, 256, 192, 32), (2, 256, 192, 96), (1, 512, 384, 32)):
    norm = SPADENorm(Namespace(), "aliasinstance", Cc, 7).cuda()
    with torch.no_grad():
        for p in norm.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    x = torch.randn(N, H, W, Cc, device="cuda")
    actv_all = torch.relu(torch.randn(N, H, W, 384, device="cuda")).to(torch.bfloat16)
    actv = actv_all[..., 128:256]
    rb = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    a_nchw = actv.float().permute(0, 3, 1, 2)
    gamma = F.conv2d(a_nchw, rb(norm.conv_gamma.weight), norm.conv_gamma.bias, padding=1)
    beta = F.conv2d(a_nchw, rb(norm.conv_beta.weight), norm.conv_beta.bias, padding=1)
    z = torch.randn(N, W, H, 1, device="cuda")
    xz = x.permute(0, 3, 1, 2) + (z * norm.noise_scale).transpose(1, 3)          # network_generator.py:93-99
    xn = F.instance_norm(xz, eps=1e-5)
    ref = F.leaky_relu(xn * (1 + gamma) + beta, 0.2).permute(0, 2, 3, 1)
    st = SpadeT(norm, ACT_LRELU, "chk")
    _pack = T.pack_weight_dev
    for env in ("1", "0"):
        os.environ["HRV_CONV_PATCH"] = "1" if env == "old" else env
        T.pack_weight_dev = (lambda w, sp, sr, c, *a, **k: _pack(w, sp, sr, {16: 0, 17: 0, 18: 6}.get(c, c), *a, **k)) if env == "old" else _pack
        cfg = ops.patch_tile(True, 3, 3, 1, 1, 1, 0, 128, st.G * 64, N, H, W, wide=True)
        out, _ = st.forward(ops.Act(x, Cc), ops.Act(actv_all, 128, 128), z, save=False)
        e = (out.t[..., :Cc].float() - ref).abs()
        print(f"N{N} {H}x{W} C={Cc} HRV_CONV_PATCH={env} tile {cfg}: max err {float(e.max()):.3e} mean {float(e.mean()):.3e} "
              f"(|ref| max {float(ref.abs().max()):.2f}, out bf16 {out.bf16})")
