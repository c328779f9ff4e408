"""Diagnostic (GPU): conv_forward_dev / conv_dgrad over a bf16-stored source with the patch tiles on vs off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hr_viton_amd  # noqa
from hr_viton_amd import ops, train_ops as T

T.MMA_BF16[0] = True
torch.manual_seed(0)
for (N, H, W, Cin, Cout) in ((4, 128, 96, 128, 256), (2, 128, 96, 128, 192), (4, 256, 192, 256, 128), (4, 128, 96, 128, 64)):
    x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    a = ops.Act(x, Cin)
    outs = {}
    for env in ("1", "0"):
        os.environ["HRV_CONV_PATCH"] = env
        cfgp = ops.patch_tile(True, 3, 3, 1, 1, 1, 0, Cin, Cout, N, H, W)
        o = T.conv_forward_dev(w, [(a, 0)], 1, 1, name="chk")
        outs[env] = (o.t[..., :Cout].float().clone(), cfgp)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1)
    for env, (o, c) in outs.items():
        print(f"N{N} {H}x{W} {Cin}->{Cout} HRV_CONV_PATCH={env} cfg {c}: max err vs torch {float((o - ref).abs().max()):.3e} (|ref| max {float(ref.abs().max()):.2f})")

# which side is wrong: the device packer's layout or the kernel?  pack with the 128-byte-row gather tile (cfg 8/9) and run
# the patch tile on it
print("--- pack with cfg 8 (64 k-values per row), run tile 17")
os.environ["HRV_CONV_PATCH"] = "1"
N, H, W, Cin, Cout = 4, 128, 96, 128, 256
x = torch.randn(N, H, W, Cin, device="cuda").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
a = ops.Act(x, Cin)
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1)
for pack_cfg in (8, 17):
    packed, _ = T.pack_weight_dev(w, [Cin], [Cin], pack_cfg, 0, 1, 1, bf16=True)
    out = ops.alloc(N, H, W, Cout, x.device)
    T._run_engine([(a, 0, Cin)], packed, Cout, 17, N, H, W, H, W, 3, 3, 1, 1, 1, out, name="chk", mma_bf16=True)
    print(f"pack cfg {pack_cfg} -> engine 17: max err {float((out.t[..., :Cout] - ref).abs().max()):.3e}")
