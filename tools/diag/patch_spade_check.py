"""Diagnostic (GPU): the generator's mixed-precision TRAINING forward with the patch tiles on vs off (same weights, same
noise): the two must agree to fp32 summation-order noise."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hr_viton_amd  # noqa
from hr_viton_amd import train_ops as T
from oracle import step_check

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 384)
opt, gen, D, vgg, x, seg, real, noise = step_check.build(H, W, 64, 64, 1, 0, 8.0)
gen.cuda().train()
nz = {k: [z.cuda() for z in v] for k, v in noise.items()}
outs = {}
T.MMA_BF16[0] = True
_pack = T.pack_weight_dev


def _old_pack(w, sp, sr, cfg, *a, **k):
    # the packer before the row-size fix: 32 k-values per row for tiles 16-18 (same column padding as tiles 0 / 6)
    return _pack(w, sp, sr, {16: 0, 17: 0, 18: 6}.get(cfg, cfg), *a, **k)


state0 = {k: v.detach().clone() for k, v in gen.state_dict().items()}
for env in ("1", "0", "old"):
    gen.load_state_dict(state0)          # every training forward advances the spectral-norm (u, v): same start for all
    os.environ["HRV_CONV_PATCH"] = "1" if env == "old" else env
    T.pack_weight_dev = _old_pack if env == "old" else _pack
    with torch.no_grad():
        outs[env] = gen(x.cuda(), seg.cuda(), noise=nz).float().cpu()
d = (outs["1"] - outs["0"]).abs()
print(f"generator training forward {H}x{W} mixed precision: patch tiles on vs off: max {float(d.max()):.3e} mean {float(d.mean()):.3e}")
d = (outs["old"] - outs["0"]).abs()
print(f"   with the pre-fix packer (32 k-values per row on tiles 16-18): max {float(d.max()):.3e} mean {float(d.mean()):.3e}")

from oracle import hrviton_oracle as O
sd = {k: v.detach().cpu().clone() for k, v in gen.state_dict().items()}      # (u, v) after the one power iteration
torch.set_num_threads(32)
with torch.no_grad():
    want = O.spade_generator_forward(sd, x, seg, H, W, opt.num_upsampling_layers, noise=noise)
for k, v in outs.items():
    e = (v - want).abs()
    print(f"   HRV_CONV_PATCH/packer '{k}': vs fp32 oracle max {float(e.max()):.3e} mean {float(e.mean()):.3e}")
