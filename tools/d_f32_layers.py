#!/usr/bin/env python3
"""What bf16 operand rounding in the PatchGAN costs the discriminator's gradient, layer by layer: the D half of
one train_generator.py iteration at 2 x 1024x768 (oracle/step_check.compare_discriminator_step: torch autograd over the fp32
oracle, computed ONCE) against the bf16 engine with chosen PatchGAN convolutions kept on fp32 operands (HRV_D_F32_MASK: bit i =
layer i of every scale; HRV_D_F32_PARTS: all | fwd | bwd) -- run via gpurun from the repo root.  TEST TOOL (imports the oracle).
    python tools/d_f32_layers.py [mask:parts ...]        default: a sweep"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import hr_viton_amd  # noqa: F401
    from oracle import step_check
    args = [a for a in sys.argv[1:] if "=" not in a]
    kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)          # seed=2 wmul=4: another draw / weight scale
    seed, wmul = int(kv.get("seed", "1")), float(kv.get("wmul", "8"))
    specs = args or ["0:all", "2:all", "2:fwd", "2:bwd", "6:fwd", "6:all", "4:all", "3:all", "14:all", "15:all"]
    engines = []
    for sp in specs:
        m, parts = sp.split(":")[:2]
        scales = sp.split(":")[2] if sp.count(":") > 1 else "255"      # mask:parts[:scale bitmask]
        engines.append((True, {"HRV_D_F32_MASK": m, "HRV_D_F32_PARTS": parts, "HRV_D_F32_LAYERS": "0", "HRV_D_F32_SCALES": scales}, sp))
    rep = step_check.compare_discriminator_step(1024, 768, 64, 64, 2, seed=seed, wmul=wmul, mixed=tuple(engines),
                                                 cpu_threads=min(os.cpu_count() or 1, 32))
    print("seed %d wmul %g" % (seed, wmul))
    out = {}
    for sp in specs:
        r = rep[sp]
        out[sp] = {x: r[x] for x in ("loss_rel_err", "grad_worst_rel_err", "grad_worst_name", "grad_median_rel_err", "grad_min_cosine",
                                     "post_step_weight_frac_off_by_more_than_lr_tenth", "bf16_rounded_oracle_vs_fp32_oracle") if x in r}
        print("mask:parts %-8s cosine %.4f  worst %.3f (%s)  median %.4f  weights off > lr/10: %.4f" %
              (sp, r["grad_min_cosine"], r["grad_worst_rel_err"], r["grad_worst_name"], r["grad_median_rel_err"],
               r["post_step_weight_frac_off_by_more_than_lr_tenth"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "d_f32_layers.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
