#!/usr/bin/env python3
"""What bf16 operand rounding in the PatchGAN's first layers costs the discriminator's gradient: the D half of
one train_generator.py iteration at 2 x 1024x768 (oracle/step_check.compare_discriminator_step: torch autograd over the fp32
oracle) for HRV_D_F32_LAYERS = 0, 1, 2, 4 -- run via gpurun from the repo root.  TEST TOOL (imports the oracle)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch  # noqa: F401
    import hr_viton_amd  # noqa: F401
    from oracle import step_check
    ks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4]
    out = {}
    for k in ks:
        os.environ["HRV_D_F32_LAYERS"] = str(k)
        r = step_check.compare_discriminator_step(1024, 768, 64, 64, 2, seed=1, mixed=(True,), cpu_threads=min(os.cpu_count() or 1, 32))[True]
        out[k] = {x: r[x] for x in ("loss_rel_err", "grad_worst_rel_err", "grad_worst_name", "grad_median_rel_err", "grad_min_cosine",
                                    "post_step_weight_frac_off_by_more_than_lr_tenth", "bf16_rounded_oracle_vs_fp32_oracle") if x in r}
        print(k, json.dumps(out[k]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "d_f32_layers.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
