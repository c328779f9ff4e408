#!/bin/bash
set -u
OUT=gpurun_out/ab_d_f32_scales.txt; : > $OUT
run() { label=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$label: $r" | tee -a $OUT; }
for r in 1 2; do
run "every PatchGAN convolution in bf16 (mask 0)        " HRV_D_F32_MASK=0
run "model1 fwd, both scales (mask 2, scales 3)          " HRV_D_F32_MASK=2 HRV_D_F32_SCALES=3
run "model1 fwd, discriminator_1 only (mask 2, scales 2) " HRV_D_F32_MASK=2 HRV_D_F32_SCALES=2
run "model1+2 fwd, discriminator_1 only (mask 6, scales 2)" HRV_D_F32_MASK=6 HRV_D_F32_SCALES=2
done
