#!/bin/bash
set -u
OUT=gpurun_out/ab_fused_min.txt; : > $OUT
run() { label=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$label: $r" | tee -a $OUT; }
for r in 1 2; do
run "fused from 2 tiles/CU (default)" HRV_SPADE_FUSED_MIN_TILES_X4=8
run "fused from 0.75 tiles/CU       " HRV_SPADE_FUSED_MIN_TILES_X4=3
done
