#!/bin/bash
set -u
OUT=gpurun_out/ab_q4.txt; : > $OUT
run() { label=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$label: $r" | tee -a $OUT; }
for r in 1 2 3; do
run "q4=6 (default)" HRV_CONV_P2_MIN_TILES_X4=6
run "q4=3          " HRV_CONV_P2_MIN_TILES_X4=3
run "q4=2          " HRV_CONV_P2_MIN_TILES_X4=2
done
