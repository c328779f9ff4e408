#!/bin/bash
set -u
OUT=gpurun_out/ab_gb_min.txt; : > $OUT
run() { label=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$label: $r" | tee -a $OUT; }
for r in 1 2 3; do
run "spade_gb from 256 tiles (default)" HRV_SPADE_GB_MIN_TILES=256
run "spade_gb from 192 tiles          " HRV_SPADE_GB_MIN_TILES=192
done
