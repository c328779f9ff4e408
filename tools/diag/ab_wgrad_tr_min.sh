#!/bin/bash
set -u
OUT=gpurun_out/ab_wgrad_tr_min.txt; : > $OUT
run() { label=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$label: $r" | tee -a $OUT; }
for r in 1 2 3; do
run "minpix 32768 (default)" HRV_WGRAD_TR_MIN_PIX=32768
run "minpix 8192           " HRV_WGRAD_TR_MIN_PIX=8192
done
