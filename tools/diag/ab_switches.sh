#!/bin/bash
# same-box A/B of existing switches on the headline iteration (ms per iteration, images/s): run via gpurun from the repo root
set -u
OUT=gpurun_out/ab_switches.txt; : > $OUT
run() { label=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$label: $r" | tee -a $OUT; }
run "default                      " HRV_X=0
run "HRV_CONV_P2_MIN_TILES_X4=4    " HRV_CONV_P2_MIN_TILES_X4=4
run "HRV_CONV_P2_MIN_TILES_X4=3    " HRV_CONV_P2_MIN_TILES_X4=3
run "HRV_NORM_BWD2=1               " HRV_NORM_BWD2=1
run "HRV_NORM_SLABS_MAX=128        " HRV_NORM_SLABS_MAX=128
run "HRV_NORM_SLABS_MAX=512        " HRV_NORM_SLABS_MAX=512
run "HRV_VGG_BATCH=0               " HRV_VGG_BATCH=0
run "default                      " HRV_X=0
