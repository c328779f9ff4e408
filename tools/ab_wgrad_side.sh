#!/bin/bash
# A/B of the weight-gradient side stream (train_ops.wgrad_side) on the headline iteration -- run via gpurun from the repo root.
# Same box, alternating order, two rounds: ms per iteration for HRV_WGRAD_SIDE=0 and for several HRV_WGRAD_SIDE_MAXPIX.
set -u
OUT=gpurun_out/ab_wgrad_side.txt; mkdir -p gpurun_out; : > $OUT
run() { label=$1; shift; r=$(env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$label: $r" | tee -a $OUT; }
for round in 1 2; do
  run "off               " HRV_WGRAD_SIDE=0
  run "maxpix 65536      " HRV_WGRAD_SIDE=1
  run "maxpix 200000     " HRV_WGRAD_SIDE=1 HRV_WGRAD_SIDE_MAXPIX=200000
  run "maxpix 800000     " HRV_WGRAD_SIDE=1 HRV_WGRAD_SIDE_MAXPIX=800000
  run "all               " HRV_WGRAD_SIDE=1 HRV_WGRAD_SIDE_MAXPIX=1000000000
done
