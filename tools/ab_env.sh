#!/bin/bash
# Same-box alternating A/B of one environment switch on the headline iteration.
#     bash tools/ab_env.sh HRV_SWITCH valueA valueB [rounds]        (via gpurun)
set -u
V=$1; A=$2; B=$3; R=${4:-3}
for r in $(seq 1 $R); do
  for x in "$A" "$B"; do
    ms=$(env $V=$x python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
    echo "round $r $V=$x: $ms ms/step"
  done
done
