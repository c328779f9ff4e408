#!/bin/bash
# Same-box alternating A/B of two builds of libhrviton_hip.so on the headline iteration: A = tools/_tmp_base/libhrviton_hip.so (a
# baseline build: e.g. the previous commit's sources), B = the in-tree build.  ROUNDS alternations (default 3), bench.py --steps 12.
#     bash tools/ab_lib.sh [rounds] [extra bench.py args]         (via gpurun)
set -u
R=${1:-3}; shift || true
LIB=hr-viton_amd/libhrviton_hip.so
cp $LIB /tmp/lib_new.so
for r in $(seq 1 $R); do
  for v in base new; do
    if [ $v = base ]; then cp tools/_tmp_base/libhrviton_hip.so $LIB; else cp /tmp/lib_new.so $LIB; fi
    ms=$(python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
    echo "round $r $v: $ms ms/step"
  done
done
cp /tmp/lib_new.so $LIB
