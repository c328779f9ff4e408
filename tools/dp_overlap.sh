#!/bin/bash
# 1-GPU overlap evidence for the bucketed gradient all-reduce: the headline step with each bucket's collective
# replaced by R device copies on a side stream (R = 18 ~ a 64 MiB bucket at ring speed over one xGMI link pair,
# 0.7 ms), timed in-process at R = 0 / 18 / 72 and traced at R = 18.  usage: tools/dp_overlap.sh   (via gpurun)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/dp_overlap; mkdir -p $OUT
for R in 0 18 72; do
  HRV_FAKE_ALLREDUCE=$R timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('HRV_FAKE_ALLREDUCE=$R', d['ms_per_step'], 'ms/step', d['value'], d['unit'])" \
    | tee -a $OUT/overlap.txt
done
cd /tmp && export TMPDIR=/tmp
HRV_FAKE_ALLREDUCE=18 timeout 400 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*.db 2>/dev/null | head -1)
[ -n "$f" ] && (cd $REPO && python tools/dp_overlap.py $f | tee -a $OUT/overlap.txt)
rm -rf $OUT/trace
