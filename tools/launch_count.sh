#!/bin/bash
# Kernel launches and kernel time per training iteration: two rocprofv3 kernel traces of the headline bench command at 2 and
# 6 timed steps, differenced (tools/rocprof_per_step.py).  Run via gpurun from the repo root; writes gpurun_out/launch_count/.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/launch_count; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in 2 6; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/t$s -o t$s -- python $REPO/bench.py --steps $s --warmup 2 --no-cpu-baseline --no-extras "$@" > $OUT/t$s.log 2>&1
  f=$(ls $OUT/t$s/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f > $OUT/trace_steps$s.summary.txt 2>&1)
  rm -rf $OUT/t$s
done
cd $REPO && python tools/rocprof_per_step.py $OUT/trace_steps2.summary.txt 2 $OUT/trace_steps6.summary.txt 6 > $OUT/per_step.txt
head -70 $OUT/per_step.txt | cut -c1-150
