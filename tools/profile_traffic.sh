#!/bin/bash
# HBM traffic of the default (headline) bench step: FETCH_SIZE and WRITE_SIZE in separate PMC passes, plus the
# rocprofv3 --kernel-trace --stats summary of the same command -- run via gpurun from the repo root.
# usage: tools/profile_traffic.sh [workload] [extra bench flags]      (default: train_generator)
set -u
WL=${1:-train_generator}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/traffic_$WL; mkdir -p $OUT
CMD="python bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-extras $*"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o $c -- python $REPO/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/$c.log 2>&1
  f=$(ls $OUT/$c/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f > $OUT/$c.summary.txt 2>&1)
  rm -rf $OUT/$c
done
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*.db 2>/dev/null | head -1)
[ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f > $OUT/trace.summary.txt 2>&1)
rm -rf $OUT/trace
cd $REPO && python tools/traffic_json.py $OUT/FETCH_SIZE.summary.txt $OUT/WRITE_SIZE.summary.txt $OUT/pmc_traffic_$WL.json "$CMD"
head -25 $OUT/trace.summary.txt | cut -c1-160
