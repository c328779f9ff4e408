#!/bin/bash
# HBM traffic of the default bench step (FETCH_SIZE and WRITE_SIZE in separate passes) -- run via gpurun.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/traffic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o $c -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1
  f=$(ls $OUT/$c/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f > $OUT/$c.summary.txt 2>&1)
  rm -rf $OUT/$c
done
grep -h "conv_mfma" $OUT/FETCH_SIZE.summary.txt $OUT/WRITE_SIZE.summary.txt | grep -E "FETCH_SIZE|WRITE_SIZE" | cut -c1-140
