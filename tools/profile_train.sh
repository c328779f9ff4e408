#!/bin/bash
# kernel trace of one training workload -- run via gpurun from the repo root.
# usage: tools/profile_train.sh <workload> <outname> [extra bench flags]
set -u
REPO=$(pwd)
WL=${1:-train_generator}; NAME=${2:-tg}; shift 2
OUT=$REPO/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $NAME -- python $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
cd $REPO
f=$(ls $OUT/trace/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/summary.txt 2>&1
tail -1 $OUT/bench.log | cut -c1-400
head -50 $OUT/summary.txt
