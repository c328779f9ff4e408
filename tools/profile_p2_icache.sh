#!/bin/bash
# Instruction-cache counters of conv_p2 on its micro-benchmark shapes (does the once-per-tile epilogue code miss?)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_p2_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  -d $OUT/a -o a -- python $REPO/tools/p2_timeline.py ${1:-1} ${2:-3} > $OUT/a.log 2>&1
cd $REPO
f=$(ls $OUT/a/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/a.summary.txt 2>&1
rm -rf $OUT/a
grep -E "conv_p2_kernel" $OUT/a.summary.txt | head -30
tail -5 $OUT/a.log
