#!/bin/bash
# SQ counters of the bf16 weight-gradient kernel -- run via gpurun from the repo root.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/wgrad_pmc; mkdir -p $OUT
python tools/wgrad_bench.py | tail -1
XBF=0 YBF=0 python tools/wgrad_bench.py | tail -1
cd /tmp && export TMPDIR=/tmp
ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/a -o a -- python $REPO/tools/wgrad_bench.py > $OUT/a.log 2>&1
ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES -d $OUT/b -o b -- python $REPO/tools/wgrad_bench.py > $OUT/b.log 2>&1
ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr -d $OUT/c -o c -- python $REPO/tools/wgrad_bench.py > $OUT/c.log 2>&1
cd $REPO
for d in a b c; do f=$(ls $OUT/$d/*.db 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocprof_summary.py $f | grep -E "wgrad_bf16|^#" | cut -c1-150 > $OUT/$d.summary.txt; rm -rf $OUT/$d; done
cat $OUT/*.summary.txt; tail -3 $OUT/c.log
