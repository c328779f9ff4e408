#!/bin/bash
# SQ counters of the fp32-source bf16-MFMA convolution (mixed-precision training) -- run via gpurun from the repo root.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/mixed_conv_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  MIXED=1 LAYER_IDX=0 COMBOS=8:1 ROUNDS=2 timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- python $REPO/tools/conv_bench.py > $OUT/$name.log 2>&1
  f=$(ls $OUT/$name/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f | grep -E "conv_mfma|^#" | cut -c1-150 > $OUT/$name.summary.txt 2>&1)
  rm -rf $OUT/$name
}
run a SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
cat $OUT/a.summary.txt $OUT/b.summary.txt; tail -4 $OUT/a.log
