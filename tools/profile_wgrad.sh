#!/bin/bash
# SQ counters of the weight-gradient kernels -- run via gpurun from the repo root.
# usage: tools/profile_wgrad.sh [MIXED=1|0]   (1: bf16 matrix cores, 0: the fp32 kernel)
set -u
MIX=${1:-1}
REPO=$(pwd); OUT=$REPO/gpurun_out/wgrad_pmc_$MIX; mkdir -p $OUT
export MIXED=$MIX XBF=$MIX YBF=$MIX
python tools/wgrad_bench.py | tail -1
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- python $REPO/tools/wgrad_bench.py > $OUT/$name.log 2>&1
  f=$(ls $OUT/$name/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f | grep -E "conv_wgrad|^#" | cut -c1-150 > $OUT/$name.summary.txt)
  rm -rf $OUT/$name
}
run a SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES
cat $OUT/a.summary.txt $OUT/b.summary.txt
