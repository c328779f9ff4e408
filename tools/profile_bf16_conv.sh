#!/bin/bash
# PMC passes on the bf16 conv micro-benchmark (SPADE gamma/beta layer, tile cfg 8) -- run via gpurun.
set -u
REPO=$(pwd)
OUT=$REPO/${1:-gpurun_out/pmc_bf16}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(TA\|TCP\|TCC\|TD\|SQ\)_[A-Z0-9_]*" | sort -u > $OUT/counters.txt
for g in 1 0; do
  HRV_CONV_GLDS=$g BF16=1 LAYER_IDX=0 COMBOS=8:1 ROUNDS=2 timeout 200 rocprofv3 --kernel-trace \
    --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
    -d $OUT/sq1_g$g -o sq1 -- python $REPO/tools/conv_bench.py > $OUT/sq1_g$g.log 2>&1
  HRV_CONV_GLDS=$g BF16=1 LAYER_IDX=0 COMBOS=8:1 ROUNDS=2 timeout 200 rocprofv3 --kernel-trace \
    --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES \
    -d $OUT/sq2_g$g -o sq2 -- python $REPO/tools/conv_bench.py > $OUT/sq2_g$g.log 2>&1
done
cd $REPO
for d in sq1_g1 sq2_g1 sq1_g0 sq2_g0; do
  f=$(ls $OUT/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/$d.summary.txt 2>&1
done
ls $OUT
