#!/bin/bash
# SQ counters of the fused SPADENorm forward kernel (spade_fused.hip) over the SPADE layers of tools/fused_bench.py at the bench size -- two PMC
# passes (own runs, kernel-trace only), run via gpurun from the repo root.  Writes gpurun_out/pmc_fused/{sq1,sq2}.summary.txt.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_fused; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace \
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  -d $OUT/sq1 -o sq1 -- python $REPO/tools/fused_bench.py 1 > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace \
  --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES \
  -d $OUT/sq2 -o sq2 -- python $REPO/tools/fused_bench.py 1 > $OUT/sq2.log 2>&1
cd $REPO
for d in sq1 sq2; do
  f=$(ls $OUT/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/$d.summary.txt 2>&1
  rm -rf $OUT/$d
done
grep -n "spade_fused_kernel" $OUT/sq1.summary.txt | head -12
