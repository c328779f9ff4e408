#!/bin/bash
# PMC passes (own runs, kernel-trace only) -- run on the GPU box via gpurun from the repo root.
# usage: tools/profile_pmc.sh <outdir>
set -u
REPO=$(pwd)
OUT=$REPO/${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1) SQ counters on the dominant conv shape (96->96 3x3 @1024x768x4), product tile/variant
LAYER_IDX=0 COMBOS=1:1 ROUNDS=2 timeout 300 rocprofv3 --kernel-trace \
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  -d $OUT/sq1 -o sq1 -- python $REPO/tools/conv_bench.py > $OUT/sq1.log 2>&1
LAYER_IDX=0 COMBOS=1:1 ROUNDS=2 timeout 300 rocprofv3 --kernel-trace \
  --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES \
  -d $OUT/sq2 -o sq2 -- python $REPO/tools/conv_bench.py > $OUT/sq2.log 2>&1
# 2) HBM traffic of one bench step (FETCH_SIZE and WRITE_SIZE need separate passes)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
# 3) plain kernel trace + stats of the bench command
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
cd $REPO
for d in sq1 sq2 fetch write trace; do
  f=$(ls $OUT/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/$d.summary.txt 2>&1
done
ls -la $OUT
