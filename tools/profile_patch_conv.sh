#!/bin/bash
# SQ counters of the patch-mode bf16 convolution (tile_cfg 17) next to the gather tile (cfg 8) -- run via gpurun.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/patch_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; cfg=$2; shift 2
  BF16=1 LAYER_IDX=0 COMBOS=$cfg:1 ROUNDS=2 timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- python $REPO/tools/conv_bench.py > $OUT/$name.log 2>&1
  f=$(ls $OUT/$name/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f | grep -E "conv_mfma" | cut -c1-150 > $OUT/$name.summary.txt 2>&1)
  rm -rf $OUT/$name
}
for cfg in 8 17; do
  run a$cfg $cfg SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
  run b$cfg $cfg SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
done
for cfg in 8 17; do echo "== cfg $cfg"; cat $OUT/a$cfg.summary.txt $OUT/b$cfg.summary.txt; done
