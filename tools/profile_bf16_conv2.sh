#!/bin/bash
# Memory-side PMC passes on the bf16 conv micro-benchmark (SPADE gamma/beta layer, tile cfg 8, LDS-DMA).
set -u
REPO=$(pwd)
OUT=$REPO/${1:-gpurun_out/pmc_bf16b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  BF16=1 LAYER_IDX=0 COMBOS=8:1 ROUNDS=2 timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- python $REPO/tools/conv_bench.py > $OUT/$name.log 2>&1
  f=$(ls $OUT/$name/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && (cd $REPO && python tools/rocprof_summary.py $f > $OUT/$name.summary.txt 2>&1)
  rm -rf $OUT/$name
}
run tcc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_REQ
run tcp TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES
run ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES
ls $OUT
