#!/bin/bash
# Round-end measurement sweep -- run on the GPU box via gpurun from the repo root.
# Writes gpurun_out/final/: one JSON line per workload, per-launch HIP-event tables, rocprofv3 kernel-trace summaries.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
b() { name=$1; shift; timeout 400 python bench.py "$@" --dump-launches $OUT/launches_$name.txt 2>$OUT/$name.err | tail -1 > $OUT/$name.json; cut -c1-260 $OUT/$name.json; }
b tocg_infer_f32 --steps 10 --warmup 3
b tocg_infer_bf16 --steps 10 --warmup 3 --bf16 --no-cpu-baseline
b tryon_infer_f32 --workload tryon_infer --steps 5 --warmup 2 --no-cpu-baseline
b tryon_infer_bf16 --workload tryon_infer --bf16 --steps 10 --warmup 3 --no-cpu-baseline
b tryon_infer_bf16_b16 --workload tryon_infer --bf16 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline
b train_generator_f32 --workload train_generator --steps 3 --warmup 2 --no-cpu-baseline
b train_generator_bf16 --workload train_generator --bf16 --steps 5 --warmup 2 --no-cpu-baseline
b train_condition_f32 --workload train_condition --steps 3 --warmup 1 --no-cpu-baseline
b train_condition_bf16 --workload train_condition --bf16 --steps 3 --warmup 2 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
for spec in "tocg_infer:" "train_generator:--bf16" "train_condition:--bf16"; do
  wl=${spec%%:*}; fl=${spec#*:}
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace_$wl -o $wl -- python $REPO/bench.py --workload $wl $fl --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace_$wl.log 2>&1
  f=$(ls $OUT/trace_$wl/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $REPO/tools/rocprof_summary.py $f > $OUT/trace_$wl.summary.txt 2>&1
  rm -rf $OUT/trace_$wl
done
ls -la $OUT | head -40
