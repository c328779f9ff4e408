#!/bin/bash
# Round-end measurement sweep -- run on the GPU box via gpurun from the repo root.
# Writes gpurun_out/final/: the HBM-traffic PMC passes and the rocprofv3 kernel-trace summary of the headline iteration (first:
# bench.py's roofline.traffic reads the per-kernel bytes of THIS build), the default bench line (headline config, with
# cpu_baseline / parity / extras), one line per secondary workload, per-launch HIP-event tables, the per-iteration launch
# table (two differenced traces), the 1-GPU collective-overlap numbers.  Copy what should be judged into profiles/.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
bash tools/profile_traffic.sh train_generator > $OUT/profile_traffic.log 2>&1
cp gpurun_out/traffic_train_generator/*.summary.txt gpurun_out/traffic_train_generator/*.json $OUT/ 2>/dev/null
# (on the box only: the default run below prices its traffic from the passes just taken; the copy under profiles/ that is
#  committed afterwards is this same file)
cp gpurun_out/traffic_train_generator/pmc_traffic_train_generator.json profiles/r06_pmc_traffic_train_generator.json 2>/dev/null
timeout 1200 python bench.py --dump-launches $OUT/launches_default.txt 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
cp gpurun_out/bench_detail.json $OUT/bench_default_detail.json 2>/dev/null      # (the full result behind the compact line)
wc -c $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
b() { name=$1; shift; timeout 400 python bench.py "$@" --no-cpu-baseline --no-extras --dump-launches $OUT/launches_$name.txt 2>$OUT/$name.err | tail -1 > $OUT/$name.json; cut -c1-200 $OUT/$name.json; }
b train_generator_bf16_graph --workload train_generator --graph --steps 8 --warmup 3
b train_generator_f32 --workload train_generator --fp32 --steps 3 --warmup 2
b train_condition_f32 --workload train_condition --steps 3 --warmup 1
b train_condition_bf16 --workload train_condition --bf16 --steps 3 --warmup 2
b tryon_infer_bf16 --workload tryon_infer --bf16 --steps 10 --warmup 3
b tocg_infer_bf16 --workload tocg_infer --bf16 --steps 10 --warmup 3
python tools/launch_summary.py $OUT/launches_default.txt > $OUT/launch_summary_default.txt 2>&1
bash tools/launch_count.sh > $OUT/launch_count.log 2>&1
cp gpurun_out/launch_count/per_step.txt $OUT/launches_per_iteration.txt 2>/dev/null
timeout 500 bash tools/dp_overlap.sh > $OUT/dp_overlap.log 2>&1
cp gpurun_out/dp_overlap/overlap.txt $OUT/dp_overlap.txt 2>/dev/null
# round 5: SQ counters of every kernel of the iteration itself (matrix-pipe busy of the dominant kernel INSIDE the iteration), the
# PatchGAN layer-by-layer precision sweep.  (tools/fused_bench.py / p2_bench.py / profile_fused.sh: the round-4 kernels are unchanged,
# their numbers are profiles/r04_final_*; HRV_FULL_SWEEP=1 re-runs them.)
timeout 400 bash tools/profile_iter_sq.sh > $OUT/profile_iter_sq.log 2>&1
cp gpurun_out/pmc_iter/mfma_busy_per_kernel.txt $OUT/mfma_busy_per_kernel.txt 2>/dev/null
cp gpurun_out/pmc_iter/sq1.summary.txt $OUT/pmc_iter_sq1.summary.txt 2>/dev/null
timeout 300 python tools/d_f32_layers.py 0:all 2:fwd:2 7:fwd:2 15:all > $OUT/d_f32_layers.txt 2>$OUT/d_f32_layers.err
# round 6: what the matrix pipes wait for (LDS / issue-stall counters per kernel inside the iteration), power + clock under the
# iteration, the conv_p2 phase timeline and micro-benchmark at HEAD
timeout 500 bash tools/profile_iter_stalls.sh > $OUT/profile_iter_stalls.log 2>&1
cp gpurun_out/pmc_stalls/stalls_per_kernel.txt $OUT/stalls_per_kernel.txt 2>/dev/null
timeout 200 python tools/power_clock_sampler.py $OUT/power_clock_bench.txt -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/power_clock_bench.log 2>&1
timeout 200 python tools/p2_timeline.py > $OUT/p2_timeline.txt 2>&1
timeout 300 python tools/p2_bench.py 5 > $OUT/p2_bench.txt 2>&1
timeout 200 python tools/p2_bench.py 5 coarse > $OUT/p2_bench_coarse.txt 2>&1
timeout 200 python tools/s2_bench.py 5 > $OUT/s2_bench.txt 2>&1
if [ "${HRV_FULL_SWEEP:-0}" = "1" ]; then
timeout 300 python tools/fused_bench.py 5 > $OUT/fused_bench.txt 2>&1
timeout 300 python tools/p2_bench.py 5 > $OUT/p2_bench.txt 2>&1
timeout 400 bash tools/profile_fused.sh > $OUT/profile_fused.log 2>&1
cp gpurun_out/pmc_fused/sq1.summary.txt $OUT/pmc_fused_sq1.summary.txt 2>/dev/null
cp gpurun_out/pmc_fused/sq2.summary.txt $OUT/pmc_fused_sq2.summary.txt 2>/dev/null
fi
for R in 0 8 16; do timeout 300 python bench.py --reserve-cus $R --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('reserve-cus $R:', d['ms_per_step'], 'ms/step', d['value'], 'img/s, persistent grid', d['config']['persistent_grid_cus'], 'CUs')"; done > $OUT/reserve_cus.txt 2>&1
# two ranks time-slicing the one GPU over gloo (functional smoke of the N > 1 launch, never a measurement): eager bucketed
# GradSync, and the captured iteration as three hipGraph segments with the all-reduces between them
HRV_DIST_BACKEND=gloo timeout 500 python bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-extras 2>$OUT/bench_2rank_gloo.err | tail -1 > $OUT/bench_2rank_gloo_one_gpu_smoke.json
# (GPU_MAX_HW_QUEUES=1: two PROCESSES replaying graphs on one device oversubscribe its hardware queues with the default of 4 per process
#  and are time-sliced at every cross-queue dependency -- 17-32 s per iteration; one queue each: the eager figure.  One rank per GPU
#  never meets this.)
GPU_MAX_HW_QUEUES=1 HRV_DIST_BACKEND=gloo timeout 500 python bench.py --gpus 2 --graph --steps 3 --warmup 2 --no-cpu-baseline --no-extras 2>$OUT/bench_2rank_gloo_graph.err | tail -1 > $OUT/bench_2rank_gloo_graph_one_gpu_smoke.json
cut -c1-400 $OUT/bench_2rank_gloo_one_gpu_smoke.json $OUT/bench_2rank_gloo_graph_one_gpu_smoke.json
ls -la $OUT | head -40
