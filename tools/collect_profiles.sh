#!/bin/bash
# Copy what tools/round_end_measure.sh wrote under gpurun_out/final/ into profiles/ under this round's names (run in the build container
# after the gpurun call returned).   usage: bash tools/collect_profiles.sh r05
set -u
R=${1:-r05}; F=gpurun_out/final; P=profiles
c() { [ -s "$F/$1" ] && cp "$F/$1" "$P/${R}_$2" && echo "$P/${R}_$2"; }
c bench_default.json final_bench_default.json
c bench_default_detail.json final_bench_detail.json
c launches_default.txt final_launches_train_generator_bf16.txt
c launch_summary_default.txt final_launch_summary_train_generator_bf16.txt
c launches_per_iteration.txt final_launches_per_iteration.txt
c trace.summary.txt final_trace_train_generator.summary.txt
c FETCH_SIZE.summary.txt final_pmc_fetch_train_generator.summary.txt
c WRITE_SIZE.summary.txt final_pmc_write_train_generator.summary.txt
c pmc_traffic_train_generator.json final_pmc_traffic_train_generator.json
c pmc_traffic_train_generator.json pmc_traffic_train_generator.json
for n in train_generator_bf16_graph train_generator_f32 train_condition_f32 train_condition_bf16 tryon_infer_bf16 tocg_infer_bf16; do
  c $n.json final_bench_$n.json
done
c launches_tryon_infer_bf16.txt final_launches_tryon_infer_bf16.txt
c dp_overlap.txt final_dp_overlap.txt
c reserve_cus.txt final_reserve_cus.txt
c mfma_busy_per_kernel.txt final_mfma_busy_per_kernel.txt
c pmc_iter_sq1.summary.txt final_pmc_iter_sq1.summary.txt
c d_f32_layers.txt final_d_f32_layers.txt
c stalls_per_kernel.txt final_stalls_per_kernel.txt
c power_clock_bench.txt final_power_clock.txt
c p2_timeline.txt final_p2_timeline.txt
c p2_bench.txt final_p2_bench.txt
c p2_bench_coarse.txt final_p2_bench_coarse.txt
c s2_bench.txt final_s2_bench.txt
c bench_2rank_gloo_one_gpu_smoke.json final_bench_2rank_gloo_one_gpu_smoke.json
c bench_2rank_gloo_graph_one_gpu_smoke.json final_bench_2rank_gloo_graph_one_gpu_smoke.json
[ -s gpurun_out/pytest_gpu.txt ] && tail -5 gpurun_out/pytest_gpu.txt > $P/${R}_final_pytest_gpu.txt
ls -la $P | grep "${R}_final" | wc -l
