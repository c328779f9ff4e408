#!/bin/bash
# Round-6 evidence lease (VERDICT r5 #1 a/b/c): sustained MFMA roof + clock, power/clock under the iteration, LDS/stall counters.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06_roof; mkdir -p $OUT
(cd tools/probes && hipcc --offload-arch=gfx950 -O3 -o mfma_roof mfma_roof.hip) > $OUT/build.log 2>&1
python tools/power_clock_sampler.py $OUT/power_clock_mfma_roof.txt -- tools/probes/mfma_roof 2 > $OUT/mfma_roof.txt 2>&1
python tools/power_clock_sampler.py $OUT/power_clock_bench.txt -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_steps50.log 2>&1
tail -1 $OUT/bench_steps50.log | cut -c1-400
bash tools/profile_iter_stalls.sh > $OUT/stalls.log 2>&1
cp gpurun_out/pmc_stalls/*.txt $OUT/ 2>/dev/null
grep '^#' $OUT/power_clock_bench.txt | head -20
cat $OUT/mfma_roof.txt | grep -v '^[0-9]' | head -40
