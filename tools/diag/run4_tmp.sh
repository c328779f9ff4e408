timeout 300 python tools/d_f32_layers.py > gpurun_out/d_f32_layers.txt 2>gpurun_out/d_f32_layers.err; cat gpurun_out/d_f32_layers.txt
b() { r=$(env "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"); echo "$*: $r" | tee -a gpurun_out/d_f32_cost.txt; }
: > gpurun_out/d_f32_cost.txt
b HRV_D_F32_MASK=0
b HRV_D_F32_MASK=2 HRV_D_F32_PARTS=all
b HRV_D_F32_MASK=2 HRV_D_F32_PARTS=fwd
b HRV_D_F32_MASK=6 HRV_D_F32_PARTS=fwd
b HRV_D_F32_MASK=6 HRV_D_F32_PARTS=all
b HRV_D_F32_MASK=0
bash tools/profile_iter_sq.sh > gpurun_out/profile_iter_sq.log 2>&1; tail -16 gpurun_out/profile_iter_sq.log
