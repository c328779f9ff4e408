#!/bin/bash
# The first informative multi-GPU run (8-GPU MI355X node; NOT runnable on the 1-GPU development lease):
#   1. weak-scaling curve of the headline workload at N = 1, 2, 4, 8, for 0 / 8 / 16 CUs left free by the persistent kernels
#      (bench.py --reserve-cus: RCCL's all-reduce kernels need CUs; the persistent grids otherwise own all of them);
#   2. a rocprofv3 kernel trace of the N = 8 run at the best setting, and per rank how much of the RCCL kernels' busy time was
#      concurrent with compute kernels (tools/dp_overlap.py: exposed collective time per iteration).
# Writes gpurun_out/scale/{scale_R<k>.jsonl, best.txt, overlap_rank*.txt}.   usage: bash tools/scale_sweep.sh [steps]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/scale; mkdir -p $OUT
STEPS=${1:-10}
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
best=0; bestv=0
for R in 0 8 16; do
  : > $OUT/scale_R$R.jsonl
  for N in 1 2 4 8; do
    [ $N -le $NGPU ] || continue
    timeout 900 python bench.py --gpus $N --steps $STEPS --warmup 3 --reserve-cus $R --no-cpu-baseline --no-extras 2>>$OUT/scale_R$R.err \
      | tail -1 >> $OUT/scale_R$R.jsonl
  done
  python - "$OUT/scale_R$R.jsonl" <<'PY' | tee -a $OUT/best.txt
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = next((r["value"] for r in rows if r["n_gpus"] == 1), None)
pts = []
for r in rows:
    eff = r["value"] / (base * r["n_gpus"]) if base else float("nan")
    print(f"{sys.argv[1].split('/')[-1]}: N={r['n_gpus']} {r['value']:.2f} img/s {r['ms_per_step']:.2f} ms/step  efficiency {eff:.3f}  "
          f"backend {r['config']['dist_backend']} rccl_ranks {r['config']['rccl_ranks']} grid CUs {r['config'].get('persistent_grid_cus')}")
    pts.append({"n_gpus": r["n_gpus"], "value": r["value"], "ms_per_step": r["ms_per_step"], "scaling_efficiency": round(eff, 4),
                "speedup_vs_1": round(r["value"] / base, 3) if base else None, "dist_backend": r["config"]["dist_backend"],
                "rccl_ranks": r["config"]["rccl_ranks"]})
# the first real run judges itself: DESIGN.md 7e's prediction next to the measured weak-scaling efficiency at the largest N
if pts and pts[-1]["n_gpus"] > 1:
    pred = "0.97-0.98 (eager GradSync: ~1.5 ms of ring all-reduce exposed behind the last two 64 MiB buckets; 0.95 if RCCL's kernels cost what --reserve-cus 8 costs)"
    print(f"{sys.argv[1].split('/')[-1]}: N={pts[-1]['n_gpus']} measured efficiency {pts[-1]['scaling_efficiency']:.3f}  |  predicted at N=8: {pred}; "
          "graph-chain mode (--graph): 0.94 (402 MB all-reduced between two replays with nothing under it)")
# ONE machine-readable line per reserve setting (the shape of a SCALE record: the bench lines' own numbers per N, weak scaling)
if rows:
    print(json.dumps({"metric": rows[0]["metric"], "unit": rows[0]["unit"], "scaling": rows[0].get("scaling", "weak"),
                      "reserve_cus": int(sys.argv[1].split("_R")[-1].split(".")[0]), "points": pts}))
PY
done
R=$(python - <<PY
import json, glob
best, bv = 0, 0.0
for f in glob.glob("$OUT/scale_R*.jsonl"):
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l)
            if r["n_gpus"] == min(8, $NGPU) and r["value"] > bv:
                best, bv = int(f.split("_R")[-1].split(".")[0]), r["value"]
print(best)
PY
)
echo "best reserve setting at N=$NGPU: $R CUs" | tee -a $OUT/best.txt
cd /tmp && export TMPDIR=/tmp
N=$((NGPU < 8 ? NGPU : 8))
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o trace -- python $REPO/bench.py --gpus $N --steps 3 --warmup 2 --reserve-cus $R \
  --no-cpu-baseline --no-extras > $OUT/trace.log 2>&1
cd $REPO
i=0
for f in $(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null); do
  python tools/dp_overlap.py $f > $OUT/overlap_rank$i.txt 2>&1; tail -4 $OUT/overlap_rank$i.txt; i=$((i+1))
done
rm -rf $OUT/trace
