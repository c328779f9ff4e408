#!/bin/bash
# What the matrix pipes wait for (VERDICT r5 "do this" #1c): LDS and issue-stall SQ counters of EVERY kernel of the headline
# iteration, two PMC passes (own runs, kernel-trace only) over `python bench.py --steps 1 --warmup 1`.
#   pass A: LDS array use and conflicts            pass B: where the wave cycles go (parked / issue-stalled / issuing, by unit)
# Run via gpurun from the repo root; writes gpurun_out/pmc_stalls/{a,b}.summary.txt and stalls_per_kernel.txt.
# HRV_STALL_CMD overrides the profiled command (e.g. a micro-benchmark).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_stalls; mkdir -p $OUT
CMD=${HRV_STALL_CMD:-"python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace \
  --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
timeout 400 rocprofv3 --kernel-trace \
  --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE \
  -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
cd $REPO
for p in a b; do
  f=$(ls $OUT/$p/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/$p.summary.txt 2>&1
  rm -rf $OUT/$p
done
python - "$OUT/a.summary.txt" "$OUT/b.summary.txt" > $OUT/stalls_per_kernel.txt <<'PY'
import re, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for tag, path in (("a", sys.argv[1]), ("b", sys.argv[2])):
    try:
        lines = open(path).read().splitlines()
    except OSError:
        continue
    for l in lines:
        m = re.match(r"(\S+)\s+([\d.]+)\s+n=\s*(\d+)\s+(.*)", l)
        if not m or not m.group(1).startswith(("SQ_", "GRBM_")):
            continue
        fam = re.sub(r"<.*", "", m.group(4).replace("void ", "")).split("(")[0].strip()
        per[fam][tag + ":" + m.group(1)] += float(m.group(2))
print("# per kernel family over the profiled run (warm-up + timed + per-launch-profiled iteration).  GRBM_GUI_ACTIVE sums the 8 XCDs;")
print("# LDS counters are cycles summed over CUs; SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES are wave quad-cycles.")
print("# lds_busy   = SQ_LDS_IDX_ACTIVE / (GUI_ACTIVE / 8 * 256 CUs)         fraction of CU-cycles the LDS array is indexed")
print("# lds_confl  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE               fraction of those cycles that are conflict replays")
print("# parked     = SQ_WAIT_ANY / SQ_WAVE_CYCLES                           waves in s_waitcnt / s_barrier")
print("# issue_wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                      waves stalled at issue (MFMA dependency / pipe busy)")
print("# lds_wait   = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES                      ... of which waiting to issue an LDS instruction")
print("# issuing    = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES;  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE / 8 * 1024 SIMDs)")
print("# %-34s %8s %9s %9s %8s %10s %9s %8s %9s" % ("kernel", "share", "mfma_busy", "lds_busy", "lds_confl", "parked", "issue_wait", "lds_wait", "issuing"))
rows = []
for fam, c in per.items():
    act_a, act_b = c.get("a:GRBM_GUI_ACTIVE", 0.0), c.get("b:GRBM_GUI_ACTIVE", 0.0)
    if act_a <= 0 and act_b <= 0:
        continue
    wa, wb = c.get("a:SQ_WAVE_CYCLES", 0.0) or 1.0, c.get("b:SQ_WAVE_CYCLES", 0.0) or 1.0
    idx = c.get("a:SQ_LDS_IDX_ACTIVE", 0.0)
    rows.append((act_b or act_a, fam,
                 c.get("b:SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / ((act_b or 1.0) / 8.0 * 1024.0),
                 idx / ((act_a or 1.0) / 8.0 * 256.0),
                 c.get("a:SQ_LDS_BANK_CONFLICT", 0.0) / (idx or 1.0),
                 c.get("b:SQ_WAIT_ANY", 0.0) / wb, c.get("b:SQ_WAIT_INST_ANY", 0.0) / wb, c.get("a:SQ_WAIT_INST_LDS", 0.0) / wa,
                 c.get("b:SQ_ACTIVE_INST_ANY", 0.0) / wb))
tot = sum(r[0] for r in rows) or 1.0
for r in sorted(rows, reverse=True)[:24]:
    print("  %-34s %7.1f%% %9.3f %9.3f %9.3f %8.3f %10.3f %9.3f %8.3f" % (r[1][:34], 100.0 * r[0] / tot, r[2], r[3], r[4], r[5], r[6], r[7], r[8]))
PY
cat $OUT/stalls_per_kernel.txt
