#!/bin/bash
# SQ counters of EVERY kernel of the headline iteration (not a micro-benchmark): one PMC pass (own run, kernel-trace only) over
# `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras` -- matrix-pipe busy cycles per kernel name next to the kernel's
# wave cycles, so the dominant kernel's MFMA utilisation is the one it has INSIDE the iteration.  Run via gpurun from the repo root;
# writes gpurun_out/pmc_iter/sq1.summary.txt and mfma_busy_per_kernel.txt.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_iter; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace \
  --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  -d $OUT/sq1 -o sq1 -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/sq1.log 2>&1
cd $REPO
f=$(ls $OUT/sq1/*.db 2>/dev/null | head -1)
[ -n "$f" ] && python tools/rocprof_summary.py $f > $OUT/sq1.summary.txt 2>&1
rm -rf $OUT/sq1
python - "$OUT/sq1.summary.txt" > $OUT/mfma_busy_per_kernel.txt <<'PY'
import re, sys, collections
# per kernel FAMILY (template instances summed): matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
# (MI355X_MICROARCH.md: the MFMA-busy counter sums over SIMDs, GRBM_GUI_ACTIVE over XCDs)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for l in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+([\d.]+)\s+n=\s*(\d+)\s+(.*)", l)
    if not m or not m.group(1).startswith(("SQ_", "GRBM_")):
        continue
    name = m.group(4)
    fam = re.sub(r"<.*", "", name.replace("void ", "")).split("(")[0].strip()
    per[fam][m.group(1)] += float(m.group(2))
    if m.group(1) == "GRBM_GUI_ACTIVE":
        per[fam]["n"] += int(m.group(3))
rows = []
for fam, c in per.items():
    act = c.get("GRBM_GUI_ACTIVE", 0.0)
    if act <= 0:
        continue
    rows.append((act, fam, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act / 8.0 * 1024.0), c.get("SQ_INSTS_MFMA", 0.0), int(c["n"])))
print("# matrix-pipe busy fraction per kernel family over the profiled run (3 iterations: warm-up + timed + per-launch-profiled)")
print("# %-44s %10s %12s %16s %8s" % ("kernel", "dispatches", "mfma_busy", "wave MFMA insts", "share"))
tot = sum(r[0] for r in rows) or 1.0
for act, fam, busy, insts, n in sorted(rows, reverse=True)[:25]:
    print("  %-44s %10d %12.3f %16.0f %7.1f%%" % (fam[:44], n, busy, insts, 100.0 * act / tot))
PY
head -14 $OUT/mfma_busy_per_kernel.txt
