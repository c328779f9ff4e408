#!/usr/bin/env python3
"""diag: top rows of the fp32-engine gradient table of the generator step (vs oracle autograd), optionally after a bf16 pass"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import step_check
eng = tuple(int(c) == 1 for c in (sys.argv[1] if len(sys.argv) > 1 else "0"))
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 384)
reps = step_check.compare_generator_step(H, W, 64, 64, 1, seed=0, wmul=8.0, mixed=eng, with_vgg=True,
                                         table_path="/tmp/probe.txt", cpu_threads=32)
for m in eng:
    r = reps[m]
    print("engine bf16" if m else "engine fp32", "worst %.3e %s median %.3e cos %.6f" % (r["grad_worst_rel_err"], r["grad_worst_name"], r["grad_median_rel_err"], r["grad_min_cosine"]))
    p = "/tmp/probe_bf16.txt" if m else "/tmp/probe_f32.txt"
    if len(eng) == 1:
        p = "/tmp/probe.txt"
    print("".join(open(p).readlines()[1:5]))
