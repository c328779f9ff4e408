#!/usr/bin/env python3
"""Per-iteration kernel table from TWO rocprofv3 --kernel-trace summaries (tools/rocprof_summary.py) of the same bench
command at different --steps: calls and time per iteration = (B - A) / (steps_B - steps_A) -- set-up work (weight upload,
optimizer set-up, the first iteration's one-by-one weight packs) cancels.
    python tools/rocprof_per_step.py A.summary.txt stepsA B.summary.txt stepsB > per_step.txt"""
import re
import sys


def load(path):
    out = {}
    for l in open(path):
        m = re.match(r'\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)$', l)
        if m:
            out[m.group(5).strip()] = (int(m.group(1)), float(m.group(2)))
    return out


def main():
    a, sa, b, sb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    d = sb - sa
    rows = []
    for k, (cb, tb) in b.items():
        ca, ta = a.get(k, (0, 0.0))
        rows.append(((cb - ca) / d, (tb - ta) / d, k))
    rows.sort(key=lambda r: -r[1])
    tc, tt = sum(r[0] for r in rows), sum(r[1] for r in rows)
    print(f"# per training iteration: {tc:.1f} kernel launches, {tt:.3f} ms of kernel time  (difference of two traces, {sa} and {sb} timed steps)")
    print("#  calls/iter   ms/iter   kernel")
    for c, t, k in rows:
        if c > 0.01 or t > 0.001:
            print(f"{c:11.1f} {t:9.3f}   {k[:150]}")


if __name__ == "__main__":
    main()
