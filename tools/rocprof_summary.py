#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace/--stats run) as text:
per-kernel calls / total / average duration, plus PMC counter sums if present.
    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r01_x.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print(f"# rocprofv3 summary of {path}")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  kernel")
    rows = list(c.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name "
                          "order by sum(end - start) desc"))
    tot = sum(r[2] for r in rows) or 1
    for name, calls, total, avg in rows:   # start/end are nanoseconds
        print(f"{calls:7d} {total / 1e6:10.3f} {avg / 1e3:10.2f} {100.0 * total / tot:6.2f}  {name[:150]}")
    try:
        rows = list(c.execute("select counter_name, sum(value), count(*) from counters_collection group by counter_name"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n# PMC counters (sum over dispatches)")
        for n, v, k in rows:
            print(f"{n:32s} {v:20.1f}  over {k} samples")
        try:
            print("\n# PMC per kernel")
            for kn, cn, v, k in c.execute(
                    "select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                    "group by kernel_name, counter_name order by kernel_name"):
                print(f"{cn:28s} {v:20.1f} n={k:5d}  {kn[:110]}")
        except sqlite3.Error as e:
            print("# (per-kernel PMC view unavailable:", e, ")")


if __name__ == "__main__":
    main(sys.argv[1])
