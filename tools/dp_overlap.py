#!/usr/bin/env python3
"""How much of the (stand-in) gradient collectives the remaining backward kernels hide, from a rocprofv3
--kernel-trace database of `HRV_FAKE_ALLREDUCE=<r> python bench.py ...` (parallel._FakeCollective: r device copies of each
bucket on a side stream, fired where the RCCL all-reduce is fired).  Side-stream kernels are the ones not on the
busiest queue/stream; for their busy intervals we report the share during which a compute kernel was running too.
    python tools/dp_overlap.py trace_results.db > profiles/r02_dp_overlap.txt
"""
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def inter_len(A, B):
    i = j = 0
    tot = 0
    while i < len(A) and j < len(B):
        lo, hi = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
        if hi > lo:
            tot += hi - lo
        if A[i][1] < B[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    key = next((k for k in ("stream_id", "queue_id", "stream", "queue") if k in cols), None)
    print(f"# {path}\n# kernels columns: {cols}\n# lane key: {key}")
    if key is None:
        return
    rows = list(c.execute(f"select {key}, start, end, name from kernels order by start"))
    busy = {}
    for k, a, b, n in rows:
        busy.setdefault(k, []).append((a, b, n))
    main_k = max(busy, key=lambda k: sum(b - a for a, b, _ in busy[k]))
    comp = union([(a, b) for a, b, _ in busy[main_k]])
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    print(f"# trace span {1e-6 * (t1 - t0):.1f} ms; compute lane {main_k}: {len(busy[main_k])} kernels, "
          f"busy {1e-6 * sum(b - a for a, b in comp):.1f} ms")
    for k, lst in busy.items():
        if k == main_k:
            continue
        side = union([(a, b) for a, b, _ in lst])
        tot = sum(b - a for a, b in side)
        hid = inter_len(side, comp)
        names = {}
        for a, b, n in lst:
            names[n[:60]] = names.get(n[:60], 0) + 1
        print(f"side lane {k}: {len(lst)} kernels {names}, busy {1e-6 * tot:.2f} ms, concurrent with compute kernels "
              f"{1e-6 * hid:.2f} ms ({100.0 * hid / max(tot, 1):.1f} %), exposed {1e-6 * (tot - hid):.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
