#!/usr/bin/env python3
"""Which Python lines launch the ATen / rocclr kernels of the headline iteration (fills, copies, muls, cats: ~130 launches, ~1.4 ms)?
One profiled iteration under torch.profiler with Python stacks; prints per (operator, innermost repo frame) the launch count and
device time.  Run via gpurun from the repo root: python tools/diag/aten_sites.py > gpurun_out/aten_sites.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import argparse
    import torch
    import torch.nn as nn
    import hr_viton_amd  # noqa: F401
    import bench
    from hr_viton_amd import dist as hdist, ops
    args = argparse.Namespace(graph=False)
    ctx = dict(torch=torch, nn=nn, hdist=hdist, ops=ops, rank=0, world=1, dev=torch.device("cuda", 0), args=args, cpu_threads=8)
    wl = bench.make_workload(ctx, "train_generator", True, 0)
    for i in range(4):
        wl["step"](i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        wl["step"](0)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children:
            # leaf aten ops only (the ones that actually launch)
            if not (ev.name.startswith("aten::") and ev.device_time_total > 0 and not any(c.name.startswith("aten::") for c in ev.cpu_children)):
                continue
        site = "?"
        for fr in (ev.stack or []):
            if ROOT in fr and "site-packages" not in fr and "dist-packages" not in fr:
                site = fr.replace(ROOT + "/", "")
                break
        a = agg[(ev.name, site)]
        a[0] += 1
        a[1] += ev.device_time_total
    tot = sum(a[1] for a in agg.values())
    print(f"# ATen operators that launched device work in ONE iteration: {sum(a[0] for a in agg.values())} ops, {tot / 1e3:.3f} ms")
    for (name, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{n:4d} {us / 1e3:8.3f} ms  {name:28s} {site}")


if __name__ == "__main__":
    main()
