#!/usr/bin/env python3
"""Which Python lines of the repo launch the ATen kernels of the headline iteration (fills, copies, cats, scalings: ~130 launches,
~1.4 ms per iteration)?  The torch entry points that launch them are wrapped for ONE iteration (all threads: the backward plans
run on autograd's thread) and every call is booked to the innermost frame inside the repo with the bytes it touches.
Run via gpurun from the repo root: python tools/diag/aten_sites.py > gpurun_out/aten_sites.txt"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BOOK = collections.defaultdict(lambda: [0, 0])
ON = [False]


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        fn = fr.filename
        if fn.startswith(ROOT) and "tools/diag" not in fn:
            return f"{os.path.relpath(fn, ROOT)}:{fr.lineno} {fr.name}"
    return "?"


def _bytes(x):
    import torch
    if isinstance(x, torch.Tensor):
        return x.numel() * x.element_size() if x.is_cuda else 0
    if isinstance(x, (list, tuple)):
        return sum(_bytes(t) for t in x)
    return 0


def wrap(owner, name, label=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        r = orig(*a, **k)
        if ON[0]:
            nb = max(_bytes(r), _bytes(a[0]) if a else 0)
            if nb > 0:
                e = BOOK[(label or name, _site())]
                e[0] += 1
                e[1] += nb
        return r
    setattr(owner, name, f)


def main():
    import argparse
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    import hr_viton_amd  # noqa: F401
    import bench
    from hr_viton_amd import dist as hdist, ops
    args = argparse.Namespace(graph=False)
    ctx = dict(torch=torch, nn=nn, hdist=hdist, ops=ops, rank=0, world=1, dev=torch.device("cuda", 0), args=args, cpu_threads=8)
    wl = bench.make_workload(ctx, "train_generator", True, 0)
    for i in range(4):
        wl["step"](i)
    torch.cuda.synchronize()
    for nm in ("copy_", "fill_", "zero_", "clone", "contiguous", "to", "add_", "mul_", "div_", "add", "mul", "div", "__add__", "__mul__",
               "__truediv__", "__rmul__", "__radd__", "__setitem__", "float", "normal_", "sum", "mean"):
        wrap(torch.Tensor, nm)
    for nm in ("zeros", "zeros_like", "full", "cat", "ones", "full_like", "randn", "stack"):
        wrap(torch, nm, "torch." + nm)
    wrap(F, "pad", "F.pad")
    ON[0] = True
    wl["step"](0)
    torch.cuda.synchronize()
    ON[0] = False
    print("# torch entry points that touched CUDA tensors in ONE iteration, by call site (calls, MB touched)")
    for (name, site), (n, nb) in sorted(BOOK.items(), key=lambda kv: -kv[1][1])[:90]:
        print(f"{n:4d} {nb / 1e6:10.2f} MB  {name:16s} {site}")


if __name__ == "__main__":
    main()
