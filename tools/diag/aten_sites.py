#!/usr/bin/env python3
"""Which lines of the package still launch ATen kernels inside one train_generator.py iteration (fills, copies, cats,
casts)?  Runs the bench's headline workload, then ONE more iteration under a TorchDispatchMode that attributes every
ATen call touching a device tensor to the innermost frame inside hr-viton_amd/ (or the repo's scripts).

    python tools/diag/aten_sites.py [--capture]      # --capture: also try a hipGraph capture of the iteration
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--capture", action="store_true")
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    import torch
    import torch.nn as nn
    from torch.utils._python_dispatch import TorchDispatchMode
    import bench
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import dist as hdist
    from hr_viton_amd import ops

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    args = argparse.Namespace(graph=False)
    ctx = dict(torch=torch, nn=nn, hdist=hdist, ops=ops, rank=0, world=1, dev=dev, args=args, cpu_threads=8)
    wl = bench.wl_generator(ctx, True, a.batch, True)
    for i in range(3):
        wl["step"](i)
    torch.cuda.synchronize()

    sites = collections.Counter()
    names = collections.defaultdict(collections.Counter)

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = func.__name__ if hasattr(func, "__name__") else str(func)
            full = str(func)
            view = any(s in full for s in ("view", "as_strided", "select", "slice", "expand", "permute", "reshape", "detach",
                                           "alias", "unsqueeze", "squeeze", "transpose", "empty", "t.default", "unbind",
                                           "split", "narrow", "_unsafe_view", "unfold", "is_", "sym_", "stride", "size",
                                           "lift_fresh", "_local_scalar", "set_", "resize_", "chunk"))
            if view:
                return out
            cuda = False
            for t in list(args) + list((kwargs or {}).values()) + [out]:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    cuda = True
                elif isinstance(t, (list, tuple)):
                    cuda = cuda or any(isinstance(u, torch.Tensor) and u.is_cuda for u in t)
            if not cuda:
                return out
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if (fn.startswith(ROOT) and "tools/diag" not in fn) and "torch/" not in fn:
                    where = f"{os.path.relpath(fn, ROOT)}:{fr.lineno} {fr.name}"
                    break
            sites[where] += 1
            names[where][full] += 1
            return out

    with Mode():
        wl["step"](0)
    torch.cuda.synchronize()
    tot = sum(sites.values())
    print(f"ATen calls on device tensors in one iteration: {tot}")
    for w, n in sites.most_common(70):
        ops_ = ", ".join(f"{k.replace('aten.', '')} x{v}" for k, v in names[w].most_common(4))
        print(f"{n:5d}  {w:58s} {ops_}")

    if a.capture:
        try:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                wl["step"](0)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                wl["step"](0)
            torch.cuda.synchronize()
            import time
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                g.replay()
            torch.cuda.synchronize()
            print(f"hipGraph capture OK: replay {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per iteration")
        except Exception as e:        # noqa: BLE001
            print("hipGraph capture FAILED:", repr(e)[:2000])
            traceback.print_exc()


if __name__ == "__main__":
    main()
