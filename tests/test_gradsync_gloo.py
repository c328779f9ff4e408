"""CPU, world_size 2, gloo: the data-parallel gradient path (parallel.GradSync): bucket assignment,
all-reduce fired from inside the backward as buckets complete, unused parameters, averaging."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, inplace=False):
    try:
        _worker_body(rank, world, port, q, inplace)
    except Exception as e:  # noqa: BLE001 -- surface the failure instead of a queue timeout
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))


def _worker_body(rank, world, port, q, inplace=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import dist as hdist
    from hr_viton_amd.gen_train import _acc, attach_grad_sync, detach_grad_sync
    from hr_viton_amd.parallel import GradSync, broadcast_module
    hdist.init_from_env("gloo")
    torch.manual_seed(rank)                         # replicas start different ...
    net = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.Linear(200, 100), torch.nn.Linear(100, 7))
    broadcast_module(net)                           # ... and are made identical
    w0 = net[0].weight.detach().clone()
    if inplace:
        # what optim.Adam.make_grad_sync sets up on the GPU: one flat gradient buffer (16-byte aligned slots), every
        # parameter knows its slot, buckets are contiguous slices of the buffer reduced in place
        plist = list(net.parameters())
        offs, n = [], 0
        for p in plist:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(n)
        spans = [(p, o, p.numel()) for p, o in zip(plist, offs)]
        for p, o, k in spans:
            p._hrv_flat_grad = flat[o:o + k].view_as(p.data)
        sync = GradSync(None, bucket_mb=0.1, flat=flat, spans=spans)
    else:
        sync = GradSync(net.parameters(), bucket_mb=0.1)   # ~26k floats per bucket -> several buckets
    attach_grad_sync(sync)
    assert len(sync.buckets) >= 2
    params = list(net.parameters())
    fired_early = []
    sync.begin()
    grads = {}
    # the "backward plan": gradients appear in reverse order; the last Linear's bias is unused
    for p in reversed(params[:-1]):
        if inplace:
            from hr_viton_amd.gen_train import grad_buffer
            g = grad_buffer(p)                      # the plan writes the gradient straight into the flat slot
            assert g.data_ptr() == p._hrv_flat_grad.data_ptr()
            g.fill_(float(rank + 1))
            _acc(grads, p, g)
            assert p.grad is not None and p.grad.data_ptr() == g.data_ptr() and p not in grads
        else:
            _acc(grads, p, torch.full_like(p, float(rank + 1)))
        fired_early.append(sum(1 for b in sync.buckets if b["handle"] is not None))
    sync.wait()
    if inplace:     # the reduced values ARE the flat buffer: nothing was copied out of or into it
        assert all(sync.grad_of(p).data_ptr() == p._hrv_flat_grad.data_ptr() for p in params[:-1])
    ok = all(torch.allclose(sync.grad_of(p), torch.full_like(p, 3.0)) for p in params[:-1])   # 1 + 2 summed
    unused = sync.grad_of(params[-1])
    q.put((rank, bool(ok), unused is None, fired_early[-1] >= 1 and fired_early[0] == 0 or len(sync.buckets) == 1,
           float(w0.sum()), sync.world))
    detach_grad_sync(params)
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("inplace", [False, True], ids=["own_buckets", "inplace_flat_buffer"])
def test_gradsync_world2_gloo(inplace):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, inplace)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for r in res:
        assert len(r) == 6, r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, ok0, un0, early0, w0, world0), (r1, ok1, un1, early1, w1, world1) = res
    assert ok0 and ok1, "all-reduced gradients must be the sum over ranks"
    assert un0 and un1, "a parameter that got no gradient reports None (its bucket is flushed with zeros)"
    assert early0 and early1, "buckets fire during the backward, not only at the end"
    assert w0 == w1, "broadcast_module makes the replicas identical"
    assert world0 == world1 == 2


def test_bucket_cut_policy_big_parameters_alone_and_a_small_tail():
    """GradSync.cut_ranges: ranges cover every parameter exactly once, are listed in firing order (back to front), none but a
    single oversized parameter exceeds the cap, a parameter of >= big elements is a bucket of its own, and the bucket that
    completes LAST (the front of the flat buffer) holds at most ``tail`` elements -- the generator's shapes: 37.7 MB
    head_0 / G_middle weights, 64 MiB cap."""
    import sys
    sys.path.insert(0, ROOT)
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd.parallel import GradSync
    MB = (1 << 20) // 4
    big3x3 = 1024 * 1024 * 9                              # 37.7 MB
    gb = 1024 * 128 * 9                                   # 4.7 MB (conv_gamma / conv_beta of a 1024-channel norm)
    sizes = ([128 * 7 * 9, 128, gb, 1024, gb, 1024, 1024] * 2 + [big3x3, 1024, big3x3, 1024]) * 3 + [64 * 64 * 9, 64, 3 * 64 * 9, 3]
    ranges = GradSync.cut_ranges(sizes, 64 * MB, 16 * MB, 8 * MB)
    covered = sorted(i for lo, hi in ranges for i in range(lo, hi))
    assert covered == list(range(len(sizes)))
    assert all(ranges[k][0] == ranges[k + 1][1] for k in range(len(ranges) - 1)) and ranges[0][1] == len(sizes) and ranges[-1][0] == 0
    for lo, hi in ranges:
        n = sum(sizes[lo:hi])
        assert n <= 64 * MB or hi - lo == 1
        if any(sizes[i] >= 16 * MB for i in range(lo, hi)):
            assert hi - lo == 1
    lo, hi = ranges[-1]
    assert sum(sizes[lo:hi]) <= 8 * MB and hi - lo >= 1
    # degenerate inputs
    assert GradSync.cut_ranges([5], 10, 100, 3) == [(0, 1)]
    assert GradSync.cut_ranges([200, 1, 1], 10, 100, 3) == [(1, 3), (0, 1)]


def _graph_sync_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch
        import torch.distributed as dist
        import hr_viton_amd  # noqa: F401
        from hr_viton_amd import dist as hdist
        from hr_viton_amd.gen_train import _acc, attach_grad_sync, detach_grad_sync, grad_buffer
        from hr_viton_amd.parallel import GraphGradSync
        hdist.init_from_env("gloo")
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.Linear(200, 100), torch.nn.Linear(100, 7))
        plist = list(net.parameters())
        offs, n = [], 0
        for p in plist:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(n)
        spans = [(p, o, p.numel()) for p, o in zip(plist, offs)]
        for p, o, k in spans:
            p._hrv_flat_grad = flat[o:o + k].view_as(p.data)
        sync = GraphGradSync(flat, spans, bucket_mb=0.1)
        attach_grad_sync(sync)
        cuts = []
        sync.cut = cuts.append            # only called while a HIP stream is capturing: never on the CPU
        sync.begin()
        grads = {}
        fired = []
        for p in reversed(plist[:-1]):    # the last bias gets no gradient
            g = grad_buffer(p)
            g.fill_(float(rank + 1))
            _acc(grads, p, g)
            fired.append(any(b["handle"] not in (None, True) for b in sync.buckets))
        flat_before = flat.clone()
        sync.wait()                       # eager: zero-fills the unused parameter, reduces the whole buffer over the bucket ranges
        ok = all(torch.equal(sync.grad_of(p), torch.full_like(p, 3.0)) for p in plist[:-1])
        lo = offs[-1]
        q.put((rank, ok, not any(fired), bool((flat_before[:lo] == rank + 1).all()), bool((flat[lo:lo + plist[-1].numel()] == 0).all()),
               len(cuts), len(sync.buckets)))
        detach_grad_sync(plist)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))


def test_graph_grad_sync_reduces_nothing_during_the_backward_and_everything_at_the_wait():
    """parallel.GraphGradSync (the synchronisation of a hipGraph-captured data-parallel iteration): the backward plans hand their
    gradients over as to GradSync, NO collective starts during the backward (a captured region cannot host one), and ``wait()`` --
    outside a capture -- reduces the optimizer's whole flat buffer over the bucket ranges: sums over ranks, zeros for a parameter
    without a gradient.  (The capture-time protocol -- wait() calls ``cut`` instead -- runs in tests/test_gpu_dp.py.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    for r in res:
        assert len(r) == 7, r
        _, ok, quiet, untouched, zeroed, ncuts, nb = r
        assert ok and quiet and untouched and zeroed and ncuts == 0 and nb >= 2
