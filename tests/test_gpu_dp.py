"""GPU: the data-parallel TRAINING path end to end with two ranks (both on cuda:0, gloo carrying the CUDA
tensors -- RCCL refuses two ranks on one device; the collective calls, bucket logic, in-place flat-buffer
reduction, fused Adam and replica broadcast are the ones the 8-GPU RCCL run uses)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), HRV_DIST_BACKEND="gloo")
        import torch.distributed as dist
        import torch.nn as nn
        import hr_viton_amd  # noqa: F401
        from hr_viton_amd import dist as hdist
        from hr_viton_amd.gen_train import attach_grad_sync
        from hr_viton_amd.losses import L1Loss
        from hr_viton_amd.networks import ConditionGenerator, GANLoss, define_D
        from hr_viton_amd.optim import Adam
        from hr_viton_amd.parallel import broadcast_module
        from hr_viton_amd.pipeline import condition_train_step
        import train_condition as tc
        hdist.init_from_env()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        opt = tc.get_opt(["--synthetic", "--Ddownx2", "--lasttvonly", "--interflowloss", "-b", "2", "--fine_height", "128",
                          "--fine_width", "96", "--ngf", "8", "--no_vgg_loss"])
        torch.manual_seed(100 + rank)                   # replicas start different; broadcast makes them equal
        tocg = ConditionGenerator(opt, 4, 16, 13, ngf=8, norm_layer=nn.BatchNorm2d).to(dev).train()
        D = define_D(input_nc=33, Ddownx2=True, Ddropout=False, n_layers_D=3, spectral=False, num_D=2).to(dev).train()
        broadcast_module(tocg)
        broadcast_module(D)
        w0 = torch.cat([p.detach().flatten() for p in tocg.parameters()]).clone()
        og = Adam(tocg.parameters(), lr=2e-4, betas=(0.5, 0.999))
        od = Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
        sg, sd = og.make_grad_sync(bucket_mb=0.25), od.make_grad_sync(bucket_mb=4.0)
        attach_grad_sync(sg)
        attach_grad_sync(sd)
        assert len(sg.buckets) >= 2
        losses = None
        for step in range(2):
            batch = tc.synthetic_batch(opt, 1, 50 + 10 * step + rank, dev)      # a different sample per rank
            losses = condition_train_step(opt, tocg, D, L1Loss(), None, GANLoss(use_lsgan=True), og, od, batch, sg, sd)
        torch.cuda.synchronize()
        wg = torch.cat([p.detach().flatten() for p in tocg.parameters()])
        wd = torch.cat([p.detach().flatten() for p in D.parameters()])
        sums = torch.stack([wg.double().sum(), wg.double().abs().sum(), wd.double().sum(), wd.double().abs().sum()]).cpu()
        gathered = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(gathered, sums)
        moved = float((wg - w0).abs().max())
        q.put((rank, [g.tolist() for g in gathered], moved, float(losses["loss_G"].detach()), float(losses["loss_D"].detach())))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "ERROR: " + traceback.format_exc()))


def test_two_rank_condition_training_keeps_replicas_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    for r in res:
        assert len(r) == 5, r
    (_, g0, moved0, lg0, ld0), (_, g1, moved1, lg1, ld1) = res
    assert g0 == g1 and g0[0] == g0[1], ("replica weight checksums differ", g0, g1)      # bitwise identical replicas
    assert moved0 > 1e-5 and moved0 == moved1
    assert all(map(lambda v: v == v and abs(v) < 1e6, (lg0, ld0, lg1, ld1)))
    assert lg0 != lg1                                   # the ranks really saw different data
