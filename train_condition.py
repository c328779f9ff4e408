#!/usr/bin/env python3
"""Drop-in for the reference's ``train_condition.py`` (same flags / loop / checkpoint files) on the
MI355X-native hot path: ConditionGenerator with batch-statistics BatchNorm, its multiscale
discriminator, and the L1 / VGG / TV / interflow / cross-entropy / LSGAN losses all run on the HIP
kernels (hr_viton_amd.cond_train, .functional, .losses, .vgg); one iteration is
``hr_viton_amd.pipeline.condition_train_step`` (train_condition.py:136-286).

  * one process per GPU: ``python -m torch.distributed.run --nproc-per-node N train_condition.py ...``;
    ``-b`` is the GLOBAL batch, split over the ranks; BatchNorm uses per-GPU batch statistics
    (the north star replaces sync_batchnorm by per-GPU BN) and gradients are bucket-all-reduced
    on RCCL during the backward (hr_viton_amd.parallel.GradSync).
  * ``--synthetic`` feeds VITON-HD-shaped random batches (no dataset / torchvision in this image);
    the tensorboard / validation-IoU blocks (train_condition.py:311-418) are out of scope.
  * --warp_feature encoder / --out_layer conv (networks.py:46-61,142-144) and --upsample nearest (the inter-flow loss's flow
    resize, train_condition.py:242 -- the only place the reference's scripts use the flag) are on the HIP path (round 5).
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import dist as hdist  # noqa: E402
from hr_viton_amd.gen_train import attach_grad_sync  # noqa: E402
from hr_viton_amd.losses import L1Loss  # noqa: E402
from hr_viton_amd.networks import (ConditionGenerator, GANLoss, VGGLoss, define_D, load_checkpoint,  # noqa: E402
                                   save_checkpoint)
from hr_viton_amd.optim import Adam  # noqa: E402
from hr_viton_amd.parallel import broadcast_module  # noqa: E402
from hr_viton_amd.pipeline import condition_train_step  # noqa: E402


def get_opt(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--name", default="test")
    p.add_argument("--gpu_ids", default="")
    p.add_argument("-j", "--workers", type=int, default=4)
    p.add_argument("-b", "--batch-size", type=int, default=8)
    p.add_argument("--fp16", action="store_true", help="use amp")
    p.add_argument("--dataroot", default="./data/")
    p.add_argument("--datamode", default="train")
    p.add_argument("--data_list", default="train_pairs.txt")
    p.add_argument("--fine_width", type=int, default=192)
    p.add_argument("--fine_height", type=int, default=256)
    p.add_argument("--tensorboard_dir", type=str, default="tensorboard")
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints")
    p.add_argument("--tocg_checkpoint", type=str, default="")
    p.add_argument("--tensorboard_count", type=int, default=100)
    p.add_argument("--display_count", type=int, default=100)
    p.add_argument("--save_count", type=int, default=10000)
    p.add_argument("--load_step", type=int, default=0)
    p.add_argument("--keep_step", type=int, default=300000)
    p.add_argument("--shuffle", action="store_true")
    p.add_argument("--semantic_nc", type=int, default=13)
    p.add_argument("--output_nc", type=int, default=13)
    p.add_argument("--warp_feature", choices=["encoder", "T1"], default="T1")
    p.add_argument("--out_layer", choices=["relu", "conv"], default="relu")
    p.add_argument("--Ddownx2", action="store_true")
    p.add_argument("--Ddropout", action="store_true")
    p.add_argument("--num_D", type=int, default=2)
    p.add_argument("--cuda", default=True)
    p.add_argument("--G_D_seperate", action="store_true")
    p.add_argument("--no_GAN_loss", action="store_true")
    p.add_argument("--lasttvonly", action="store_true")
    p.add_argument("--interflowloss", action="store_true")
    p.add_argument("--clothmask_composition", type=str, choices=["no_composition", "detach", "warp_grad"],
                   default="warp_grad")
    p.add_argument("--edgeawaretv", type=str, choices=["no_edge", "last_only", "weighted"], default="no_edge")
    p.add_argument("--add_lasttv", action="store_true")
    p.add_argument("--no_test_visualize", action="store_true")
    p.add_argument("--num_test_visualize", type=int, default=3)
    p.add_argument("--test_datasetting", default="unpaired")
    p.add_argument("--test_dataroot", default="./data/")
    p.add_argument("--test_data_list", default="test_pairs.txt")
    p.add_argument("--G_lr", type=float, default=0.0002)
    p.add_argument("--D_lr", type=float, default=0.0002)
    p.add_argument("--CElamda", type=float, default=10)
    p.add_argument("--GANlambda", type=float, default=1)
    p.add_argument("--tvlambda", type=float, default=2)
    p.add_argument("--upsample", type=str, default="bilinear", choices=["nearest", "bilinear"])
    p.add_argument("--val_count", type=int, default=1000)
    p.add_argument("--spectral", action="store_true")
    p.add_argument("--occlusion", action="store_true")
    # additions
    p.add_argument("--synthetic", action="store_true", help="synthetic VITON-HD-shaped batches")
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many steps (0: keep_step)")
    p.add_argument("--ngf", type=int, default=96)
    p.add_argument("--no_vgg_loss", action="store_true", help="drop the VGG terms (train_condition.py always has them)")
    p.add_argument("--vgg_weights", type=str, default="", help="torchvision vgg19 state_dict (.pth): the reference's models.vgg19(pretrained=True) weights")
    p.add_argument("--vgg_random_init", action="store_true",
                   help="plumbing / bench runs only: a RANDOMLY initialised VGG19 in the perceptual loss (no network here to "
                        "download the pretrained weights); implied by --synthetic")
    opt = p.parse_args(argv)
    return opt


def synthetic_batch(opt, n, seed, device):
    """One cp_dataset.py-shaped batch: blocky label maps, smooth images."""
    g = torch.Generator().manual_seed(seed)
    H, W = opt.fine_height, opt.fine_width
    blk = 16
    lab = torch.randint(0, 13, (n, 1, H // blk, W // blk), generator=g).repeat_interleave(blk, 2).repeat_interleave(blk, 3)
    agn = torch.randint(0, 13, (n, 1, H // blk, W // blk), generator=g).repeat_interleave(blk, 2).repeat_interleave(blk, 3)
    u = lambda c: (torch.rand(n, c, H, W, generator=g) * 2 - 1)  # noqa: E731
    b = {"cloth": u(3), "cloth_mask": (torch.rand(n, 1, H // blk, W // blk, generator=g) > 0.4).float()
         .repeat_interleave(blk, 2).repeat_interleave(blk, 3),
         "parse_agnostic": torch.zeros(n, 13, H, W).scatter_(1, agn, 1.0), "densepose": u(3),
         "parse_onehot": lab.float(), "parse": torch.zeros(n, 13, H, W).scatter_(1, lab, 1.0),
         "pcm": (lab == 3).float(), "parse_cloth": u(3)}
    return {k: v.to(device) for k, v in b.items()}


def _rank_loader(opt, per_rank, rank, world):
    """CPDataset over the reference's on-disk layout; every rank draws its own shuffled stream of per_rank samples."""
    import copy
    from hr_viton_amd.cp_dataset import CPDataLoader, CPDataset
    o = copy.copy(opt)
    o.batch_size = per_rank
    torch.manual_seed(hdist.shard_seed(97, rank))     # the sampler's permutation differs per rank
    return CPDataLoader(o, CPDataset(o), rank, world)


def disk_batch(inputs, device):
    """cp_dataset.py batch -> the flat dictionary condition_train_step takes (train_condition.py:136-153)."""
    return {"cloth": inputs["cloth"]["paired"].to(device), "cloth_mask": inputs["cloth_mask"]["paired"].to(device),
            "parse_agnostic": inputs["parse_agnostic"].to(device), "densepose": inputs["densepose"].to(device),
            "parse_onehot": inputs["parse_onehot"].to(device), "parse": inputs["parse"].to(device),
            "pcm": inputs["pcm"].to(device), "parse_cloth": inputs["parse_cloth"].to(device)}


def main(argv=None):
    opt = get_opt(argv)
    rank, local_rank, world = hdist.init_from_env()
    if opt.fp16:
        from hr_viton_amd import train_ops as _T
        _T.MMA_BF16[0] = True      # bf16 matrix cores for the training convolutions (fp32 storage / accumulate)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    assert opt.batch_size % world == 0, "Batch size %d must be a multiple of # GPUs %d." % (opt.batch_size, world)
    per_rank = opt.batch_size // world
    if rank == 0:
        print(opt)
        print("Start to train %s!" % opt.name)
    input1_nc, input2_nc = 4, opt.semantic_nc + 3
    tocg = ConditionGenerator(opt, input1_nc=input1_nc, input2_nc=input2_nc, output_nc=opt.output_nc, ngf=opt.ngf,
                              norm_layer=nn.BatchNorm2d)
    D = define_D(input_nc=input1_nc + input2_nc + opt.output_nc, Ddownx2=opt.Ddownx2, Ddropout=opt.Ddropout,
                 n_layers_D=3, spectral=opt.spectral, num_D=opt.num_D)
    if opt.tocg_checkpoint and os.path.exists(opt.tocg_checkpoint):
        load_checkpoint(tocg, opt.tocg_checkpoint, opt)
    tocg.to(dev).train()
    D.to(dev).train()
    broadcast_module(tocg)
    broadcast_module(D)
    crit_l1, crit_gan = L1Loss(), GANLoss(use_lsgan=True)
    crit_vgg = None
    if not opt.no_vgg_loss:
        crit_vgg = VGGLoss(opt)
        if opt.vgg_weights:
            crit_vgg.vgg.load_torchvision_state_dict(torch.load(opt.vgg_weights, map_location="cpu"))
        elif opt.vgg_random_init or opt.synthetic:
            if rank == 0:
                print("WARNING: VGGLoss runs on a RANDOMLY initialised VGG19 (--vgg_random_init / --synthetic): the "
                      "perceptual term is not the reference's objective; pass --vgg_weights for real training.", flush=True)
        else:
            # the reference builds models.vgg19(pretrained=True) (networks.py:204); silently optimising random features
            # would be a different objective
            raise SystemExit("VGGLoss needs the pretrained torchvision vgg19 weights: pass --vgg_weights <state_dict.pth>, "
                             "or --no_vgg_loss, or --vgg_random_init for plumbing runs")
        crit_vgg.to(dev)
        broadcast_module(crit_vgg)
    opt_g = Adam(tocg.parameters(), lr=opt.G_lr, betas=(0.5, 0.999))
    opt_d = Adam(D.parameters(), lr=opt.D_lr, betas=(0.5, 0.999))
    sync_g = opt_g.make_grad_sync() if world > 1 else None
    sync_d = opt_d.make_grad_sync() if world > 1 else None
    for s in (sync_g, sync_d):
        if s is not None:
            attach_grad_sync(s)
    loader = None
    if not opt.synthetic:
        loader = _rank_loader(opt, per_rank, rank, world)
    last = opt.keep_step if not opt.max_steps else min(opt.keep_step, opt.load_step + opt.max_steps)
    for step in range(opt.load_step, last):
        t0 = time.time()
        if loader is None:
            batch = synthetic_batch(opt, per_rank, hdist.shard_seed(4321 + step * 89, rank), dev)
        else:
            batch = disk_batch(loader.next_batch(), dev)
        losses = condition_train_step(opt, tocg, D, crit_l1, crit_vgg, crit_gan, opt_g, opt_d, batch, sync_g, sync_d)
        if (step + 1) % opt.display_count == 0 and rank == 0:
            torch.cuda.synchronize()
            t = time.time() - t0
            f = lambda k: float(losses[k].detach()) if k in losses and torch.is_tensor(losses[k]) else float(losses.get(k, 0.0))  # noqa: E731
            print("step: %8d, time: %.3f\nloss G: %.4f, L1_cloth loss: %.4f, VGG loss: %.4f, TV loss: %.4f CE: %.4f, "
                  "G GAN: %.4f\nloss D: %.4f, D real: %.4f, D fake: %.4f"
                  % (step + 1, t, f("loss_G"), f("l1"), f("vgg"), f("tv"), f("ce"), f("g_gan"), f("loss_D"), f("d_real"),
                     f("d_fake")), flush=True)
        if (step + 1) % opt.save_count == 0 and rank == 0:
            save_checkpoint(tocg, os.path.join(opt.checkpoint_dir, opt.name, "tocg_step_%06d.pth" % (step + 1)), opt)
            save_checkpoint(D, os.path.join(opt.checkpoint_dir, opt.name, "D_step_%06d.pth" % (step + 1)), opt)
    if rank == 0:
        save_checkpoint(tocg, os.path.join(opt.checkpoint_dir, opt.name, "tocg_final.pth"), opt)
        save_checkpoint(D, os.path.join(opt.checkpoint_dir, opt.name, "D_final.pth"), opt)
        print("Finished training %s!" % opt.name)


if __name__ == "__main__":
    main()
