#!/usr/bin/env python3
"""Drop-in for the reference's ``train_generator.py`` (same flags / loop / checkpoint files)
on the MI355X-native hot path.

  * one process per GPU: ``python -m torch.distributed.run --nproc-per-node N train_generator.py ...``
    (``--gpu_ids 0,1,..`` is accepted for CLI compatibility; the rank's device is LOCAL_RANK).  Replicas
    hold identical weights; gradients are bucket-all-reduced on RCCL during the backward
    (hr_viton_amd.parallel.GradSync) instead of the reference's nn.DataParallel wrapper.
  * ``-b`` is the GLOBAL batch like the reference (split over ranks, train_generator.py:124-126).
  * ``--fp16``: mixed precision like the reference's apex O1 -- every training convolution (forward, data and
    weight gradient) runs on the bf16 matrix cores over fp32 tensors (hr_viton_amd.train_ops.MMA_BF16);
    normalisations, losses, the optimizer and everything in HBM stay fp32.
  * ``--synthetic`` feeds VITON-HD-shaped random batches (no dataset / torchvision in this image);
    tensorboard / LPIPS evaluation blocks (train_generator.py:364-584) are out of scope.
Bug-fixes of the reference call sites: tocg(input1, input2) arity, load_checkpoint arity,
Adam betas as floats.
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
from hr_viton_amd.checkpoint import load_checkpoint, save_checkpoint  # noqa: E402
from hr_viton_amd.gen_train import attach_grad_sync  # noqa: E402
from hr_viton_amd.losses import GANLoss, L1Loss  # noqa: E402
from hr_viton_amd.network_generator import MultiscaleDiscriminator, SPADEGenerator  # noqa: E402
from hr_viton_amd.networks import ConditionGenerator  # noqa: E402
from hr_viton_amd.optim import Adam  # noqa: E402
from hr_viton_amd.parallel import broadcast_module  # noqa: E402
from hr_viton_amd.pipeline import generator_train_step, make_generator_inputs  # noqa: E402
from hr_viton_amd.vgg import VGGLoss  # noqa: E402


def get_opt(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--name", type=str, required=True)
    p.add_argument("--gpu_ids", type=str, default="0")
    p.add_argument("-j", "--workers", type=int, default=4)
    p.add_argument("-b", "--batch_size", type=int, default=8)
    p.add_argument("--fp16", action="store_true", help="use amp")
    p.add_argument("--cuda", default=True)
    p.add_argument("--dataroot", default="./data/")
    p.add_argument("--datamode", default="train")
    p.add_argument("--data_list", default="train_pairs.txt")
    p.add_argument("--fine_width", type=int, default=768)
    p.add_argument("--fine_height", type=int, default=1024)
    p.add_argument("--radius", type=int, default=20)
    p.add_argument("--grid_size", type=int, default=5)
    p.add_argument("--tensorboard_dir", type=str, default="tensorboard")
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints")
    p.add_argument("--tocg_checkpoint", type=str)
    p.add_argument("--gen_checkpoint", type=str, default="")
    p.add_argument("--dis_checkpoint", type=str, default="")
    p.add_argument("--tensorboard_count", type=int, default=100)
    p.add_argument("--display_count", type=int, default=100)
    p.add_argument("--save_count", type=int, default=10000)
    p.add_argument("--load_step", type=int, default=0)
    p.add_argument("--keep_step", type=int, default=100000)
    p.add_argument("--decay_step", type=int, default=100000)
    p.add_argument("--shuffle", action="store_true")
    p.add_argument("--lpips_count", type=int, default=1000)
    p.add_argument("--test_datasetting", default="paired")
    p.add_argument("--test_dataroot", default="./data/")
    p.add_argument("--test_data_list", default="test_pairs.txt")
    p.add_argument("--G_lr", type=float, default=0.0001)
    p.add_argument("--D_lr", type=float, default=0.0004)
    p.add_argument("--GMM_const", type=float, default=None)
    p.add_argument("--semantic_nc", type=int, default=13)
    p.add_argument("--gen_semantic_nc", type=int, default=7)
    p.add_argument("--norm_G", type=str, default="spectralaliasinstance")
    p.add_argument("--norm_D", type=str, default="spectralinstance")
    p.add_argument("--ngf", type=int, default=64)
    p.add_argument("--ndf", type=int, default=64)
    p.add_argument("--num_upsampling_layers", choices=["normal", "more", "most"], default="most")
    p.add_argument("--init_type", type=str, default="xavier")
    p.add_argument("--init_variance", type=float, default=0.02)
    p.add_argument("--no_ganFeat_loss", action="store_true")
    p.add_argument("--no_vgg_loss", action="store_true")
    p.add_argument("--lambda_l1", type=float, default=1.0)
    p.add_argument("--lambda_feat", type=float, default=10.0)
    p.add_argument("--lambda_vgg", type=float, default=10.0)
    p.add_argument("--n_layers_D", type=int, default=3)
    p.add_argument("--netD_subarch", type=str, default="n_layer")
    p.add_argument("--num_D", type=int, default=2)
    p.add_argument("--GT", action="store_true")
    p.add_argument("--occlusion", action="store_true")
    p.add_argument("--warp_feature", choices=["encoder", "T1"], default="T1")
    p.add_argument("--out_layer", choices=["relu", "conv"], default="relu")
    p.add_argument("--clothmask_composition", type=str, choices=["no_composition", "detach", "warp_grad"], default="warp_grad")
    p.add_argument("--num_test_visualize", type=int, default=3)
    # additions
    p.add_argument("--synthetic", action="store_true", help="synthetic VITON-HD-shaped batches")
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many steps (0: keep_step+decay_step)")
    p.add_argument("--tocg_ngf", type=int, default=96)
    p.add_argument("--vgg_weights", type=str, default="", help="torchvision vgg19 state_dict (.pth): the reference's models.vgg19(pretrained=True) weights")
    p.add_argument("--vgg_random_init", action="store_true",
                   help="plumbing / bench runs only: a RANDOMLY initialised VGG19 in the perceptual loss (no network here to "
                        "download the pretrained weights); implied by --synthetic")
    opt = p.parse_args(argv)
    opt.gpu_ids = [int(s) for s in str(opt.gpu_ids).split(",") if s.strip() and int(s) >= 0]
    return opt


def synthetic_batch(opt, n, seed, device):
    g = torch.Generator().manual_seed(seed)
    H, W = opt.fine_height, opt.fine_width
    lab = torch.randint(0, 13, (n, 1, H // 32, W // 32), generator=g).repeat_interleave(32, 2).repeat_interleave(32, 3)
    u = lambda c: (torch.rand(n, c, H, W, generator=g) * 2 - 1).to(device)  # noqa: E731
    onehot = torch.zeros(n, 13, H, W).scatter_(1, lab, 1.0).to(device)
    return {"cloth": u(3), "cloth_mask": (torch.rand(n, 1, H, W, generator=g) > 0.4).float().to(device),
            "parse_agnostic": onehot, "densepose": u(3), "agnostic": u(3), "image": u(3),
            "parse": onehot, "parse_cloth": u(3)}


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

    tocg = ConditionGenerator(opt, input1_nc=4, input2_nc=opt.semantic_nc + 3, output_nc=opt.semantic_nc,
                              ngf=opt.tocg_ngf, norm_layer=nn.BatchNorm2d)
    if opt.tocg_checkpoint:
        load_checkpoint(tocg, opt.tocg_checkpoint, opt)
    tocg.to(dev).eval()
    generator = SPADEGenerator(opt, 3 + 3 + 3)
    generator.print_network() if rank == 0 else None
    generator.init_weights(opt.init_type, opt.init_variance)
    discriminator = MultiscaleDiscriminator(opt)
    discriminator.init_weights(opt.init_type, opt.init_variance)
    if opt.gen_checkpoint and os.path.exists(opt.gen_checkpoint):
        load_checkpoint(generator, opt.gen_checkpoint, opt)
    if opt.dis_checkpoint and os.path.exists(opt.dis_checkpoint):
        load_checkpoint(discriminator, opt.dis_checkpoint, opt)
    generator.to(dev).train()
    discriminator.to(dev).train()
    broadcast_module(generator)
    broadcast_module(discriminator)

    crit_gan, crit_feat = GANLoss("hinge"), L1Loss()
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

    opt_g = Adam(generator.parameters(), lr=opt.G_lr, betas=(0.0, 0.9))
    opt_d = Adam(discriminator.parameters(), lr=opt.D_lr, betas=(0.0, 0.9))
    # in-place bucketed all-reduce on the optimizers' flat gradient buffers, fired from inside the backward
    sync_g = opt_g.make_grad_sync() if world > 1 else None
    sync_d = opt_d.make_grad_sync() if world > 1 else None
    for s in (sync_g, sync_d):
        if s is not None:
            attach_grad_sync(s)
    lam = lambda step: 1.0 - max(0, step * 1000 + opt.load_step - opt.keep_step) / float(opt.decay_step + 1)  # noqa: E731
    sched_g = torch.optim.lr_scheduler.LambdaLR(opt_g, lr_lambda=lam)
    sched_d = torch.optim.lr_scheduler.LambdaLR(opt_d, lr_lambda=lam)

    loader = None
    if not opt.synthetic:
        import copy
        from hr_viton_amd.cp_dataset import CPDataLoader, CPDataset
        o = copy.copy(opt)
        o.batch_size = per_rank
        torch.manual_seed(hdist.shard_seed(97, rank))
        loader = CPDataLoader(o, CPDataset(o), rank, world)
    last = opt.keep_step + opt.decay_step
    if opt.max_steps:
        last = min(last, opt.load_step + opt.max_steps)
    for step in range(opt.load_step, last):
        t0 = time.time()
        if loader is None:
            batch = synthetic_batch(opt, per_rank, hdist.shard_seed(1234 + step * 97, rank), dev)
        else:
            raw = loader.next_batch()                                  # train_generator.py:194-212
            batch = {"cloth": raw["cloth"]["paired"].to(dev), "cloth_mask": raw["cloth_mask"]["paired"].to(dev),
                     "parse_agnostic": raw["parse_agnostic"].to(dev), "densepose": raw["densepose"].to(dev),
                     "agnostic": raw["agnostic"].to(dev), "image": raw["image"].to(dev),
                     "parse": raw["parse"].to(dev), "parse_cloth": raw["parse_cloth"].to(dev)}
        x, parse7 = make_generator_inputs(opt, tocg, batch)
        losses, _ = generator_train_step(opt, generator, discriminator, crit_gan, crit_feat, crit_vgg, opt_g, opt_d, x,
                                         parse7, batch["image"], sync_g, sync_d)
        if (step + 1) % opt.display_count == 0 and rank == 0:
            torch.cuda.synchronize()
            t = time.time() - t0
            ld = sum(v.item() for k, v in losses.items() if k.startswith("D_"))
            lg = sum(v.item() for k, v in losses.items() if not k.startswith("D_"))
            print("step: %8d, time: %.3f, G_loss: %.4f, G_adv_loss: %.4f, D_loss: %.4f, D_fake_loss: %.4f, D_real_loss: %.4f"
                  % (step + 1, t, lg, losses["GAN"].item(), ld, losses["D_Fake"].item(), losses["D_Real"].item()), flush=True)
        if (step + 1) % opt.save_count == 0 and rank == 0:
            save_checkpoint(generator, os.path.join(opt.checkpoint_dir, opt.name, "gen_step_%06d.pth" % (step + 1)), opt)
            save_checkpoint(discriminator, os.path.join(opt.checkpoint_dir, opt.name, "dis_step_%06d.pth" % (step + 1)), opt)
        if (step + 1) % 1000 == 0:
            sched_g.step()
            sched_d.step()
    if rank == 0:
        save_checkpoint(generator, os.path.join(opt.checkpoint_dir, opt.name, "gen_model_final.pth"), opt)
        save_checkpoint(discriminator, os.path.join(opt.checkpoint_dir, opt.name, "dis_model_final.pth"), opt)
        print("Finished training %s!" % opt.name)


if __name__ == "__main__":
    main()
