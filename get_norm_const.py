#!/usr/bin/env python3
"""Drop-in for the reference's ``get_norm_const.py`` (same flags) on the MI355X hot path: the
normalising constant of the discriminator-rejection score = the largest odds l/(1-l) of the tocg
discriminator's logit over `--length` samples (get_norm_const.py:65-132), computed by
``hr_viton_amd.rejection.get_const``.  ``--synthetic`` feeds VITON-HD-shaped random batches."""
import argparse
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd.networks import ConditionGenerator, define_D, load_checkpoint  # noqa: E402
from hr_viton_amd.rejection import get_const  # noqa: E402
from train_condition import synthetic_batch  # noqa: E402


def get_opt(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpu_ids", default="")
    p.add_argument("-j", "--workers", type=int, default=4)
    p.add_argument("-b", "--batch-size", type=int, default=8)
    p.add_argument("--fp16", action="store_true", help="use amp")
    p.add_argument("--dataroot", default="./data")
    p.add_argument("--datamode", default="train")
    p.add_argument("--data_list", default="train_pairs_zalando.txt")
    p.add_argument("--fine_width", type=int, default=192)
    p.add_argument("--fine_height", type=int, default=256)
    p.add_argument("--tensorboard_dir", type=str, default="tensorboard")
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints")
    p.add_argument("--D_checkpoint", type=str, default="")
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
    p.add_argument("--clothmask_composition", type=str, choices=["no_composition", "detach", "warp_grad"],
                   default="warp_grad")
    p.add_argument("--Ddownx2", action="store_true")
    p.add_argument("--Ddropout", action="store_true")
    p.add_argument("--num_D", type=int, default=2)
    p.add_argument("--spectral", action="store_true")
    p.add_argument("--test_datasetting", default="unpaired")
    p.add_argument("--test_dataroot", default="./data/zalando-hd-resize")
    p.add_argument("--test_data_list", default="test_pairs.txt")
    # additions
    p.add_argument("--cuda", default=True)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--length", type=int, default=0, help="samples to scan (reference: the whole training set)")
    p.add_argument("--ngf", type=int, default=96)
    return p.parse_args(argv)


def main(argv=None):
    opt = get_opt(argv)
    print(opt)
    dev = torch.device("cuda", 0)
    input1_nc, input2_nc = 4, opt.semantic_nc + 3
    D = define_D(input_nc=input1_nc + input2_nc + opt.output_nc, Ddownx2=opt.Ddownx2, Ddropout=opt.Ddropout,
                 n_layers_D=3, spectral=opt.spectral, num_D=opt.num_D)
    tocg = ConditionGenerator(opt, input1_nc=input1_nc, input2_nc=input2_nc, output_nc=opt.output_nc, ngf=opt.ngf,
                              norm_layer=nn.BatchNorm2d)
    if opt.D_checkpoint:
        load_checkpoint(D, opt.D_checkpoint, opt)
    if opt.tocg_checkpoint:
        load_checkpoint(tocg, opt.tocg_checkpoint, opt)
    tocg.to(dev)
    D.to(dev)
    if opt.synthetic:
        length = opt.length or 4 * opt.batch_size
        batches = (synthetic_batch(opt, opt.batch_size, 777 + i, dev) for i in range(length // opt.batch_size))
    else:
        from hr_viton_amd.cp_dataset import CPDataLoader, CPDataset
        from train_condition import disk_batch
        ds = CPDataset(opt)
        loader = CPDataLoader(opt, ds)
        length = opt.length or len(ds)
        batches = (disk_batch(loader.next_batch(), dev) for _ in range(length // opt.batch_size))
    print(get_const(opt, batches, tocg, D))


if __name__ == "__main__":
    main()
