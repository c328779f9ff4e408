#!/usr/bin/env python3
"""Drop-in for the reference's ``test_condition.py`` (same flags) on the MI355X hot path: tocg inference
with the cloth-mask composition, and -- when a discriminator checkpoint and ``--norm_const`` are given --
the discriminator-rejection score per sample, written to ``rejection_prob.txt`` sorted by score
(test_condition.py:64-160).  The image grids of the reference (torchvision make_grid / tensorboard) are out
of scope; ``--synthetic`` feeds VITON-HD-shaped random batches."""
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
from hr_viton_amd.networks import ConditionGenerator, define_D, load_checkpoint  # noqa: E402
from hr_viton_amd.rejection import rejection_scores  # noqa: E402
from train_condition import synthetic_batch  # noqa: E402


def get_opt(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpu_ids", default="")
    p.add_argument("-j", "--workers", type=int, default=4)
    p.add_argument("-b", "--batch-size", type=int, default=8)
    p.add_argument("--fp16", action="store_true", help="use amp")
    p.add_argument("--dataroot", default="./data/zalando-hd-resize")
    p.add_argument("--datamode", default="test")
    p.add_argument("--data_list", default="test_pairs.txt")
    p.add_argument("--datasetting", default="paired")
    p.add_argument("--fine_width", type=int, default=192)
    p.add_argument("--fine_height", type=int, default=256)
    p.add_argument("--tensorboard_dir", type=str, default="tensorboard")
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints")
    p.add_argument("--tocg_checkpoint", type=str, default="")
    p.add_argument("--D_checkpoint", type=str, default="")
    p.add_argument("--tensorboard_count", type=int, default=100)
    p.add_argument("--shuffle", action="store_true")
    p.add_argument("--semantic_nc", type=int, default=13)
    p.add_argument("--output_nc", type=int, default=13)
    p.add_argument("--warp_feature", choices=["encoder", "T1"], default="T1")
    p.add_argument("--out_layer", choices=["relu", "conv"], default="relu")
    p.add_argument("--clothmask_composition", type=str, choices=["no_composition", "detach", "warp_grad"],
                   default="warp_grad")
    p.add_argument("--upsample", type=str, default="bilinear", choices=["nearest", "bilinear"])
    p.add_argument("--occlusion", action="store_true")
    p.add_argument("--Ddownx2", action="store_true")
    p.add_argument("--Ddropout", action="store_true")
    p.add_argument("--num_D", type=int, default=2)
    p.add_argument("--spectral", action="store_true")
    p.add_argument("--norm_const", type=float)
    # additions
    p.add_argument("--cuda", default=True)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--num_batches", type=int, default=2)
    p.add_argument("--ngf", type=int, default=96)
    p.add_argument("--output_dir", type=str, default="./output")
    return p.parse_args(argv)


def main(argv=None):
    opt = get_opt(argv)
    print(opt)
    print("Start to test %s!" % (opt.tocg_checkpoint or "random-init tocg"))
    dev = torch.device("cuda", 0)
    input1_nc, input2_nc = 4, opt.semantic_nc + 3
    tocg = ConditionGenerator(opt, input1_nc=input1_nc, input2_nc=input2_nc, output_nc=opt.output_nc, ngf=opt.ngf,
                              norm_layer=nn.BatchNorm2d)
    D = None
    use_d = bool(opt.D_checkpoint and os.path.exists(opt.D_checkpoint)) or (opt.synthetic and opt.norm_const is not None)
    if use_d:
        if opt.norm_const is None:
            raise NotImplementedError("--norm_const is required with a discriminator (test_condition.py:172-173)")
        D = define_D(input_nc=input1_nc + input2_nc + opt.output_nc, Ddownx2=opt.Ddownx2, Ddropout=opt.Ddropout,
                     n_layers_D=3, spectral=opt.spectral, num_D=opt.num_D)
    if opt.tocg_checkpoint:
        load_checkpoint(tocg, opt.tocg_checkpoint, opt)
    if D is not None and opt.D_checkpoint and os.path.exists(opt.D_checkpoint):
        load_checkpoint(D, opt.D_checkpoint, opt)
    tocg.to(dev).eval()
    if D is not None:
        D.to(dev).eval()
    disk = None
    if not opt.synthetic:
        from hr_viton_amd.cp_dataset import CPDataLoader, CPDatasetTest
        disk = iter(CPDataLoader(opt, CPDatasetTest(opt)).data_loader)
    parts = (opt.tocg_checkpoint or "random/tocg").split("/")
    out_dir = os.path.join(opt.output_dir, parts[-2] if len(parts) > 1 else "run", parts[-1], opt.datamode,
                           opt.datasetting, "multi-task")
    os.makedirs(out_dir, exist_ok=True)
    t0 = time.time()
    scores, num = [], 0
    for i in range(opt.num_batches if disk is None else 1 << 30):
        names = None
        if disk is None:
            batch = synthetic_batch(opt, opt.batch_size, 555 + i, dev)
        else:
            raw = next(disk, None)
            if raw is None:
                break
            ds_key = opt.datasetting
            batch = {"cloth": raw["cloth"][ds_key].to(dev), "cloth_mask": raw["cloth_mask"][ds_key].to(dev),
                     "parse_agnostic": raw["parse_agnostic"].to(dev), "densepose": raw["densepose"].to(dev),
                     "parse": raw["parse"].to(dev)}
            names = [n_.replace(".jpg", ".png") for n_ in raw["c_name"]["paired"]]
        score, misalign, fake_segmap, warped_c, warped_cm1 = rejection_scores(opt, tocg, D, batch, opt.norm_const or 1.0)
        if score is not None:
            print("prob0", score)
            for j in range(opt.batch_size):
                scores.append((names[j] if names else "synthetic_%05d.png" % (num + j), score[j].item()))
        num += opt.batch_size
        print(num)
    if D is not None:
        scores.sort(key=lambda x: x[1], reverse=True)
        with open(os.path.join(out_dir, "rejection_prob.txt"), "a") as f:
            for name, s in scores:
                f.write(name + " " + str(s) + "\n")
    print(f"Test time {time.time() - t0}")
    print("Finished testing!")


if __name__ == "__main__":
    main()
