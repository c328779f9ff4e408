#!/usr/bin/env python3
"""Drop-in for the reference's ``test_generator.py`` (same flags, same checkpoint
files, same output files) running the MI355X-native hot path.

Differences from the reference script, all outside the hot path:
  * ``--synthetic N`` feeds N synthetic VITON-HD-shaped pairs (there is no dataset or
    torchvision in this image); without it the reference's ``cp_dataset_test`` module is
    imported from PYTHONPATH and used unchanged.
  * the per-sample tensorboard/grid visualisation (test_generator.py:221-227) is dropped;
    the try-on JPEGs (saved under a .png name, utils.py:93-109) are written the same way.
  * ``--cuda`` defaults to True: the HIP path has no CPU fallback.
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
from hr_viton_amd.checkpoint import load_checkpoint, load_checkpoint_G  # noqa: E402
from hr_viton_amd.network_generator import SPADEGenerator  # noqa: E402
from hr_viton_amd.networks import ConditionGenerator  # noqa: E402
from hr_viton_amd.pipeline import tryon_step  # noqa: E402


def get_opt(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpu_ids", default="")
    p.add_argument("-j", "--workers", type=int, default=4)
    p.add_argument("-b", "--batch-size", type=int, default=1)
    p.add_argument("--fp16", action="store_true", help="use amp")
    p.add_argument("--cuda", default=True, help="cuda or cpu (the HIP path needs cuda)")
    p.add_argument("--test_name", type=str, default="test")
    p.add_argument("--dataroot", default="./data/zalando-hd-resize")
    p.add_argument("--datamode", default="test")
    p.add_argument("--data_list", default="test_pairs.txt")
    p.add_argument("--output_dir", type=str, default="./Output")
    p.add_argument("--datasetting", default="unpaired")
    p.add_argument("--fine_width", type=int, default=768)
    p.add_argument("--fine_height", type=int, default=1024)
    p.add_argument("--tensorboard_dir", type=str, default="./data/zalando-hd-resize/tensorboard")
    p.add_argument("--checkpoint_dir", type=str, default="checkpoints")
    p.add_argument("--tocg_checkpoint", type=str, default="./eval_models/weights/v0.1/mtviton.pth")
    p.add_argument("--gen_checkpoint", type=str, default="./eval_models/weights/v0.1/gen.pth")
    p.add_argument("--tensorboard_count", type=int, default=100)
    p.add_argument("--shuffle", action="store_true")
    p.add_argument("--semantic_nc", type=int, default=13)
    p.add_argument("--output_nc", type=int, default=13)
    p.add_argument("--gen_semantic_nc", type=int, default=7)
    p.add_argument("--warp_feature", choices=["encoder", "T1"], default="T1")
    p.add_argument("--out_layer", choices=["relu", "conv"], default="relu")
    p.add_argument("--clothmask_composition", type=str, choices=["no_composition", "detach", "warp_grad"],
                   default="warp_grad")
    p.add_argument("--upsample", type=str, default="bilinear", choices=["nearest", "bilinear"])
    p.add_argument("--occlusion", action="store_true", help="Occlusion handling")
    p.add_argument("--norm_G", type=str, default="spectralaliasinstance")
    p.add_argument("--ngf", type=int, default=64)
    p.add_argument("--init_type", type=str, default="xavier")
    p.add_argument("--init_variance", type=float, default=0.02)
    p.add_argument("--num_upsampling_layers", choices=("normal", "more", "most"), default="most")
    # additions
    p.add_argument("--synthetic", type=int, default=0, help="number of synthetic VITON-HD-shaped pairs to run")
    p.add_argument("--random_init_tocg", action="store_true",
                   help="do not load --tocg_checkpoint (random-init weights; for plumbing runs)")
    p.add_argument("--tocg_ngf", type=int, default=96)
    p.add_argument("--no_save", action="store_true")
    return p.parse_args(argv)


def synthetic_batches(opt, n, seed=0):
    """Batches with the schema of CPDatasetTest.__getitem__ (cp_dataset_test.py:114-237)."""
    g = torch.Generator().manual_seed(seed)
    H, W, B = opt.fine_height, opt.fine_width, opt.batch_size
    done = 0
    while done < n:
        b = min(B, n - done)
        lab = torch.randint(0, 13, (b, 1, H // 32, W // 32), generator=g)
        lab = lab.repeat_interleave(32, 2).repeat_interleave(32, 3)
        onehot = torch.zeros(b, 13, H, W).scatter_(1, lab, 1.0)
        u = lambda c: torch.rand(b, c, H, W, generator=g) * 2 - 1  # noqa: E731
        yield {"pose": u(3), "cloth_mask": {opt.datasetting: (torch.rand(b, 1, H, W, generator=g) > 0.4).float()},
               "parse": onehot, "parse_agnostic": onehot, "agnostic": u(3), "cloth": {opt.datasetting: u(3)},
               "densepose": u(3), "image": u(3),
               "c_name": {"paired": [f"c{done + i:05d}.jpg" for i in range(b)],
                          opt.datasetting: [f"u{done + i:05d}.jpg" for i in range(b)]},
               "im_name": [f"im{done + i:05d}.jpg" for i in range(b)]}
        done += b


def save_images(img_tensors, img_names, save_dir):
    """utils.py:93-109: (x+1)*0.5*255, clamp, uint8 truncation, JPEG bytes under the given name."""
    from PIL import Image
    for t, name in zip(img_tensors, img_names):
        arr = ((t.clone() + 1) * 0.5 * 255).cpu().clamp(0, 255).numpy().astype("uint8")
        arr = arr.squeeze(0) if arr.shape[0] == 1 else arr.swapaxes(0, 1).swapaxes(1, 2)
        Image.fromarray(arr).save(os.path.join(save_dir, name), format="JPEG")


def test(opt, batches, tocg, generator):
    tocg.cuda().eval()
    generator.cuda().eval()
    output_dir = opt.output_dir or os.path.join("./output", opt.test_name, opt.datamode, opt.datasetting,
                                                "generator", "output")
    os.makedirs(output_dir, exist_ok=True)
    num = 0
    t0 = time.time()
    for inputs in batches:
        dev = {"cloth": inputs["cloth"][opt.datasetting].cuda(), "cloth_mask": inputs["cloth_mask"][opt.datasetting].cuda(),
               "parse_agnostic": inputs["parse_agnostic"].cuda(), "densepose": inputs["densepose"].cuda(),
               "agnostic": inputs["agnostic"].cuda()}
        res = tryon_step(opt, tocg, generator, dev)
        names = [inputs["c_name"]["paired"][i].split(".")[0] + "_" + inputs["c_name"][opt.datasetting][i].split(".")[0]
                 + ".png" for i in range(dev["cloth"].shape[0])]
        if not opt.no_save:
            save_images(res["output"], names, output_dir)
        num += dev["cloth"].shape[0]
        print(num)
    torch.cuda.synchronize()
    print(f"Test time {time.time() - t0}")
    return num


def main(argv=None):
    opt = get_opt(argv)
    print(opt)
    print("Start to test %s!")
    if opt.gpu_ids:
        os.environ["CUDA_VISIBLE_DEVICES"] = opt.gpu_ids
    if opt.synthetic > 0:
        batches = synthetic_batches(opt, opt.synthetic)
    else:
        # the reference's on-disk VITON-HD layout through the torchvision-free pipeline (hr_viton_amd.cp_dataset)
        from hr_viton_amd.cp_dataset import CPDataLoader, CPDatasetTest
        batches = CPDataLoader(opt, CPDatasetTest(opt)).data_loader
    tocg = ConditionGenerator(opt, input1_nc=4, input2_nc=opt.semantic_nc + 3, output_nc=opt.output_nc,
                              ngf=opt.tocg_ngf, norm_layer=nn.BatchNorm2d)
    opt.semantic_nc = 7
    generator = SPADEGenerator(opt, 3 + 3 + 3)
    generator.print_network()
    if not opt.random_init_tocg:
        load_checkpoint(tocg, opt.tocg_checkpoint, opt)
    load_checkpoint_G(generator, opt.gen_checkpoint, opt)
    test(opt, batches, tocg, generator)
    print("Finished testing!")


if __name__ == "__main__":
    main()
