"""VITON-HD data pipeline of the reference without torchvision (this image has PIL + numpy only):
``CPDataset`` (cp_dataset.py:13-247), ``CPDatasetTest`` (cp_dataset_test.py:12-237 = cp_dataset.py:250-401)
and ``CPDataLoader`` (cp_dataset_test.py:240-263) with the reference's constructor arguments, on-disk
layout, dictionary keys, shapes, dtypes and value ranges, so ``train_condition.py`` / ``train_generator.py``
/ ``test_generator.py`` run on a real ``{dataroot}/{datamode}/{image,cloth,cloth-mask,image-parse-v3,
image-parse-agnostic-v3.2,openpose_img,openpose_json,image-densepose}`` tree.

The torchvision calls the reference makes are restated from their documented PIL semantics
(parity with torchvision itself is unpinned offline -- it is not installed here):
    transforms.Resize(w, interpolation)   smaller edge -> w, the other int(w * long / short); PIL resample
    transforms.ToTensor()                 HWC uint8 -> CHW float32 / 255
    transforms.Normalize(0.5, 0.5)        (x - 0.5) / 0.5
Everything else (label merging, agnostic-person drawing, cloth mask threshold) is checked against the
reference's own dataset class in tests/test_cp_dataset.py (golden made by oracle/make_golden.py).

This is host-side input plumbing (CPU, PIL); the tensors it yields are moved to the GPU by the scripts.
"""
from __future__ import annotations

import json
import os.path as osp
from typing import Dict, List

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image, ImageDraw

# 20 LIP labels -> the 13 try-on classes (cp_dataset_test.py:147-161)
LABEL_GROUPS: List[List[int]] = [[0, 10], [1, 2], [4, 13], [5, 6, 7], [9, 12], [14], [15], [16], [17], [18], [19], [8],
                                 [3, 11]]
_GRAY = "gray"


def resize_to_width(img: Image.Image, size: int, interpolation: int) -> Image.Image:
    """transforms.Resize(int): the smaller edge becomes ``size`` (aspect kept, long edge truncated to int)."""
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    return img.resize((ow, oh), interpolation)


def to_normalized_tensor(img: Image.Image) -> torch.Tensor:
    """ToTensor + Normalize((.5,.5,.5),(.5,.5,.5)): uint8 HWC -> float32 CHW in [-1, 1]."""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div_(255.0)
    return t.sub_(0.5).div_(0.5)


def merge_labels(index_map: torch.Tensor, semantic_nc: int):
    """[1,H,W] int64 LIP indices -> (13-channel one-hot map, [1,H,W] float map of merged class indices)."""
    _, H, W = index_map.shape
    full = torch.zeros(20, H, W).scatter_(0, index_map, 1.0)
    merged = torch.zeros(semantic_nc, H, W)
    idx = torch.zeros(1, H, W)
    for k, group in enumerate(LABEL_GROUPS):
        for lab in group:
            merged[k] += full[lab]
            idx[0] += full[lab] * k
    return merged, idx


def _known(p) -> bool:
    return not (p[0] == 0.0 and p[1] == 0.0)


def make_agnostic(im: Image.Image, im_parse: Image.Image, pose: np.ndarray) -> Image.Image:
    """Cloth-agnostic person image (cp_dataset_test.py:47-113): torso, neck and arms painted gray from the
    OpenPose keypoints, then head, lower body and the hands (arm parse minus the drawn arm) pasted back.
    ``pose``: [25,2] keypoints in the 768x1024 source frame (modified in place like the reference does)."""
    labels = np.array(im_parse)
    head = ((labels == 4) | (labels == 13)).astype(np.float32)
    lower = np.isin(labels, (9, 12, 16, 17, 18, 19)).astype(np.float32)
    out = im.copy()
    draw = ImageDraw.Draw(out)
    shoulder_w = np.linalg.norm(pose[5] - pose[2])
    hip_w = np.linalg.norm(pose[12] - pose[9])
    mid = (pose[9] + pose[12]) / 2
    for i in (9, 12):                       # hips re-spaced to the shoulder width
        pose[i] = mid + (pose[i] - mid) / hip_w * shoulder_w
    r = int(shoulder_w / 16) + 1
    pt = lambda i: tuple(pose[i])           # noqa: E731

    def blob(d, i, rx, ry, fill):
        x, y = pose[i]
        d.ellipse((x - rx, y - ry, x + rx, y + ry), fill, fill)

    # torso
    for i in (9, 12):
        blob(draw, i, r * 3, r * 6, _GRAY)
    draw.line([pt(2), pt(9)], _GRAY, width=r * 6)
    draw.line([pt(5), pt(12)], _GRAY, width=r * 6)
    draw.line([pt(9), pt(12)], _GRAY, width=r * 12)
    draw.polygon([pt(2), pt(5), pt(12), pt(9)], _GRAY, _GRAY)
    # neck
    nx, ny = pose[1]
    draw.rectangle((nx - r * 5, ny - r * 9, nx + r * 5, ny), _GRAY, _GRAY)
    # arms
    draw.line([pt(2), pt(5)], _GRAY, width=r * 12)
    for i in (2, 5):
        blob(draw, i, r * 5, r * 6, _GRAY)
    for i in (3, 4, 6, 7):
        if not (_known(pose[i - 1]) and _known(pose[i])):
            continue
        draw.line([pt(i - 1), pt(i)], _GRAY, width=r * 10)
        blob(draw, i, r * 5, r * 5, _GRAY)
    # hands: keep the arm-parse pixels that the drawn arm does not cover
    for parse_id, chain in ((14, (5, 6, 7)), (15, (2, 3, 4))):
        keep = Image.new("L", (768, 1024), "white")
        kd = ImageDraw.Draw(keep)
        blob(kd, chain[0], r * 5, r * 6, "black")
        x, y = pose[chain[0]]
        for i in chain[1:]:
            if not (_known(pose[i - 1]) and _known(pose[i])):
                continue
            kd.line([pt(i - 1), pt(i)], "black", width=r * 10)
            x, y = pose[i]
            if i != chain[-1]:
                blob(kd, i, r * 5, r * 5, "black")
        kd.ellipse((x - r * 4, y - r * 4, x + r * 4, y + r * 4), "black", "black")
        hand = (np.array(keep) / 255) * (labels == parse_id).astype(np.float32)
        out.paste(im, None, Image.fromarray(np.uint8(hand * 255), "L"))
    out.paste(im, None, Image.fromarray(np.uint8(head * 255), "L"))
    out.paste(im, None, Image.fromarray(np.uint8(lower * 255), "L"))
    return out


class _CPBase(data.Dataset):
    """Shared loader; ``cloth_keys`` = ('paired',) for training, ('paired', 'unpaired') for testing."""
    cloth_keys = ("paired", "unpaired")

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.root = opt.dataroot
        self.datamode = opt.datamode
        self.data_list = opt.data_list
        self.fine_height, self.fine_width = opt.fine_height, opt.fine_width
        self.semantic_nc = opt.semantic_nc
        self.data_path = osp.join(opt.dataroot, opt.datamode)
        im_names, c_names = [], []
        with open(osp.join(opt.dataroot, opt.data_list), "r") as f:
            for line in f.readlines():
                a, b = line.strip().split()
                im_names.append(a)
                c_names.append(b)
        self.im_names = im_names
        self.c_names = {"paired": im_names, "unpaired": c_names}

    def name(self):
        return "CPDataset"

    def _open(self, sub: str, name: str) -> Image.Image:
        return Image.open(osp.join(self.data_path, sub, name))

    def _rs(self, img: Image.Image, interp: int) -> Image.Image:
        return resize_to_width(img, self.fine_width, interp)

    def __getitem__(self, index) -> Dict:
        im_name = self.im_names[index]
        c_name, c, cm = {}, {}, {}
        for key in self.cloth_keys:
            c_name[key] = self.c_names[key][index]
            c[key] = to_normalized_tensor(self._rs(self._open("cloth", c_name[key]).convert("RGB"), Image.BILINEAR))
            mask = np.array(self._rs(self._open("cloth-mask", c_name[key]), Image.NEAREST))
            cm[key] = torch.from_numpy((mask >= 128).astype(np.float32)).unsqueeze_(0)
        im_big = self._open("image", im_name)
        im = to_normalized_tensor(self._rs(im_big, Image.BILINEAR))
        parse_name = im_name.replace(".jpg", ".png")
        parse_big = self._open("image-parse-v3", parse_name)
        parse_idx = torch.from_numpy(np.array(self._rs(parse_big, Image.NEAREST))[None]).long()
        new_parse_map, parse_onehot = merge_labels(parse_idx, self.semantic_nc)
        agn_idx = torch.from_numpy(np.array(self._rs(self._open("image-parse-agnostic-v3.2", parse_name),
                                                      Image.NEAREST))[None]).long()
        new_parse_agnostic_map, _ = merge_labels(agn_idx, self.semantic_nc)
        pcm = new_parse_map[3:4]
        im_c = im * pcm + (1 - pcm)
        pose_map = to_normalized_tensor(self._rs(self._open("openpose_img", im_name.replace(".jpg", "_rendered.png")),
                                                 Image.BILINEAR))
        with open(osp.join(self.data_path, "openpose_json", im_name.replace(".jpg", "_keypoints.json")), "r") as f:
            kp = np.array(json.load(f)["people"][0]["pose_keypoints_2d"]).reshape((-1, 3))[:, :2]
        densepose = to_normalized_tensor(self._rs(self._open("image-densepose", im_name), Image.BILINEAR))
        agnostic = to_normalized_tensor(self._rs(make_agnostic(im_big, parse_big, kp), Image.BILINEAR))
        return {"c_name": c_name, "im_name": im_name, "cloth": c, "cloth_mask": cm,
                "parse_agnostic": new_parse_agnostic_map, "densepose": densepose, "pose": pose_map,
                "parse_onehot": parse_onehot, "parse": new_parse_map, "pcm": pcm, "parse_cloth": im_c,
                "image": im, "agnostic": agnostic}

    def __len__(self):
        return len(self.im_names)


class CPDatasetTest(_CPBase):
    """cp_dataset_test.py:12-237."""
    cloth_keys = ("paired", "unpaired")


class CPDataset(_CPBase):
    """cp_dataset.py:13-247 (training: the paired cloth only; ``im_name`` carries the reference's 'image/' prefix)."""
    cloth_keys = ("paired",)

    def __getitem__(self, index) -> Dict:
        r = super().__getitem__(index)
        r["im_name"] = "image/" + r["im_name"]
        return r


class CPDataLoader(object):
    """cp_dataset_test.py:240-263: DataLoader wrapper with ``next_batch()`` that restarts at the end."""

    def __init__(self, opt, dataset, rank: int = 0, world: int = 1):
        """``rank`` / ``world`` (data-parallel training, one process per GPU): the ranks draw DISJOINT shards of one
        shared per-epoch permutation (DistributedSampler), so no sample is seen twice within an epoch -- the reference's
        single-process loader feeds every GPU from one stream (train_generator.py:171-178)."""
        self.epoch = 0
        if world > 1:
            self.sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank,
                                                                           shuffle=True, seed=97, drop_last=True)   # (the reference's loader shuffles with and without opt.shuffle)
        else:
            self.sampler = torch.utils.data.sampler.RandomSampler(dataset) if opt.shuffle else None
        self.data_loader = torch.utils.data.DataLoader(dataset, batch_size=opt.batch_size, shuffle=(self.sampler is None),
                                                       num_workers=opt.workers, pin_memory=True, drop_last=True,
                                                       sampler=self.sampler)
        self.dataset = dataset
        self.data_iter = iter(self.data_loader)

    def next_batch(self):
        try:
            return next(self.data_iter)
        except StopIteration:
            self.epoch += 1
            if hasattr(self.sampler, "set_epoch"):
                self.sampler.set_epoch(self.epoch)          # a new shared permutation, again cut into disjoint shards
            self.data_iter = iter(self.data_loader)
            return next(self.data_iter)


# ----------------------------------------------------------------------------------------------
# synthetic VITON-HD-shaped data set on disk (no network for the real one)
# ----------------------------------------------------------------------------------------------
def write_synthetic_dataset(root: str, n: int = 4, datamode: str = "test", list_name: str = "test_pairs.txt",
                            seed: int = 0):
    """Writes ``n`` person/cloth pairs in the VITON-HD layout: 768x1024 JPEG images, mode-'P' parse PNGs with
    LIP labels 0..19 (a crude but anatomically ordered figure), OpenPose-style keypoint JSON (25 x 3), rendered
    pose / densepose images."""
    import os
    rng = np.random.RandomState(seed)
    base = osp.join(root, datamode)
    for sub in ("image", "cloth", "cloth-mask", "image-parse-v3", "image-parse-agnostic-v3.2", "openpose_img",
                "openpose_json", "image-densepose"):
        os.makedirs(osp.join(base, sub), exist_ok=True)
    W, H = 768, 1024
    names = ["%05d_00.jpg" % (i + 1) for i in range(n)]

    def smooth_rgb():
        lo = rng.randint(0, 256, size=(8, 6, 3)).astype(np.uint8)
        return Image.fromarray(lo, "RGB").resize((W, H), Image.BILINEAR)

    for i, name in enumerate(names):
        cx = W / 2 + rng.uniform(-30, 30)
        kp = np.zeros((25, 3))
        pts = {0: (cx, 150), 1: (cx, 260), 2: (cx - 130, 270), 3: (cx - 180, 430), 4: (cx - 190, 580),
               5: (cx + 130, 270), 6: (cx + 180, 430), 7: (cx + 190, 580), 8: (cx, 600), 9: (cx - 80, 610),
               10: (cx - 85, 800), 11: (cx - 85, 980), 12: (cx + 80, 610), 13: (cx + 85, 800), 14: (cx + 85, 980)}
        for k, (x, y) in pts.items():
            kp[k] = (x + rng.uniform(-8, 8), y + rng.uniform(-8, 8), 0.9)
        parse = Image.new("P", (W, H), 0)
        parse.putpalette([(j * 37) % 256 for j in range(768)])
        d = ImageDraw.Draw(parse)
        d.ellipse((cx - 70, 60, cx + 70, 140), 2)                      # hair
        d.ellipse((cx - 60, 110, cx + 60, 240), 13)                    # face
        d.rectangle((cx - 25, 235, cx + 25, 275), 10)                  # neck
        d.polygon([(cx - 140, 260), (cx + 140, 260), (cx + 110, 620), (cx - 110, 620)], 5)   # upper clothes
        d.line([(cx - 135, 275), (cx - 180, 430), (cx - 190, 590)], 15, width=60)             # right arm
        d.line([(cx + 135, 275), (cx + 180, 430), (cx + 190, 590)], 14, width=60)             # left arm
        d.polygon([(cx - 115, 615), (cx + 115, 615), (cx + 125, 820), (cx - 125, 820)], 9)    # pants
        d.line([(cx - 85, 820), (cx - 85, 960)], 17, width=70)
        d.line([(cx + 85, 820), (cx + 85, 960)], 16, width=70)
        d.ellipse((cx - 130, 950, cx - 40, 1010), 19)
        d.ellipse((cx + 40, 950, cx + 130, 1010), 18)
        parse.save(osp.join(base, "image-parse-v3", name.replace(".jpg", ".png")))
        agn = parse.copy()
        da = ImageDraw.Draw(agn)
        da.polygon([(cx - 200, 255), (cx + 200, 255), (cx + 200, 625), (cx - 200, 625)], 0)   # clothes + arms removed
        agn.save(osp.join(base, "image-parse-agnostic-v3.2", name.replace(".jpg", ".png")))
        person = smooth_rgb()
        person.save(osp.join(base, "image", name), quality=95)
        smooth_rgb().save(osp.join(base, "cloth", name), quality=95)
        cm = Image.new("L", (W, H), 0)
        ImageDraw.Draw(cm).polygon([(150, 120), (618, 120), (700, 400), (600, 950), (168, 950), (68, 400)], 255)
        cm.save(osp.join(base, "cloth-mask", name), quality=95)
        smooth_rgb().save(osp.join(base, "openpose_img", name.replace(".jpg", "_rendered.png")))
        smooth_rgb().save(osp.join(base, "image-densepose", name), quality=95)
        with open(osp.join(base, "openpose_json", name.replace(".jpg", "_keypoints.json")), "w") as f:
            json.dump({"version": 1.3, "people": [{"pose_keypoints_2d": kp.reshape(-1).tolist()}]}, f)
    with open(osp.join(root, list_name), "w") as f:
        for i, name in enumerate(names):
            f.write("%s %s\n" % (name, names[(i + 1) % n]))
    return names
