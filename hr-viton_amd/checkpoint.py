"""Checkpoint I/O with the reference's file format (torch.save'd state_dict)."""
from __future__ import annotations

import os
from collections import OrderedDict

import torch

from .networks import load_checkpoint, save_checkpoint  # noqa: F401  (networks.py:411-425)


def load_checkpoint_G(model, checkpoint_path, opt=None):
    """test_generator.py:77-86: tolerate a missing file ("Invalid path!"), apply the vestigial
    'ace'->'alias' / '.Spade'->'' key renames to keys and ``_metadata``, load strict."""
    if not os.path.exists(checkpoint_path):
        print("Invalid path!")
        return
    state_dict = torch.load(checkpoint_path, map_location="cpu")

    def ren(k):
        return k.replace("ace", "alias").replace(".Spade", "")

    new_sd = OrderedDict((ren(k), v) for k, v in state_dict.items())
    meta = getattr(state_dict, "_metadata", None)
    if meta is not None:
        new_sd._metadata = OrderedDict((ren(k), v) for k, v in meta.items())
    model.load_state_dict(new_sd, strict=True)
    if opt is not None and getattr(opt, "cuda", False):
        model.cuda()
