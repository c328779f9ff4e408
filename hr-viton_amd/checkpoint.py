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


def save_training_state(path, modules, optimizers=None, schedulers=None, step=0, extra=None):
    """Resume-exact checkpoint (an extension -- the reference saves weights only): module
    state_dicts (same keys as the reference's .pth files), optimizer moments, LR-scheduler state, the step counter
    and the CPU / current-device RNG states, in one torch.save file."""
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    blob = {"step": int(step),
            "modules": {k: {n: t.detach().cpu() for n, t in m.state_dict().items()} for k, m in modules.items()},
            "optimizers": {k: o.state_dict() for k, o in (optimizers or {}).items()},
            "schedulers": {k: s.state_dict() for k, s in (schedulers or {}).items()},
            "rng": {"cpu": torch.get_rng_state(),
                    "cuda": torch.cuda.get_rng_state() if torch.cuda.is_available() else None},
            "extra": extra or {}}
    for k, o in blob["optimizers"].items():
        for ent in o["state"].values():
            for n in ("exp_avg", "exp_avg_sq"):
                ent[n] = ent[n].cpu()
    torch.save(blob, path)


def load_training_state(path, modules, optimizers=None, schedulers=None):
    """Inverse of save_training_state; returns (step, extra)."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    for k, m in modules.items():
        m.load_state_dict(blob["modules"][k], strict=True)
    for k, o in (optimizers or {}).items():
        o.load_state_dict(blob["optimizers"][k])
    for k, s in (schedulers or {}).items():
        s.load_state_dict(blob["schedulers"][k])
    torch.set_rng_state(blob["rng"]["cpu"])
    if blob["rng"]["cuda"] is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(blob["rng"]["cuda"])
    from . import ops
    ops.WEIGHTS_EPOCH[0] += 1       # load_state_dict's copy_ bumps every tensor's _version: cached plans rebuild
    ops.LOAD_EPOCH[0] += 1
    return blob["step"], blob["extra"]
