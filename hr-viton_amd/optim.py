"""torch.optim.Adam drop-in whose step is ONE fused HIP launch over a flat parameter buffer
(train_generator.py:154-157,322,360).  Parameters are re-pointed to views of the flat buffer on the
first step; gradients are gathered into a flat buffer (or taken from a ``GradSync``'s all-reduced
buckets).  LambdaLR and friends work unchanged (it is a torch.optim.Optimizer)."""
from __future__ import annotations


import torch

from . import ops
from . import train_ops as T


def _flat_order(ps):
    """Order of the parameters inside the flat buffers: registration order, except that a parameter carrying
    ``_hrv_flat_after = mate`` sits right behind its mate (SPADE's conv_beta behind conv_gamma: their gradients are
    produced as one [gamma ; beta] matrix, gen_train.SpadeT.backward).  The optimizer's arithmetic is element-wise:
    the order changes no value."""
    ids = {id(p) for p in ps}
    follow = {}
    for p in ps:
        mate = getattr(p, "_hrv_flat_after", None)
        if mate is not None and id(mate) in ids and mate is not p:
            follow.setdefault(id(mate), []).append(p)
    placed, out = set(), []

    def put(p):
        if id(p) in placed:
            return
        placed.add(id(p))
        out.append(p)
        for q in follow.get(id(p), ()):
            put(q)
    for p in ps:
        mate = getattr(p, "_hrv_flat_after", None)
        if mate is not None and id(mate) in ids and mate is not p and id(p) not in placed:
            put(mate)
        put(p)
    return out


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, grad_sync=None, device_step=False):
        """``device_step``: the step count and the learning rate live on the device (hrv_adam_hyper_f32 / _dev_f32), so a
        hipGraph-captured iteration (graph.GraphedTrainStep) keeps counting and follows the scheduler; every parameter must
        then receive a gradient in every iteration."""
        defaults = dict(lr=lr, betas=(float(betas[0]), float(betas[1])), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_sync = grad_sync
        self._flat = {}
        self.device_step = bool(device_step)

    def push_lr(self):
        """device_step: copy each group's current learning rate to its device scalar (call BEFORE replaying a captured
        iteration; an eager step() does it itself)."""
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is not None and "lr_dev" in st:
                st["lr_dev"].fill_(float(group["lr"]))

    def _setup(self, gi, group):
        ps = _flat_order([p for p in group["params"] if p.requires_grad])
        for p in ps:
            ops.require_cuda(p.data, "hr_viton_amd.optim.Adam parameter")
        # every parameter starts on a 16-byte boundary of the flat buffer: the conv epilogues take
        # float4 loads of bias / scale vectors (unaligned ones fall back to the scalar epilogue)
        n = sum((p.numel() + 3) // 4 * 4 for p in ps)
        dev = ps[0].device
        w = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        spans = []
        for p in ps:
            k = p.numel()
            w[off:off + k].copy_(p.data.reshape(-1))
            p.data = w[off:off + k].view_as(p.data)      # the parameter now lives in the flat buffer
            spans.append((p, off, k))
            off += (k + 3) // 4 * 4
        # ``steps``: one counter per parameter, like torch.optim.Adam's state[p]["step"] -- a parameter without a gradient
        # in some iteration (skipped below) falls behind the others, and its bias correction must use ITS count
        st = dict(w=w, m=torch.zeros_like(w), v=torch.zeros_like(w), g=torch.zeros_like(w), spans=spans, step=0,
                  steps=[0] * len(spans))
        self._flat[gi] = st
        # the backward plans write gradients straight into these slots (gen_train.grad_buffer / _acc)
        for p, off, k in spans:
            p._hrv_flat_grad = st["g"][off:off + k].view_as(p.data)
        return st

    def make_grad_sync(self, bucket_mb: float = 64.0, process_group=None, graph: bool = False):
        """Data-parallel gradient synchronisation that all-reduces contiguous slices of THIS optimizer's flat
        gradient buffer in place (parallel.GradSync): the backward plans write each gradient into its slot, the
        bucket collectives run on the buffer itself, the fused step reads it -- zero gradient copies.
        ``graph=True``: the variant for a hipGraph-captured iteration (parallel.GraphGradSync: one collective over the whole
        buffer between two graph segments)."""
        from .parallel import GradSync, GraphGradSync
        if len(self.param_groups) != 1:
            raise ops.HrvError("make_grad_sync: one parameter group per fused optimizer")
        st = self._flat.get(0) or self._setup(0, self.param_groups[0])
        if graph:
            self.grad_sync = GraphGradSync(st["g"], st["spans"], process_group, bucket_mb)
        else:
            self.grad_sync = GradSync(None, bucket_mb, process_group, flat=st["g"], spans=st["spans"])
        return self.grad_sync

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():            # like torch.optim: the closure may call backward()
                loss = closure()
        from . import train_ops as _T
        _T.wgrad_join()            # weight gradients of the last backward may sit on the side stream (train_ops.wgrad_side)
        world_scale = 1.0
        if self.grad_sync is not None:
            self.grad_sync.wait()
            world_scale = 1.0 / self.grad_sync.world
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi) or self._setup(gi, group)
            g = st["g"]
            base = g.data_ptr()
            skipped = []
            steps = st["steps"]
            for i, (p, off, k) in enumerate(st["spans"]):
                src = self.grad_sync.grad_of(p) if self.grad_sync is not None else p.grad
                if src is None:
                    # torch's Adam SKIPS a parameter without a gradient (weight, moments, weight decay and its step count
                    # untouched): the fused launch covers whole runs of the flat buffer, so the span is snapshotted here
                    # and restored below
                    g[off:off + k].zero_()
                    skipped.append((off, k, st["w"][off:off + k].clone(), st["m"][off:off + k].clone(),
                                    st["v"][off:off + k].clone()))
                else:
                    steps[i] += 1
                    if src.data_ptr() != base + 4 * off:   # already produced in place by the backward plan otherwise
                        g[off:off + k].copy_(src.reshape(-1))
            st["step"] += 1
            b1, b2 = group["betas"]
            if self.device_step:
                if skipped:
                    raise ops.HrvError("device_step Adam: every parameter needs a gradient in every iteration (one device-side "
                                       f"step count serves the whole buffer); {len(skipped)} parameter(s) had none")
                if "step_dev" not in st:
                    dev = g.device
                    st["step_dev"] = torch.full((1,), st["step"] - 1, dtype=torch.int32, device=dev)
                    st["lr_dev"] = torch.full((1,), float(group["lr"]), dtype=torch.float32, device=dev)
                    st["hyper"] = torch.zeros(3, dtype=torch.float32, device=dev)
                if not torch.cuda.is_current_stream_capturing():
                    st["lr_dev"].fill_(float(group["lr"]))       # (a captured iteration reads what push_lr() wrote)
                T.adam_step_dev(st["w"], g, st["m"], st["v"], st["step_dev"], st["lr_dev"], st["hyper"], b1, b2, group["eps"],
                                group["weight_decay"], world_scale)
                continue
            # one launch per run of consecutive spans that share a step count (a skipped span rides along with either
            # neighbour: it is restored afterwards) -- ONE launch over the whole buffer unless some parameter fell behind
            runs, n_sp = [], len(st["spans"])
            skipped_offs = {off for off, _, _, _, _ in skipped}
            i = 0
            while i < n_sp:
                if st["spans"][i][1] in skipped_offs:
                    i += 1
                    continue
                j, cnt = i, steps[i]
                while j + 1 < n_sp and (st["spans"][j + 1][1] in skipped_offs or steps[j + 1] == cnt):
                    j += 1
                a = st["spans"][i][1]
                b = st["spans"][j][1] + (st["spans"][j][2] + 3) // 4 * 4
                runs.append((a, min(b, g.numel()), cnt))
                i = j + 1
            if len(runs) == 1 and len(skipped) == 0:
                runs = [(0, g.numel(), runs[0][2])]
            for a, b, cnt in runs:
                T.adam_step(st["w"][a:b], g[a:b], st["m"][a:b], st["v"][a:b], float(group["lr"]), b1, b2, group["eps"],
                            group["weight_decay"], cnt, world_scale)
            for off, k, w0, m0, v0 in skipped:
                st["w"][off:off + k].copy_(w0)
                st["m"][off:off + k].copy_(m0)
                st["v"][off:off + k].copy_(v0)
        for group in self.param_groups:          # cached inference plans of THESE parameters are stale now
            for p in group["params"]:
                p._hrv_epoch = getattr(p, "_hrv_epoch", 0) + 1
        ops.WEIGHTS_EPOCH[0] += 1
        return loss

    # ------------------------------------------------------------------ (de)serialisation
    def state_dict(self):
        """torch.optim.Adam-compatible state_dict: per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` cut out
        of the flat moment buffers (an extension: the reference saves weights only)."""
        state, groups, idx = {}, [], 0
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            spans = {id(p): (off, k) for p, off, k in st["spans"]} if st else {}
            pos = {id(p): i for i, (p, _, _) in enumerate(st["spans"])} if st else {}
            ids = []
            dev_step = int(st["step_dev"].item()) if st and "step_dev" in st else None      # ONE host sync per group
            for p in group["params"]:
                if st and id(p) in spans:
                    off, k = spans[id(p)]
                    stp = dev_step if dev_step is not None else st["steps"][pos[id(p)]]
                    state[idx] = {"step": torch.tensor(float(stp)),
                                  "exp_avg": st["m"][off:off + k].view_as(p).clone(),
                                  "exp_avg_sq": st["v"][off:off + k].view_as(p).clone()}
                ids.append(idx)
                idx += 1
            g = {k: v for k, v in group.items() if k != "params"}
            g["params"] = ids
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        idx = 0
        for gi, (group, saved) in enumerate(zip(self.param_groups, sd["param_groups"])):
            for k, v in saved.items():
                if k != "params":
                    group[k] = tuple(v) if k == "betas" else v
            st = self._flat.get(gi) or self._setup(gi, group)
            spans = {id(p): (off, k) for p, off, k in st["spans"]}
            pos = {id(p): i for i, (p, _, _) in enumerate(st["spans"])}
            loaded = None
            for p in group["params"]:
                ent = sd["state"].get(idx)
                if ent is not None and id(p) in spans:
                    off, k = spans[id(p)]
                    st["m"][off:off + k].copy_(ent["exp_avg"].reshape(-1).to(st["m"].device))
                    st["v"][off:off + k].copy_(ent["exp_avg_sq"].reshape(-1).to(st["v"].device))
                    st["steps"][pos[id(p)]] = int(ent["step"])
                    loaded = int(ent["step"]) if loaded is None else max(loaded, int(ent["step"]))
                idx += 1
            if loaded is not None:
                st["step"] = loaded
            if "step_dev" in st:
                # device-side step count / learning rate (device_step=True after the first step): a resume must continue the
                # bias correction from the LOADED count, not from the one this optimizer had reached before the load
                st["step_dev"].fill_(int(st["step"]))
                st["lr_dev"].fill_(float(group["lr"]))
