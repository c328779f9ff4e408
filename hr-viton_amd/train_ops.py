"""Backward-pass building blocks over the C ABI (conv_bwd.hip + the forward engine):
device-side weight packing, data gradient (stride 1 and the four phases of stride 2),
weight gradient, bias gradient.  Everything is NHWC fp32 ``Act`` views like ops.py."""
from __future__ import annotations

import ctypes as C
import itertools
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib, ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, Act, _ceil4, _stream, _Timed, _workspace


# Mixed-precision training switch (the reference's --fp16 / apex-O1 role): when on, every training
# convolution (forward and data gradient) rounds its fp32 operands to bf16 while staging them and runs on
# v_mfma_f32_32x32x16_bf16 with fp32 accumulation; all tensors in HBM, the epilogues, the normalisations,
# the losses and the optimizer stay fp32.  The weight gradient has its own switch-aware kernel.
MMA_BF16 = [False]


def _bf16_tile(cout: int) -> int:
    """128-byte-row tile of the bf16 engine with the least column padding (cfg 8: 128 columns, cfg 9: 64)."""
    p64, p128 = (cout + 63) // 64 * 64, (cout + 127) // 128 * 128
    return 8 if p128 <= p64 else 9


_FROZEN_PACKS: dict = {}


def evict_serving_packs(tokens):
    """Drop the cached packed streams of a serving plan that is being replaced (network_generator.SPADEGenerator._get_plan): their
    keys carry the plan's own tokens (``frozen=("serve", tok, k)``) and can never be hit again -- without this they sat in
    _FROZEN_PACKS until its wholesale clear at 256 entries, which also drops VGG19's packs.  A captured hipGraph that
    recorded such a stream by address keeps the tensor alive itself (graph.CaptureGuard)."""
    toks = set(tokens)
    dead = [k for k in _FROZEN_PACKS
            if any(isinstance(e, tuple) and len(e) == 3 and e[0] == "serve" and e[1] in toks for e in k)]
    for k in dead:
        del _FROZEN_PACKS[k]
    return len(dead)


# ---------------------------------------------------------------------------------------------------------------
# Weight gradients on a second stream.  In a backward plan only the DATA gradients form the dependency chain; a layer's weight
# gradient hangs off it as a leaf.  At the low-resolution levels of the SPADE generator (8x6 .. 128x96 pixels, 1024 / 512 / 256
# channels: network_generator.py:188-198,224-236) a kernel launches 48-400 blocks on 256 CUs, so chain and leaves run next to
# each other instead of one after the other.  ``with wgrad_side(pixels, dy, x):`` forks (the side stream waits for what the
# current stream has enqueued so far), runs the body on the side stream, and tells the caching allocator which of the current
# stream's tensors the side stream reads; ``wgrad_join()`` -- end of every backward Function, start of the fused optimizer step,
# a gradient bucket's all-reduce -- makes the current stream wait for the side stream.  OPT-IN (HRV_WGRAD_SIDE=1): measured on the
# headline iteration (tools/ab_wgrad_side.sh, profiles/r05_ab_wgrad_side.txt) it buys nothing -- 72.1-72.7 ms off, 71.5-72.7 ms on for
# every threshold, inside the run-to-run spread: the iteration is bound by the power the chip may draw, and two under-filled kernels
# side by side cost the energy of the two one after the other.  HRV_WGRAD_SIDE_MAXPIX (default 65536) is the largest N*H*W that goes
# to the side stream.  Never inside a hipGraph capture (graph.GraphedIteration: one stream).
# ---------------------------------------------------------------------------------------------------------------
class _Side:
    streams: dict = {}      # device index -> torch.cuda.Stream
    pending: dict = {}      # device index -> True while the side stream holds work the main stream has not waited for
    active = [False]        # inside a ``with wgrad_side`` body


def wgrad_side_maxpix() -> int:
    if os.environ.get("HRV_WGRAD_SIDE", "0") != "1":
        return 0
    return int(os.environ.get("HRV_WGRAD_SIDE_MAXPIX", "65536") or 0)


class wgrad_side:
    def __init__(self, pixels: int, *reads):
        self.on = (0 < pixels <= wgrad_side_maxpix()) and not _Side.active[0] and not torch.cuda.is_current_stream_capturing()
        self.reads = reads

    def __enter__(self):
        if not self.on:
            return self
        dev = torch.cuda.current_device()
        side = _Side.streams.get(dev)
        if side is None:
            side = _Side.streams[dev] = torch.cuda.Stream(device=dev)
        self.side = side
        side.wait_stream(torch.cuda.current_stream(dev))
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        _Side.active[0] = True
        _Side.pending[dev] = True
        ops.WS_PER_STREAM[0] = True        # (the side stream's own split-K / weight-gradient scratch)
        return self

    def __exit__(self, *exc):
        if not self.on:
            return False
        _Side.active[0] = False
        ops.WS_PER_STREAM[0] = False
        self.ctx.__exit__(*exc)
        for t in self.reads:          # allocated on the main stream, read on the side stream: not to be reused before that read
            if t is not None:
                (t.t if isinstance(t, Act) else t).record_stream(self.side)
        return False


class side_region(wgrad_side):
    """``with side_region(*reads):`` -- the body runs on the device's side stream NEXT to what the caller enqueues on the current stream
    afterwards, until ``wgrad_join()``: an independent chain of small launches (the frozen condition generator's second encoder: 15
    convolutions of 6 .. 768 tiles each).  Same mechanics as wgrad_side (own split-K scratch, no nesting, never under a hipGraph
    capture) without its opt-in switch and size cap."""

    def __init__(self, *reads, on: bool = True):
        self.on = on and not _Side.active[0] and not torch.cuda.is_current_stream_capturing()
        self.reads = reads


def wgrad_join(device=None):
    """The current stream waits for the weight gradients enqueued on the side stream (no-op when there are none)."""
    if not _Side.pending:          # never forked in this process (also: a CPU-only process)
        return
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if _Side.pending.get(dev) and not _Side.active[0]:
        torch.cuda.current_stream(dev).wait_stream(_Side.streams[dev])
        _Side.pending[dev] = False


def wgrad_sync_for_collective():
    """A gradient bucket is about to be all-reduced from the current stream: its slots were written on both streams."""
    if not _Side.pending:
        return
    dev = torch.cuda.current_device()
    if not _Side.pending.get(dev):
        return
    if _Side.active[0]:
        pass          # on the side stream: the fork already ordered it behind everything the main stream had enqueued
    else:
        wgrad_join(dev)


_PACK_TOKENS = itertools.count(1)


def frozen_stamp(p) -> Optional[tuple]:
    """Stamp for pack_weight_dev(frozen=...): None for a trainable parameter.  Carries a token that is unique to the
    Parameter OBJECT: a storage address alone can be recycled by the caching allocator for another frozen network of
    the same shapes (two VGG criteria built one after the other, as consecutive tests do) and would then return the dead
    network's packed weights."""
    if p.requires_grad:
        return None
    tok = getattr(p, "_hrv_pack_token", None)
    if tok is None:
        tok = p._hrv_pack_token = next(_PACK_TOKENS)
    return (tok, p._version, getattr(p, "_hrv_epoch", 0))


class PackBatch:
    """The weight packs of one network's training plan, re-packed by ONE launch (hrv_conv2d_pack_weight_multi) when
    ``prepare_convs`` runs at the top of the plan's forward -- a training iteration re-packs ~180 weights otherwise, one
    5-11 us launch each.  The first iteration packs one by one and records what was asked for (arguments + a buffer
    that stays); later iterations find their packs here.  A record is served only while the batch is FRESH: ``run`` was
    called after the last optimizer step / weight load (ops.WEIGHTS_EPOCH / LOAD_EPOCH) -- anything else falls back to
    the single pack."""

    def __init__(self, device):
        self.device = device
        self.index: dict = {}            # key -> record number
        self.bufs, self.geoms, self.blocks, self.records, self.group = [], [], [], [], []
        self.tables = [None, None]       # per group (0: forward packs, 1: data-gradient packs): (records, first block, n, blocks)
        self.dirty = False
        self.stamp = None                # epoch of the last run()
        self.bwd_fresh = False           # ... which also packed group 1
        self.launches = 0
        self.generation = 0              # bumped by reset(): a captured hipGraph that used this batch is then stale
        self.runs = 0                    # run() calls (graph.CaptureGuard: which batches did the captured region use)
        ops.GRAPH_WATCH.add(self)

    def graph_keep(self):
        """every device tensor whose address a captured run() / conv launch of this batch holds"""
        return list(self.bufs) + [x for t in self.tables if t is not None for x in t[:2]]

    def reset(self):
        """Forget every record (the weights moved: e.g. the fused optimizer re-pointed the parameters into its flat buffer
        at its first step -- records made before that hold addresses of freed storages)."""
        self.index.clear()
        self.bufs, self.geoms, self.blocks, self.records, self.group = [], [], [], [], []
        self.tables = [None, None]
        self.dirty, self.stamp, self.bwd_fresh = False, None, False
        self.generation += 1

    @staticmethod
    def _now():
        return (ops.WEIGHTS_EPOCH[0], ops.LOAD_EPOCH[0])

    def fresh(self) -> bool:
        return self.stamp == self._now()

    def lookup(self, key):
        i = self.index.get(key)
        if i is None or self.dirty or not self.fresh() or (self.group[i] == 1 and not self.bwd_fresh):
            return None
        return self.bufs[i], self.geoms[i]

    def add(self, key, record: bytes, blocks: int, buf: torch.Tensor, geom: tuple, group: int = 0):
        self.index[key] = len(self.bufs)
        self.bufs.append(buf); self.geoms.append(geom); self.blocks.append(blocks); self.records.append(record)
        self.group.append(group)
        self.dirty = True

    def run(self, backward: bool = True):
        """(Re)pack the recorded weights from their current values -- ``backward`` False: the forward packs only (a no_grad
        forward never asks for the data-gradient forms); marks the batch fresh."""
        self.stamp = self._now()
        self.bwd_fresh = backward
        self.runs += 1
        if not self.bufs:
            return
        lib = _lib.load()
        if self.dirty:
            import numpy as np
            for g in (0, 1):
                idx = [i for i in range(len(self.bufs)) if self.group[i] == g]
                if not idx:
                    self.tables[g] = None
                    continue
                tbl = np.frombuffer(b"".join(self.records[i] for i in idx), dtype=np.uint8).copy()
                first = np.zeros(len(idx) + 1, dtype=np.int32)
                first[1:] = np.cumsum(np.asarray([self.blocks[i] for i in idx], dtype=np.int64))
                self.tables[g] = (torch.from_numpy(tbl).to(self.device), torch.from_numpy(first).to(self.device), len(idx),
                                  int(first[-1]))
            self.dirty = False
        for g in ((0, 1) if backward else (0,)):
            t = self.tables[g]
            if t is not None:
                _lib.check(lib.hrv_conv2d_pack_weight_multi(t[0].data_ptr(), t[1].data_ptr(), t[2], t[3], _stream()),
                           "hrv_conv2d_pack_weight_multi")
                self.launches += 1


PACK_BATCHING = [os.environ.get("HRV_PACK_BATCH", "1") != "0"]


def _pack_batched(batch, w, w2, pair_mode, rows_each, Cout, KH, KW, src_pad, src_real, cfg, mode, stride, pad, phase, wscale, sigma,
                  bf16, nelem):
    """-> (buf, geom) from ``batch``, the PackBatch of the plan that owns the weight (handed down by its TConv / SpadeT;
    recording the pack on its first use), or None: not batched.  (The owner is named by the caller, never looked up by
    storage address: a temporary at a recycled address must not be served another network's pack.)"""
    if batch is None or not PACK_BATCHING[0] or not batch.fresh():
        return None
    key = (w.data_ptr(), 0 if w2 is None else w2.data_ptr(), pair_mode, rows_each, Cout, KH, KW, tuple(src_pad), tuple(src_real),
           cfg, mode, stride, pad, tuple(phase), wscale, 0 if sigma is None else sigma.data_ptr(), bf16)
    hit = batch.lookup(key)
    if hit is not None:
        return hit
    if key in batch.index or torch.cuda.is_current_stream_capturing():
        return None                      # recorded earlier in this iteration (served from the next run() on) / no uploads now
    lib = _lib.load()
    n = len(src_pad)
    buf = torch.empty(nelem, dtype=torch.bfloat16 if bf16 else torch.float32, device=w.device)
    rec = C.create_string_buffer(lib.hrv_conv2d_pack_record_bytes())
    geom, blocks = (C.c_int32 * 8)(), C.c_int32(0)
    _lib.check(lib.hrv_conv2d_pack_weight_record(w.data_ptr(), Cout, KH, KW, n, (C.c_int32 * n)(*src_pad), (C.c_int32 * n)(*src_real),
                                                 cfg, mode, stride, pad, phase[0], phase[1], wscale,
                                                 None if sigma is None else sigma.data_ptr(), 1 if bf16 else 0,
                                                 None if w2 is None else w2.data_ptr(), pair_mode, rows_each, buf.data_ptr(), geom,
                                                 rec, C.byref(blocks)), "hrv_conv2d_pack_weight_record")
    assert geom[7] <= nelem, (geom[7], nelem)      # (stride-2 phases use fewer taps than the bound)
    # this first time the single-record batch IS the pack (same kernel arithmetic); from the next run() on it rides along
    one = PackBatch(w.device)
    one.add(key, rec.raw, blocks.value, buf, tuple(geom))
    one.run()
    batch.add(key, rec.raw, blocks.value, buf, tuple(geom), group=0 if mode == 0 else 1)
    return buf, tuple(geom)


def pack_weight_dev(w: torch.Tensor, src_pad: Sequence[int], src_real: Sequence[int], cfg: int, mode: int = 0,
                    stride: int = 1, pad: int = 0, phase: Tuple[int, int] = (0, 0), wscale: float = 1.0,
                    sigma: Optional[torch.Tensor] = None, bf16: bool = False, frozen=None, batch=None):
    """hrv_conv2d_pack_weight_dev_f32.  ``w``: OIHW fp32 on the device.  Returns (packed, geom)
    with geom = (KHp, KWp, pad_h, pad_w, rows, rows_pad, chunks_total, elems).
    ``frozen`` (a hashable stamp of the owning parameter, or None): the weight belongs to a network that is never
    trained on this path (VGG19 of the perceptual loss: 39 packs of 20 M parameters per training step otherwise) -- the
    packed form is kept, keyed on the storage, the stamp (``frozen_stamp``) and ops.LOAD_EPOCH."""
    key = None
    if frozen is not None and sigma is None:
        key = (w.data_ptr(), frozen, ops.LOAD_EPOCH[0], tuple(w.shape), tuple(src_pad), tuple(src_real), cfg, mode,
               stride, pad, tuple(phase), wscale, bf16)
        hit = _FROZEN_PACKS.get(key)
        if hit is not None:
            return hit
    lib = _lib.load()
    ops.require_cuda(w, "pack_weight_dev")
    w = w.contiguous()
    Cout, cin, KH, KW = w.shape
    n = len(src_pad)
    srcC = (C.c_int32 * n)(*src_pad)
    srcR = (C.c_int32 * n)(*src_real)
    bn = lib.hrv_conv2d_tile_bn(cfg)
    rows = Cout if mode == 0 else cin
    rows_pad = (rows + bn - 1) // bn * bn
    bke = lib.hrv_conv2d_tile_row_bytes(cfg) // 2 if bf16 else 16      # k-values per packed row (the tile's row size)
    chunks = sum((c + bke - 1) // bke for c in src_pad) if mode == 0 else (_ceil4(Cout) + bke - 1) // bke
    if key is None:
        hit = _pack_batched(batch, w, None, 0, 0, Cout, KH, KW, src_pad, src_real, cfg, mode, stride, pad, phase, wscale, sigma, bf16,
                            KH * KW * chunks * rows_pad * bke)
        if hit is not None:
            return hit
    buf = torch.empty(KH * KW * chunks * rows_pad * bke, dtype=torch.bfloat16 if bf16 else torch.float32, device=w.device)
    geom = (C.c_int32 * 8)()
    fn = lib.hrv_conv2d_pack_weight_dev_bf16 if bf16 else lib.hrv_conv2d_pack_weight_dev_f32
    _lib.check(fn(w.data_ptr(), Cout, KH, KW, n, srcC, srcR, cfg, mode, stride, pad, phase[0], phase[1], wscale,
                  None if sigma is None else sigma.data_ptr(), buf.data_ptr(), geom, _stream()),
               "hrv_conv2d_pack_weight_dev_" + ("bf16" if bf16 else "f32"))
    if key is not None:
        if len(_FROZEN_PACKS) > 256:
            _FROZEN_PACKS.clear()
        _FROZEN_PACKS[key] = (buf, tuple(geom))
    return buf, tuple(geom)


def pack_weight_pair_dev(w_a: torch.Tensor, w_b: torch.Tensor, pair_mode: int, src_pad: Sequence[int], src_real: Sequence[int],
                         cfg: int, mode: int, pad: int, bf16: bool, batch=None):
    """hrv_conv2d_pack_weight_pair_dev: (conv_gamma.weight, conv_beta.weight) packed as ONE matrix without a
    concatenated copy.  pair_mode 1 (forward): rows interleaved (gamma32 | beta32); 2 (data gradient over
    [dgamma | dbeta]).  Returns (packed, geom, virtual Cout)."""
    lib = _lib.load()
    ops.require_cuda(w_a, "pack_weight_pair_dev")
    assert w_a.is_contiguous() and w_b.is_contiguous() and w_a.shape == w_b.shape
    rows_each, cin, KH, KW = w_a.shape
    Cout = (rows_each + 31) // 32 * 64 if pair_mode == 1 else 2 * rows_each
    assert pair_mode == 1 or rows_each % 4 == 0, "the [dgamma | dbeta] halves must be dense (C % 4 == 0)"
    n = len(src_pad)
    srcC = (C.c_int32 * n)(*src_pad)
    srcR = (C.c_int32 * n)(*src_real)
    bn = lib.hrv_conv2d_tile_bn(cfg)
    rows = Cout if mode == 0 else cin
    rows_pad = (rows + bn - 1) // bn * bn
    bke = lib.hrv_conv2d_tile_row_bytes(cfg) // 2 if bf16 else 16
    chunks = sum((c + bke - 1) // bke for c in src_pad) if mode == 0 else (_ceil4(Cout) + bke - 1) // bke
    hit = _pack_batched(batch, w_a, w_b, pair_mode, rows_each, Cout, KH, KW, src_pad, src_real, cfg, mode, 1, pad, (0, 0), 1.0, None, bf16,
                        KH * KW * chunks * rows_pad * bke)
    if hit is not None:
        return hit[0], hit[1], Cout
    buf = torch.empty(KH * KW * chunks * rows_pad * bke, dtype=torch.bfloat16 if bf16 else torch.float32, device=w_a.device)
    geom = (C.c_int32 * 8)()
    _lib.check(lib.hrv_conv2d_pack_weight_pair_dev(w_a.data_ptr(), w_b.data_ptr(), rows_each, pair_mode, Cout, KH, KW, n, srcC,
                                                   srcR, cfg, mode, pad, 1 if bf16 else 0, buf.data_ptr(), geom, _stream()),
               "hrv_conv2d_pack_weight_pair_dev")
    return buf, tuple(geom), Cout


def spade_vec_prep(gamma_bias: torch.Tensor, beta_bias: torch.Tensor, noise_scale: torch.Tensor):
    """-> (bias of the fused gamma|beta conv in its interleaved column order, noise scale padded to ceil4(C))."""
    lib = _lib.load()
    C_ = gamma_bias.numel()
    bc = torch.empty((C_ + 31) // 32 * 64, dtype=torch.float32, device=gamma_bias.device)
    ns = torch.empty(_ceil4(C_), dtype=torch.float32, device=gamma_bias.device)
    _lib.check(lib.hrv_spade_vec_prep_f32(gamma_bias.data_ptr(), beta_bias.data_ptr(), noise_scale.data_ptr(), C_,
                                          bc.data_ptr(), ns.data_ptr(), _stream()), "hrv_spade_vec_prep_f32")
    return bc, ns


def spade_vec_prep_multi(items) -> list:
    """spade_vec_prep of every (gamma_bias, beta_bias, noise_scale) triple of ``items`` in ONE launch per 32 norms
    (hrv_spade_vec_prep_multi_f32); returns [(bias interleaved, noise scale padded)] as 16-byte-aligned views of two buffers."""
    lib = _lib.load()
    if not items:
        return []
    dev = items[0][0].device
    Cs = [g.numel() for g, _, _ in items]
    nb = [(c + 31) // 32 * 64 for c in Cs]
    nn_ = [_ceil4(c) for c in Cs]
    ob = [0] + list(itertools.accumulate(nb))
    on = [0] + list(itertools.accumulate(nn_))
    bc_all = torch.empty(ob[-1], dtype=torch.float32, device=dev)
    ns_all = torch.empty(on[-1], dtype=torch.float32, device=dev)
    for i0 in range(0, len(items), 32):
        ch = items[i0:i0 + 32]
        n = len(ch)
        P = C.c_void_p * n
        I = C.c_int32 * n
        _lib.check(lib.hrv_spade_vec_prep_multi_f32(n, P(*[g.data_ptr() for g, _, _ in ch]), P(*[b.data_ptr() for _, b, _ in ch]),
                                                    P(*[s_.data_ptr() for _, _, s_ in ch]), I(*Cs[i0:i0 + n]), I(*ob[i0:i0 + n]),
                                                    I(*on[i0:i0 + n]), bc_all.data_ptr(), ns_all.data_ptr(), _stream()),
                   "hrv_spade_vec_prep_multi_f32")
    return [(bc_all[ob[i]:ob[i + 1]], ns_all[on[i]:on[i + 1]]) for i in range(len(items))]


def shared_taps_prep(ws: Sequence[torch.Tensor], bs: Sequence[torch.Tensor], cp: int):
    """conv_shared [hid, c, 3, 3] x n -> one 1x1 weight [n*hid, 9*cp, 1, 1] over the tap-expanded label map, + bias."""
    lib = _lib.load()
    n = len(ws)
    hid, c = ws[0].shape[0], ws[0].shape[1]
    assert all(w.is_contiguous() and tuple(w.shape) == (hid, c, 3, 3) for w in ws)
    wt = torch.empty((n * hid, 9 * cp, 1, 1), dtype=torch.float32, device=ws[0].device)
    bt = torch.empty(n * hid, dtype=torch.float32, device=ws[0].device)
    wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
    bp = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
    _lib.check(lib.hrv_shared_taps_prep_f32(wp, bp, n, hid, c, cp, wt.data_ptr(), bt.data_ptr(), _stream()),
               "hrv_shared_taps_prep_f32")
    return wt, bt


def shared_taps_grad(dw: torch.Tensor, db: torch.Tensor, gws: Sequence[torch.Tensor], gbs: Sequence[torch.Tensor], cp: int):
    """Inverse of shared_taps_prep for the gradients: writes every gws[i] [hid, c, 3, 3] and gbs[i] [hid]."""
    lib = _lib.load()
    n = len(gws)
    hid, c = gws[0].shape[0], gws[0].shape[1]
    assert all(g.is_contiguous() for g in gws) and all(g.is_contiguous() for g in gbs)
    gp = (C.c_void_p * n)(*[g.data_ptr() for g in gws])
    bp = (C.c_void_p * n)(*[g.data_ptr() for g in gbs])
    _lib.check(lib.hrv_shared_taps_grad_f32(dw.data_ptr(), db.data_ptr(), n, hid, c, cp, gp, bp, _stream()),
               "hrv_shared_taps_grad_f32")


def _run_engine(srcs, w_packed, Cout, cfg, N, H, W, Ho, Wo, KH, KW, stride, pad_h, pad_w, out: Act, scale=None,
                shift=None, residual: Optional[Act] = None, res_mode: int = 0, act: int = ACT_NONE, slope: float = 0.2,
                free_extent: int = 0, out_step: int = 0, out_off=(0, 0), out_hw=(0, 0), out_up: int = 0,
                name: str = "conv", flops: float = 0.0, mma_bf16: bool = False, spade=None):
    """Raw launch of hrv_conv2d_nhwc_f32 (or, with ``mma_bf16``, of the bf16 matrix-core engine over the same
    fp32 tensors: mixed_flags 15) with an explicit, already packed weight."""
    lib = _lib.load()
    d = _lib.hrv_conv2d_t()
    d.N, d.H, d.W, d.Ho, d.Wo = N, H, W, Ho, Wo
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad_h
    d.pad_w_plus1 = pad_w + 1
    d.nsrc = len(srcs)
    src_bf16 = srcs[0][0].bf16
    for i, (a, up, creal) in enumerate(srcs):
        s = d.src[i]
        assert a.bf16 == src_bf16, f"{name}: the sources of one convolution share one storage type"
        s.ptr, s.C, s.cstride, s.coff, s.up_shift, s.pre_act, s.C_real = (a.t.data_ptr(), ops._cpad(creal, a.bf16),
                                                                          a.cstride, a.coff, up, 0, creal)
    d.w_packed = w_packed.data_ptr()
    d.Cout, d.tile_cfg = Cout, cfg
    d.scale = None if scale is None else scale.data_ptr()
    d.shift = None if shift is None else shift.data_ptr()
    if residual is not None:
        d.residual, d.res_cstride, d.res_coff = residual.t.data_ptr(), residual.cstride, residual.coff
    d.res_mode = res_mode
    d.act, d.act_slope = act, slope
    d.out, d.out_cstride, d.out_coff = out.t.data_ptr(), out.cstride, out.coff
    d.out_up_shift = out_up
    d.free_extent, d.out_step = free_extent, out_step
    d.out_off_h, d.out_off_w, d.out_H, d.out_W = out_off[0], out_off[1], out_hw[0], out_hw[1]
    need = lib.hrv_conv2d_workspace_bytes(C.byref(d))
    if need > 0:
        ws = _workspace(out.t.device, need)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if spade is not None:
        d.spade = C.pointer(spade)
    if mma_bf16:
        # storage types of the tensors around the bf16 matrix cores: fp32 unless a tensor is one that only
        # matrix cores read (kept in bf16: same operand bits, half the bytes, LDS-DMA staging)
        d.mixed_flags = ((0 if out.bf16 else 1) | (0 if (residual is not None and residual.bf16) else 2) | 4 |
                         (0 if src_bf16 else 8))
    else:
        assert not (src_bf16 or out.bf16 or (residual is not None and residual.bf16)), f"{name}: bf16 tensors need MMA_BF16"
    nbytes = (sum(ops.act_bytes(a, creal) for a, _, creal in srcs) + ops.act_bytes(out, Cout) +       # (``out`` already has the upsampled extent when out_up is set)
             
              ops.act_bytes(residual, Cout) + float(Cout) * sum(c for _, _, c in srcs) * KH * KW * (2 if mma_bf16 else 4))
    with _Timed("conv", name, flops, nbytes, f"conv_mfma_kernel[tile {cfg}]"):
        fn = lib.hrv_conv2d_nhwc_bf16 if mma_bf16 else lib.hrv_conv2d_nhwc_f32
        _lib.check(fn(C.byref(d), _stream()), f"hrv_conv2d_nhwc_{'bf16' if mma_bf16 else 'f32'}[{name}]")
    return out


def _thin_ok(a: Act, KH: int, KW: int, stride: int, pad: int, cols: int, N: int, H: int, W: int) -> bool:
    """thin_conv.hip serves this layer: mixed precision, ONE bf16-stored source read at its own resolution, 3x3 / 1x1
    stride-1 'same', <= 96 channels on either side (the 1024x768 level), enough pixels for a persistent grid."""
    import os
    return (MMA_BF16[0] and a.bf16 and stride == 1 and KH == KW and pad == KH // 2 and N * H * W >= 65536 and
            os.environ.get("HRV_THIN_CONV", "1") != "0" and
            bool(_lib.load().hrv_thin_conv_supported(KH, KW, a.Cp, cols)))


def _thin_conv(src: Act, w: torch.Tensor, mode: int, sigma, wscale: float, shift, residual: Optional[Act], res_mode: int,
               act: int, slope: float, out: Act, name: str, flops: float):
    lib = _lib.load()
    d = _lib.hrv_thin_conv_t()
    d.src, d.N, d.H, d.W = src.t.data_ptr(), src.N, src.H, src.W
    d.src_channels, d.src_cstride, d.src_coff = src.Cp, src.cstride, src.coff
    d.w_oihw, d.Cout, d.Cin, d.KH, d.KW = w.data_ptr(), w.shape[0], w.shape[1], w.shape[2], w.shape[3]
    d.sigma = None if sigma is None else sigma.data_ptr()
    d.wscale, d.mode = wscale, mode
    d.shift = None if shift is None else shift.data_ptr()
    if residual is not None:
        d.residual, d.res_cstride, d.res_coff = residual.t.data_ptr(), residual.cstride, residual.coff
        d.res_bf16 = 1 if residual.bf16 else 0
    d.res_mode, d.act, d.act_slope = res_mode, act, slope
    d.out, d.out_cstride, d.out_coff, d.out_bf16 = out.t.data_ptr(), out.cstride, out.coff, 1 if out.bf16 else 0
    nbytes = ops.act_bytes(src) + ops.act_bytes(out) + ops.act_bytes(residual, out.C) + 4.0 * w.numel()
    with _Timed("conv", name, flops, nbytes, "thin_conv_kernel"):
        _lib.check(lib.hrv_thin_conv_bf16(C.byref(d), _stream()), f"hrv_thin_conv_bf16[{name}]")
    return out


def _cout1_ok(w: torch.Tensor, x: Act, stride: int, pad: int, part: str = "fwd") -> bool:
    """conv_cout1.hip serves this layer: ONE output channel, K <= 4, stride 1, pad >= (K-1)/2, an fp32 source with 4-channel
    granules (PatchGAN's last convolution).  HRV_CONV_COUT1: "0" off, "fwd" the forward kernel only, default all three.
    Measured at 2 x 4 x 131 x 99 x 256: forward 0.233 -> 0.109 ms; the first data- / weight-gradient kernels (16 global dY
    loads per pixel) were no faster than the padded matrix-core path (0.099 / 0.72 ms against 0.094 / 0.19) and were
    rewritten with the dY rows of an input row staged in LDS."""
    Cout, cin, KH, KW = w.shape
    mode = os.environ.get("HRV_CONV_COUT1", "1")
    return (Cout == 1 and KH == KW and KH <= 4 and stride == 1 and 0 <= pad < KH and 2 * pad >= KH - 1 and not x.bf16 and
            x.C == cin and cin % 4 == 0 and cin <= 2048 and x.cstride % 4 == 0 and x.coff % 4 == 0 and w.is_contiguous() and
            mode != "0" and (mode != "fwd" or part == "fwd"))


def _cout1_desc(w, x: Act, pad: int, wscale: float, sigma, y: Act):
    d = _lib.hrv_conv_cout1_t()
    d.x, d.N, d.H, d.W, d.C, d.x_cstride, d.x_coff = x.t.data_ptr(), x.N, x.H, x.W, x.C, x.cstride, x.coff
    d.w_oihw, d.wscale, d.K, d.pad = w.data_ptr(), wscale, w.shape[2], pad
    d.sigma = None if sigma is None else sigma.data_ptr()
    d.y, d.y_cstride, d.y_coff = y.t.data_ptr(), y.cstride, y.coff
    d.round_bf16 = 1 if MMA_BF16[0] else 0
    return d


_ALT_F32_TILE = {0: 7, 6: 4, 1: 2, 5: 3}      # 128-row fp32 tiles -> the 256-row tile of the same width


def _f32_tile(M: int, cout: int) -> int:
    """fp32-engine tile of a training convolution: hrv_conv2d_pick_tile, or -- HRV_CONV_TILE_TRAIN=bm256, a TEST knob -- the
    256-row tile of the same width (another block shape, wave layout and split-K geometry for the same convolution: the
    at-size self-consistency check of tests/test_gpu_fullsize_tocg.py)."""
    cfg = _lib.load().hrv_conv2d_pick_tile(M, cout)
    if os.environ.get("HRV_CONV_TILE_TRAIN") == "bm256":
        cfg = _ALT_F32_TILE.get(cfg, cfg)
    return cfg


def _p2_fwd_ok(w: torch.Tensor, a0: Act, stride: int, pad: int, out: Optional[Act], out_is_bf16: bool, residual, act: int) -> bool:
    """csrc/conv_p2.hip serves this forward layer: 3x3 stride-1 'same' over ONE bf16-stored source read at its own resolution."""
    Cout, cin, KH, KW = w.shape
    oal = 8 if out_is_bf16 else 4
    # a K that is not a multiple of 16 or a column count off the 16-byte store granule: only where it was measured to win -- a
    # 3-channel image into >= 64 columns (VGG19 features.0: the thin kernel's tile loop is latency-bound there)
    odd = cin % 16 != 0 or cin < 32 or Cout % oal != 0
    return ((KH, KW, stride, pad) == (3, 3, 1, 1) and act in (ACT_NONE, ACT_RELU, ACT_LRELU) and a0.bf16 and a0.C == cin and
            (not odd or (Cout >= 64 and Cout % oal == 0 and os.environ.get("HRV_CONV_P2_ODD", "1") != "0")) and
            a0.cstride % 8 == 0 and a0.coff % 8 == 0 and a0.coff + (cin + 7) // 8 * 8 <= a0.cstride and w.is_contiguous() and
            (out is None or (out.cstride % oal == 0 and out.coff % oal == 0)) and
            (residual is None or (type(residual) is Act and residual.C == Cout and residual.cstride % 4 == 0 and residual.coff % 4 == 0 and
                                  residual.t.data_ptr() % 16 == 0)) and
            conv_p2_ok(cin, Cout, a0.N, a0.H, a0.W))


def conv_forward_fast(w: torch.Tensor, a0: Act, pad: int, wscale: float, shift: Optional[torch.Tensor], residual: Optional[Act], act: int,
                      slope: float, out: Optional[Act], out_bf16: bool, name: str, frozen) -> Optional[Act]:
    """bf16 SERVING plans (network_generator): a stride-1 layer over one bf16-stored source on conv_p2.hip or thin_conv.hip when one
    of them serves it -- the kernels of the training forward, same operand rounding as the plan's own ConvLayer (bf16 operands, fp32
    accumulate) -- else None and the caller's ConvLayer runs.  ``frozen``: any hashable that identifies the weight version (the
    packed stream is cached under it)."""
    if os.environ.get("HRV_SERVE_FAST", "1") == "0" or not a0.bf16:
        return None
    Cout, cin, KH, KW = w.shape
    prev = MMA_BF16[0]
    MMA_BF16[0] = True
    try:
        bf = out.bf16 if out is not None else (out_bf16 and Cout % 8 == 0)
        if not (_p2_fwd_ok(w, a0, 1, pad, out, bf, residual, act) or
                (w.is_contiguous() and a0.C == cin and _thin_ok(a0, KH, KW, 1, pad, Cout, a0.N, a0.H, a0.W) and
                 (out is None or out.cstride % 4 == 0))):
            return None
        return conv_forward_dev(w, [(a0, 0)], 1, pad, wscale=wscale, shift=shift, residual=residual, act=act, slope=slope, out=out,
                                name=name, out_bf16=out_bf16, frozen=frozen)
    finally:
        MMA_BF16[0] = prev


def conv_forward_dev(w: torch.Tensor, srcs: Sequence[Tuple[Act, int]], stride: int, pad: int, wscale: float = 1.0,
                     sigma: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, residual: Optional[Act] = None, act: int = ACT_NONE,
                     slope: float = 0.2, out: Optional[Act] = None, out_up: int = 0, name: str = "conv",
                     out_bf16: bool = False, frozen=None, batch=None) -> Act:
    """Forward convolution with device-resident, per-step packed weights.  ``srcs``: (Act, up_shift).
    ``out_bf16`` (mixed precision only): store the result in bf16 -- for tensors that only matrix cores read."""
    lib = _lib.load()
    Cout, cin, KH, KW = w.shape
    a0, up0 = srcs[0]
    N = a0.N
    H, W = (a0.H << up0, a0.W << up0) if up0 >= 0 else (a0.H >> -up0, a0.W >> -up0)
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    mb = MMA_BF16[0]
    if (len(srcs) == 1 and up0 == 0 and out_up == 0 and residual is None and act == ACT_NONE and out is None and
            _cout1_ok(w, a0, stride, pad)):
        out = ops.alloc(N, Ho, Wo, 1, a0.t.device)          # (pad channels 1..3 are zero)
        d = _cout1_desc(w, a0, pad, wscale, sigma, out)
        d.bias = None if shift is None else shift.data_ptr()
        with _Timed("conv", name, 2.0 * N * Ho * Wo * cin * KH * KW, ops.act_bytes(a0) + 4.0 * N * Ho * Wo, "cout1_kernel"):
            _lib.check(lib.hrv_conv_cout1_fwd_f32(C.byref(d), _stream()), f"hrv_conv_cout1_fwd_f32[{name}]")
        return out
    p2_bf = out.bf16 if out is not None else (out_bf16 and Cout % 8 == 0)
    p2 = mb and len(srcs) == 1 and up0 == 0 and out_up == 0 and _p2_fwd_ok(w, a0, stride, pad, out, p2_bf, residual, act)
    if p2 and os.environ.get("HRV_CONV_P2_WIDE", "1") == "0":      # A/B: the kernel's round-4 mid range (64-column multiples, thin kernel first)
        p2 = (Cout % 64 == 0 and residual is None and
              not _thin_ok(a0, KH, KW, stride, pad, Cout, N, H, W))
    if (not p2 and len(srcs) == 1 and up0 == 0 and out_up == 0 and w.is_contiguous() and
            _thin_ok(a0, KH, KW, stride, pad, Cout, N, H, W) and (out is None or out.cstride % 4 == 0)):
        assert a0.C == cin, (name, a0.C, cin)
        if out is None:
            out = ops.alloc(N, Ho, Wo, Cout, a0.t.device, bf16=out_bf16 and Cout % 4 == 0)
        return _thin_conv(a0, w, 0, sigma, wscale, shift, residual, 0, act, slope, out, name,
                          2.0 * N * Ho * Wo * Cout * cin * KH * KW)
    if p2:
        # plain 3x3 over one bf16 source: the two-blocks-per-CU kernel (VGG19's 128..512-channel layers; SPADEResBlock.conv_0 of
        # up_2 / up_3: 272 -> 128, 144 -> 64)
        if out is None:
            out = ops.alloc(N, Ho, Wo, Cout, a0.t.device, bf16=p2_bf)
        pk = conv_p2_pack(0, w, None, cin, Cout, sigma, wscale, frozen)
        return conv_p2(a0, pk, Cout, out, bias=shift, act=act, slope=slope, residual=residual, name=name,
                       flops=2.0 * N * Ho * Wo * Cout * cin * 9)
    cfg = _bf16_tile(Cout) if mb else _f32_tile(N * Ho * Wo, Cout)
    if mb:                 # bf16-stored source: the halo patch stays in LDS (ops.patch_tile)
        if KH == 1 and KW == 1 and a0.bf16 and len(srcs) == 1 and a0.Cp <= 128 and Cout % 64 == 0:
            cfg = 6        # one or two K-tiles: the small tile with 64-byte rows keeps more blocks resident
        cfg = ops.patch_tile(a0.bf16, KH, KW, stride, pad, len(srcs), up0, a0.Cp, Cout, N, H, W) or cfg
    real = [a.C for a, _ in srcs]
    assert sum(real) == cin, (name, real, cin)
    packed, _ = pack_weight_dev(w, [a.Cp for a, _ in srcs], real, cfg, 0, stride, pad, wscale=wscale, sigma=sigma,
                                bf16=mb, frozen=frozen, batch=batch)
    if out is None:
        out = ops.alloc(N, Ho << out_up, Wo << out_up, Cout, a0.t.device, bf16=out_bf16 and mb)
    fl = 2.0 * N * Ho * Wo * Cout * cin * KH * KW
    return _run_engine([(a, up, a.C) for a, up in srcs], packed, Cout, cfg, N, H, W, Ho, Wo, KH, KW, stride, pad, pad,
                       out, shift=shift, residual=residual, act=act, slope=slope, out_up=out_up, name=name, flops=fl,
                       mma_bf16=mb)


def conv_dgrad(dy: Act, w, H: int, W: int, stride: int, pad: int, wscale: float = 1.0,
               sigma: Optional[torch.Tensor] = None, act_mask: Optional[Act] = None, slope: float = 0.2, out: Optional[Act] = None,
               name: str = "dgrad", out_bf16: bool = False, frozen=None, add: Optional[Act] = None, batch=None,
               add_after: Optional[Act] = None) -> Act:
    """dX [N,H,W,Cin] of y = conv(x, w*wscale) given dY ([N,Ho,Wo,Cout]); optionally multiplied by the
    activation derivative of ``act_mask`` (x = act(pre) with act = ReLU/LeakyReLU: mask tensor = x).
    ``w`` may be a PAIR (w_gamma, w_beta) for dY = [dgamma | dbeta] (stride 1): packed without a concatenated copy.
    ``add`` (instead of ``act_mask``): a second gradient of the same tensor, summed in the epilogue (the feature-matching
    tap gradient of a PatchGAN feature joins the gradient flowing down through it: no separate accumulation pass).
    ``add_after`` (with or without ``act_mask``): a gradient w.r.t. the same PRE-activation, added behind the mask (VGG19's tap
    gradients, which carry their ReLU derivative already) -- in the epilogue of csrc/conv_p2.hip where that kernel serves the layer,
    else by a separate add_slice pass."""
    lib = _lib.load()
    if add_after is not None:
        assert add is None and (add_after.N, add_after.H, add_after.W) == (dy.N, H, W)
    assert add is None or act_mask is None, "conv_dgrad: one residual slot (mask or addend)"
    pair = w if isinstance(w, (tuple, list)) else None
    if pair is not None:
        assert stride == 1 and sigma is None and wscale == 1.0
        w = pair[0]
        Cout, cin, KH, KW = 2 * w.shape[0], w.shape[1], w.shape[2], w.shape[3]
    else:
        Cout, cin, KH, KW = w.shape
    N, Ho, Wo = dy.N, dy.H, dy.W
    assert dy.C == Cout
    mb = MMA_BF16[0]
    if out is None:
        out = ops.alloc(N, H, W, cin, dy.t.device, bf16=out_bf16 and mb and cin % 8 == 0 and stride == 1)
    cfg = _bf16_tile(cin) if mb else _f32_tile(N * H * W, cin)
    res_mode = 1 if act_mask is not None else 0
    if add is not None:
        act_mask = add           # (the engine's residual slot: res_mode 0 adds it)
    fl = 2.0 * N * Ho * Wo * Cout * cin * KH * KW
    if (pair is None and res_mode == 0 and not dy.bf16 and not out.bf16 and (add is None or not add.bf16) and
            (Ho, Wo) == (H + 2 * pad - KH + 1, W + 2 * pad - KW + 1) and _cout1_ok(w, Act(out.t, cin, out.coff), stride, pad, "dgrad")):
        d = _cout1_desc(w, Act(out.t, cin, out.coff), pad, wscale, sigma, dy)      # (x slot: geometry only)
        d.dx, d.dx_cstride, d.dx_coff = out.t.data_ptr(), out.cstride, out.coff
        if add is not None:
            d.add, d.add_cstride, d.add_coff = add.t.data_ptr(), add.cstride, add.coff
        with _Timed("conv", name, fl, ops.act_bytes(out) * (2 if add is not None else 1) + 4.0 * N * Ho * Wo, "cout1_kernel"):
            _lib.check(lib.hrv_conv_cout1_dgrad_f32(C.byref(d), _stream()), f"hrv_conv_cout1_dgrad_f32[{name}]")
        return out
    oal = 8 if out.bf16 else 4
    # (columns off the 16-byte store granule: only the >= 64-channel gradient into a 3-channel image, VGG19 features.0 -- the padded
    #  lanes of `out` receive zeros)
    p2 = (mb and stride == 1 and (KH, KW, pad) == (3, 3, 1) and (Ho, Wo) == (H, W) and dy.bf16 and add is None and dy.C == Cout and
          Cout % 16 == 0 and dy.cstride % 8 == 0 and dy.coff % 8 == 0 and
          (cin % oal == 0 or (pair is None and Cout >= 64 and out.coff + (cin + oal - 1) // oal * oal <= out.cstride and
                              os.environ.get("HRV_CONV_P2_ODD", "1") != "0")) and
          out.cstride % oal == 0 and out.coff % oal == 0 and
          (act_mask is None or (act_mask.bf16 and act_mask.C == cin and act_mask.cstride % 4 == 0 and act_mask.coff % 4 == 0)) and
          w.is_contiguous() and conv_p2_ok(Cout, cin, N, H, W))
    if p2 and os.environ.get("HRV_CONV_P2_WIDE", "1") == "0":
        p2 = Cout % 32 == 0 and cin % 64 == 0 and not (pair is None and _thin_ok(dy, KH, KW, 1, pad, cin, N, H, W))
    if add_after is not None:
        ride = (p2 and add_after.C == cin and add_after.cstride % 4 == 0 and add_after.coff % 4 == 0 and add_after.t.data_ptr() % 16 == 0 and
                add_after.coff + (cin + 3) // 4 * 4 <= add_after.cstride and os.environ.get("HRV_DGRAD_ADD_AFTER", "1") != "0")
        if not ride:
            out = conv_dgrad(dy, pair if pair is not None else w, H, W, stride, pad, wscale, sigma,
                             act_mask if res_mode == 1 else None, slope, out, name, out_bf16, frozen, None, batch)
            add_slice(add_after, out, True)
            return out
    if (not p2 and add is None and pair is None and stride == 1 and (Ho, Wo) == (H, W) and w.is_contiguous() and out.cstride % 4 == 0 and
            _thin_ok(dy, KH, KW, 1, pad, cin, N, H, W)):
        return _thin_conv(dy, w, 1, sigma, wscale, None, act_mask, res_mode, ACT_NONE, slope, out, name, fl)
    if p2:
        # a stride-1 data gradient is a 'same' 3x3 convolution over dY: the two-blocks-per-CU kernel
        pk = (conv_p2_pack(2, pair[0], pair[1], Cout, cin) if pair is not None else
              conv_p2_pack(1, w, None, Cout, cin, sigma, wscale, frozen))
        return conv_p2(dy, pk, cin, out, mask=act_mask if res_mode == 1 else None, mask_slope=slope, name=name, flops=fl,
                       residual=add_after, res_after_mask=add_after is not None)
    if stride == 1:
        if mb and (Ho, Wo) == (H, W):   # a stride-1 data gradient is a 'same' 3x3 convolution over dY
            cfg = ops.patch_tile(dy.bf16, KH, KW, 1, KH - 1 - pad, 1, 0, dy.Cp, cin, N, H, W) or cfg
        if pair is not None:
            packed, g, _ = pack_weight_pair_dev(pair[0], pair[1], 2, [_ceil4(cin)], [cin], cfg, 1, pad, mb, batch=batch)
        else:
            packed, g = pack_weight_dev(w, [_ceil4(cin)], [cin], cfg, 1, 1, pad, wscale=wscale, sigma=sigma, bf16=mb,
                                        frozen=frozen, batch=batch)
        _run_engine([(dy, 0, Cout)], packed, cin, cfg, N, Ho, Wo, H, W, g[0], g[1], 1, g[2], g[3], out,
                    residual=act_mask, res_mode=res_mode, slope=slope, name=name, flops=fl, mma_bf16=mb)
        return out
    assert stride == 2, "data gradient implemented for stride 1 and 2"
    for a in range(2):
        for b in range(2):
            Hp, Wp = (H - a + 1) // 2, (W - b + 1) // 2
            if Hp <= 0 or Wp <= 0:
                continue
            cfg_p = _bf16_tile(cin) if mb else _f32_tile(N * Hp * Wp, cin)
            packed, g = pack_weight_dev(w, [_ceil4(cin)], [cin], cfg_p, 2, 2, pad, (a, b), wscale, sigma, bf16=mb, batch=batch)
            _run_engine([(dy, 0, Cout)], packed, cin, cfg_p, N, Ho, Wo, Hp, Wp, g[0], g[1], 1, g[2], g[3], out,
                        residual=act_mask, res_mode=res_mode, slope=slope, free_extent=1, out_step=2, out_off=(a, b),
                        out_hw=(H, W), name=f"{name}[phase {a}{b}]", flops=fl / 4, mma_bf16=mb)
    return out


def conv_wgrad(dy: Act, x: Act, x_up: int, ci_base: int, cin_tot: int, KH: int, KW: int, stride: int, pad: int,
               dw: torch.Tensor, accumulate: bool = False, name: str = "wgrad", dbias: Optional[torch.Tensor] = None,
               dbias_accumulate: bool = False):
    """dW[:, ci_base:ci_base+x.C] (+)= wgrad(dY, x) -- hrv_conv2d_wgrad_nhwc_f32.  ``dw``: OIHW fp32 device.
    ``dbias`` ([Cout], optional): the bias gradient, fused as a ones-column of the same reduction."""
    lib = _lib.load()
    N, Ho, Wo, Cout = dy.N, dy.H, dy.W, dy.C
    H, W = (x.H << x_up, x.W << x_up) if x_up >= 0 else (x.H >> -x_up, x.W >> -x_up)
    assert dw.is_contiguous() and tuple(dw.shape) == (Cout, cin_tot, KH, KW), (dw.shape, Cout, cin_tot, KH, KW)
    if (Cout == 1 and x_up == 0 and ci_base == 0 and cin_tot == x.C and not dy.bf16 and
            (Ho, Wo) == (H + 2 * pad - KH + 1, W + 2 * pad - KW + 1) and _cout1_ok(dw, x, stride, pad, "wgrad")):
        S = lib.hrv_conv_cout1_wgrad_slabs(N, Ho, Wo)
        ws = _workspace(dy.t.device, 4 * S * (x.C * KH * KW + 1))
        d = _cout1_desc(dw, x, pad, 1.0, None, dy)
        d.workspace = ws.data_ptr()
        with _Timed("wgrad", name, 2.0 * N * Ho * Wo * x.C * KH * KW, ops.act_bytes(x) + 4.0 * N * Ho * Wo, "cout1_kernel"):
            _lib.check(lib.hrv_conv_cout1_wgrad_f32(C.byref(d), dw.data_ptr(), 1 if accumulate else 0,
                                                    None if dbias is None else dbias.data_ptr(), 1 if dbias_accumulate else 0,
                                                    _stream()), f"hrv_conv_cout1_wgrad_f32[{name}]")
        return
    wo_real = Wo
    tr2 = (stride == 1 and (Ho, Wo) == (H, W) and KH == KW == 2 and pad == 1 and x_up == 0 and Cout % 64 == 0 and 32 < x.Cp <= 64 and
           N * Ho * Wo >= 8192 and os.environ.get("HRV_WGRAD_TR", "1") != "0")          # (wgrad_tr.hip's 2x2 class takes any width)
    if (not tr2 and MMA_BF16[0] and dy.bf16 and x.bf16 and x_up == 0 and (KH, KW, stride, pad) == (4, 4, 2, 2) and
            (Ho, Wo) == (H // 2 + 1, W // 2 + 1) and
            lib.hrv_conv2d_wgrad_s2_supported(Cout, x.Cp, x.cstride, x.coff, dy.cstride, dy.coff, N, H, W)):
        tr2 = True                      # (wgrad_s2.hip: any width as well)
    if MMA_BF16[0] and Wo % 4 != 0 and dy.bf16 and x.bf16 and dy.coff == 0 and dy.cstride == dy.C and dy.C % 8 == 0 and not tr2:
        dy = pad_width_bf16(dy)          # (the same zero columns for a bf16-stored dY: the PatchGAN with bf16 feature maps)
        Wo = dy.W
    if (MMA_BF16[0] and Wo % 4 != 0 and Wo >= 32 and not (x.bf16 or dy.bf16) and dy.coff == 0 and dy.cstride == dy.Cp):
        # odd-sized maps (the PatchGAN's 513 / 257 / 129 columns): the bf16 matrix-core kernel stages quads of 4 pixels
        # of one image row, so dY gets zero columns up to the next multiple of 4 -- they add nothing to dW or the bias
        # gradient (the X taps they would pair with are never weighted) -- instead of taking the fp32 kernel (60-100 TFLOP/s)
        dy = Act(torch.nn.functional.pad(dy.t, (0, 0, 0, (-Wo) % 4)), dy.C)
        Wo = dy.W
    need = lib.hrv_conv2d_wgrad_workspace_bytes(Cout, cin_tot, KH, KW, N * Ho * Wo)
    ws = _workspace(dy.t.device, need)
    fl = 2.0 * N * Ho * wo_real * Cout * x.C * KH * KW
    # mixed precision: bf16 matrix cores (needs Wo % 4 == 0: narrow odd-sized maps keep the fp32 kernel)
    fn = lib.hrv_conv2d_wgrad_bf16mma_nhwc_f32 if (MMA_BF16[0] and Wo % 4 == 0) else lib.hrv_conv2d_wgrad_nhwc_f32
    args = (dy.t.data_ptr(), dy.cstride, dy.coff, Cout, x.t.data_ptr(), x.Cp, x.cstride, x.coff, x_up, x.C, ci_base,
            cin_tot, N, H, W, Ho, Wo, KH, KW, stride, pad, ws.data_ptr(), ws.numel() * 4, dw.data_ptr(),
            1 if accumulate else 0, None if dbias is None else dbias.data_ptr(), 1 if dbias_accumulate else 0)
    nbytes = ops.act_bytes(dy) + ops.act_bytes(x) + 4.0 * Cout * x.C * KH * KW
    with _Timed("wgrad", name, fl, nbytes, "conv_wgrad_tr_kernel" if (x.bf16 or dy.bf16) else "conv_wgrad_kernel"):
        if x.bf16 or dy.bf16:
            assert MMA_BF16[0] and (Wo % 4 == 0 or tr2), f"{name}: bf16-stored operands need the bf16 matrix-core weight gradient"
            assert x.bf16, f"{name}: bf16 dY with an fp32 X is not built"
            _lib.check(lib.hrv_conv2d_wgrad_bf16mma_st_nhwc_f32(*args, (1 if dy.bf16 else 0) | 2, _stream()),
                       "hrv_conv2d_wgrad_bf16mma_st_nhwc_f32")
        else:
            _lib.check(fn(*args, _stream()), "hrv_conv2d_wgrad_nhwc_f32")


def colsum(a: Act, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[c] (+)= sum over pixels of a[..., c]  (bias gradient) -- hrv_colsum_nhwc_f32."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(a.C, dtype=torch.float32, device=a.t.device)
        accumulate = False
    P = a.N * a.H * a.W
    ws = _workspace(a.t.device, 256 * a.Cp * 4)
    _lib.check(lib.hrv_colsum_nhwc_f32(a.t.data_ptr(), P, a.C, a.cstride, a.coff, ws.data_ptr(), ws.numel() * 4,
                                       out.data_ptr(), 1 if accumulate else 0, _stream()), "hrv_colsum_nhwc_f32")
    return out


# ---------------------------------------------------------------------------------------------
# HBM-bound training kernels (train.hip)
# ---------------------------------------------------------------------------------------------
def _norm_bwd_desc(x, mean, rstd, dout, act, slope, out, g1p, z, noise_scale, want_dgb, dx, dx_accumulate, dnoise_scale, dns_accumulate,
                   dgb_bf16, dx_bf16, dgb=None):
    """-> (descriptor, dx Act, dgb Act or None, keep-alive tensors)"""
    lib = _lib.load()
    N, H, W, Cp = x.N, x.H, x.W, x.Cp
    dev = x.t.device
    if dx is None:
        # ``dx_bf16``: the gradient of a convolution OUTPUT that only that convolution's backward reads (matrix cores)
        dx = ops.alloc(N, H, W, x.C, dev, bf16=dx_bf16 and x.C % 8 == 0)
        dx_accumulate = False
    # stage 1 -> stage 2 intermediate: bf16 in mixed precision (written once, read once: 4 of the ~34 bytes per element)
    dnh = torch.empty((N, H, W, Cp), dtype=torch.bfloat16 if MMA_BF16[0] else torch.float32, device=dev)
    if dgb is None:      # (a caller-provided dgb: its dbeta half already holds dout -- hrviton_hip.h, "dbeta in place")
        dgb = Act(torch.empty((N, H, W, 2 * Cp), dtype=torch.bfloat16 if dgb_bf16 else torch.float32, device=dev),
                  2 * Cp) if want_dgb else None
    ws = torch.empty(lib.hrv_norm_bwd_workspace_elems(N, H, W, Cp), dtype=torch.float32, device=dev)
    d = _lib.hrv_norm_bwd_t()
    d.N, d.H, d.W, d.C = N, H, W, Cp
    if isinstance(x, ops.ActUp):
        d.x, d.x_cstride, d.x_coff = x.lo.t.data_ptr(), x.lo.cstride, x.lo.coff
        d.x_up_channels, d.x2, d.x2_cstride, d.x2_coff = x.lo.C, x.hi.t.data_ptr(), x.hi.cstride, x.hi.coff
    else:
        d.x, d.x_cstride, d.x_coff = x.t.data_ptr(), x.cstride, x.coff
    if z is not None:
        d.noise_z, d.noise_scale = z.data_ptr(), noise_scale.data_ptr()
    d.mean, d.rstd = mean.data_ptr(), rstd.data_ptr()
    if out is not None:
        d.out, d.out_cstride, d.out_coff = out.t.data_ptr(), out.cstride, out.coff
        d.out_bf16 = 1 if out.bf16 else 0
    if g1p is not None:
        d.g1p, d.g1p_cstride, d.g1p_coff = g1p.t.data_ptr(), g1p.cstride, g1p.coff
        d.g1p_bf16 = 1 if g1p.bf16 else 0
    d.dout, d.dout_cstride, d.dout_coff = dout.t.data_ptr(), dout.cstride, dout.coff
    d.dout_bf16 = 1 if dout.bf16 else 0
    d.dnh, d.dnh_cstride, d.dnh_coff = dnh.data_ptr(), Cp, 0
    d.dnh_bf16 = 1 if dnh.dtype == torch.bfloat16 else 0
    if dgb is not None:
        d.dgb, d.dgb_cstride, d.dgb_coff = dgb.t.data_ptr(), dgb.cstride, dgb.coff
        d.dgb_bf16 = 1 if dgb.bf16 else 0
    d.dx, d.dx_cstride, d.dx_coff = dx.t.data_ptr(), dx.cstride, dx.coff
    d.dx_accumulate = 1 if dx_accumulate else 0
    d.dx_bf16 = 1 if dx.bf16 else 0
    assert not (dx.bf16 and dx_accumulate)
    d.act, d.act_slope = act, slope
    d.dns_accumulate = 1 if dns_accumulate else 0
    d.dnoise_scale = None if dnoise_scale is None else dnoise_scale.data_ptr()
    d.workspace = ws.data_ptr()
    return d, dx, dgb, (dnh, ws)


def norm_bwd(x: Act, mean: torch.Tensor, rstd: torch.Tensor, dout: Act, act: int = ACT_NONE, slope: float = 0.2,
             out: Optional[Act] = None, g1p: Optional[Act] = None, z: Optional[torch.Tensor] = None,
             noise_scale: Optional[torch.Tensor] = None, want_dgb: bool = False, dx: Optional[Act] = None,
             dx_accumulate: bool = False, dnoise_scale: Optional[torch.Tensor] = None, dns_accumulate: bool = False,
             dgb_bf16: bool = False, dx_bf16: bool = False, dgb: Optional[Act] = None):
    """hrv_spade_norm_bwd_nhwc_f32.  Returns (dx Act, dgb Act [.., 2C] or None).  ``dgb_bf16``: store
    [dgamma | dbeta] in bf16 (mixed precision: only the gamma|beta conv's matrix-core backward reads it).  ``dgb``: the
    [dgamma | dbeta] tensor allocated by the caller whose dbeta half ``dout`` is (norm_bwd_dgb / the header's "dbeta in place")."""
    lib = _lib.load()
    d, dx, dgb, _keep = _norm_bwd_desc(x, mean, rstd, dout, act, slope, out, g1p, z, noise_scale, want_dgb, dx, dx_accumulate, dnoise_scale,
                                       dns_accumulate, dgb_bf16, dx_bf16, dgb)
    with _Timed("norm_bwd", "spade_norm_bwd", 0.0, 4.0 * x.N * x.H * x.W * x.Cp * (7 + (2 if want_dgb else 0))):
        _lib.check(lib.hrv_spade_norm_bwd_nhwc_f32(C.byref(d), _stream()), "hrv_spade_norm_bwd_nhwc_f32")
    return dx, dgb


def norm_bwd_dgb(x: Act, bf16: bool) -> Tuple[Act, Act]:
    """The [dgamma | dbeta] tensor of a SPADE norm over ``x`` and its dbeta half: the data gradient that produces the norm's ``dout``
    writes it straight into that half (activation derivative applied in its epilogue), norm_bwd(dgb=...) then neither reads the
    activation output nor stores dbeta a second time."""
    dgb = Act(torch.empty((x.N, x.H, x.W, 2 * x.Cp), dtype=torch.bfloat16 if bf16 else torch.float32, device=x.t.device), 2 * x.Cp)
    return dgb, dgb.slice(x.Cp, x.C)


def norm_bwd2(x: Act, a: dict, b: dict):
    """Two normalisations of the same x in one pass per stage (hrv_spade_norm_bwd2_nhwc_f32): ``a`` / ``b`` hold the keyword
    arguments of norm_bwd except x / dx / dx_accumulate.  Returns (dx = dx_a + dx_b (fp32), dgb_a, dgb_b)."""
    lib = _lib.load()
    da, dx, dgb_a, _ka = _norm_bwd_desc(x, a["mean"], a["rstd"], a["dout"], a.get("act", ACT_NONE), a.get("slope", 0.2), a.get("out"), a.get("g1p"),
                                        a.get("z"), a.get("noise_scale"), a.get("want_dgb", False), None, False, a.get("dnoise_scale"),
                                        a.get("dns_accumulate", False), a.get("dgb_bf16", False), False)
    db, _, dgb_b, _kb = _norm_bwd_desc(x, b["mean"], b["rstd"], b["dout"], b.get("act", ACT_NONE), b.get("slope", 0.2), b.get("out"), b.get("g1p"),
                                       b.get("z"), b.get("noise_scale"), b.get("want_dgb", False), dx, False, b.get("dnoise_scale"),
                                       b.get("dns_accumulate", False), b.get("dgb_bf16", False), False)
    with _Timed("norm_bwd", "spade_norm_bwd x2", 0.0, 4.0 * x.N * x.H * x.W * x.Cp * (10 + 2 * (a.get("want_dgb", False) + b.get("want_dgb", False)))):
        _lib.check(lib.hrv_spade_norm_bwd2_nhwc_f32(C.byref(da), C.byref(db), _stream()), "hrv_spade_norm_bwd2_nhwc_f32")
    return dx, dgb_a, dgb_b


LOSS_L1, LOSS_HINGE_D_FAKE, LOSS_HINGE_D_REAL, LOSS_NEG_MEAN, LOSS_MSE = 0, 1, 2, 3, 4
LOSS_RELU_MASK = 16      # or-ed into the mode: a = ReLU(pre), the returned gradient is w.r.t. pre


def loss(a: torch.Tensor, b: Optional[torch.Tensor], mode: int, lscale: float, gscale: float, loss_out: torch.Tensor,
         accumulate: bool = True, want_grad: bool = True, grad_bf16: bool = False) -> Optional[torch.Tensor]:
    """hrv_loss_f32 over flat contiguous tensors: loss_out[0] (+)= lscale*sum(l); returns grad (same shape as a).
    ``grad_bf16`` (bf16-stored operands only): the gradient is stored in bf16 too (mixed-precision VGG backward)."""
    lib = _lib.load()
    cl = torch.channels_last
    bf = a.dtype == torch.bfloat16          # bf16-stored operands (mixed-precision VGG taps): loss and gradient stay fp32
    assert b is None or b.dtype == a.dtype, "loss: both operands share one storage type"
    ok = a.is_contiguous() and (b is None or (b.shape == a.shape and b.is_contiguous()))
    ok = ok or (a.dim() == 4 and a.is_contiguous(memory_format=cl) and
                (b is None or (b.shape == a.shape and b.is_contiguous(memory_format=cl))))
    assert ok, "loss: operands must share one dense layout (strides of size-1 dims do not matter)"
    grad_bf16 = bool(grad_bf16 and bf and a.numel() % 4 == 0)
    grad = torch.empty_like(a, dtype=torch.bfloat16 if grad_bf16 else torch.float32) if want_grad else None
    if grad_bf16:
        mode |= 32
    ws = _workspace(a.device, 4096)
    fn = lib.hrv_loss_bf16in_f32 if bf else lib.hrv_loss_f32
    with _Timed("loss", "loss_f32", 0.0, (2.0 if bf else 4.0) * a.numel() * (1 + (b is not None)) + 4.0 * a.numel() * (grad is not None)):
        _lib.check(fn(a.data_ptr(), None if b is None else b.data_ptr(), a.numel(), mode, lscale, gscale,
                      None if grad is None else grad.data_ptr(), ws.data_ptr(), loss_out.data_ptr(),
                      1 if accumulate else 0, _stream()), "hrv_loss_f32")
    return grad


def downsum2x2(dhi: Act, dlo: Optional[Act] = None, accumulate: bool = False, out_bf16: bool = False) -> Act:
    lib = _lib.load()
    Hl, Wl = dhi.H // 2, dhi.W // 2
    if dlo is None:
        dlo = ops.alloc(dhi.N, Hl, Wl, dhi.C, dhi.t.device, bf16=out_bf16 and dhi.C % 8 == 0)
        accumulate = False
    assert not (dlo.bf16 and accumulate)
    with _Timed("ew", "downsum2x2", 0.0, ops.act_bytes(dhi) * (1.25 + (0.25 if accumulate else 0))):
        _lib.check(lib.hrv_downsum2x2_nhwc_f32(dhi.t.data_ptr(), dhi.N, Hl, Wl, dhi.Cp, dhi.cstride, dhi.coff,
                                               dlo.t.data_ptr(), dlo.cstride, dlo.coff,
                                               2 if dlo.bf16 else (1 if accumulate else 0), _stream()),
                   "hrv_downsum2x2_nhwc_f32")
    return dlo


def avgpool3x3s2_bwd(dy: Act, H: int, W: int, dx: Optional[Act] = None, accumulate: bool = False) -> Act:
    lib = _lib.load()
    if dx is None:
        dx = ops.alloc(dy.N, H, W, dy.C, dy.t.device)
        accumulate = False
    with _Timed("pool", "avgpool3x3s2_bwd", 0.0, ops.act_bytes(dy) + ops.act_bytes(dx) * (2 if accumulate else 1)):
        _lib.check(lib.hrv_avgpool3x3s2_bwd_nhwc_f32(dy.t.data_ptr(), dy.N, H, W, dy.Cp, dy.cstride, dy.coff,
                                                     dx.t.data_ptr(), dx.cstride, dx.coff, 1 if accumulate else 0,
                                                     _stream()), "hrv_avgpool3x3s2_bwd_nhwc_f32")
    return dx


def maxpool2x2(x: Act) -> Act:
    """2x2 / 2 max pool.  A bf16-stored x gives a bf16-stored y, exact; that variant is for ReLU outputs (x >= +0, as in
    VGG19): it takes the integer max of the stored 16-bit patterns, whose order is the order of non-negative values."""
    lib = _lib.load()
    assert x.coff == 0 and x.cstride == x.Cp
    y = ops.alloc(x.N, x.H // 2, x.W // 2, x.C, x.t.device, bf16=x.bf16)
    fn = lib.hrv_maxpool2x2_nhwc_bf16 if x.bf16 else lib.hrv_maxpool2x2_nhwc_f32
    with _Timed("pool", "maxpool2x2", 0.0, ops.act_bytes(x) * 1.25):
        _lib.check(fn(x.t.data_ptr(), x.N, x.H, x.W, x.Cp, y.t.data_ptr(), _stream()), "hrv_maxpool2x2_nhwc")
    return y


def maxpool2x2_bwd(x: Act, dy: Act, relu: bool = False) -> Act:
    """``relu``: x = ReLU(pre) -- the result is the gradient w.r.t. pre (the ReLU derivative rides along)."""
    lib = _lib.load()
    dx = ops.alloc(x.N, x.H, x.W, x.C, x.t.device, bf16=dy.bf16)
    assert (relu or not x.bf16) and (not dy.bf16 or (x.bf16 and relu)), \
        "maxpool2x2_bwd: a bf16-stored x needs relu=True; bf16 gradients need a bf16 x"
    if dy.bf16:
        fn = lib.hrv_maxpool2x2_bwd_relu_nhwc_bf16
    else:
        fn = (lib.hrv_maxpool2x2_bwd_relu_nhwc_xbf16 if x.bf16 else lib.hrv_maxpool2x2_bwd_relu_nhwc_f32) if relu \
            else lib.hrv_maxpool2x2_bwd_nhwc_f32
    with _Timed("pool", "maxpool2x2_bwd", 0.0, ops.act_bytes(x) + 5.0 * x.N * x.H * x.W * x.C):
        _lib.check(fn(x.t.data_ptr(), dy.t.data_ptr(), x.N, x.H, x.W, x.Cp, dx.t.data_ptr(), _stream()),
                   "hrv_maxpool2x2_bwd_nhwc_f32")
    return dx


def adam_step(w: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float,
              eps: float, weight_decay: float, step: int, grad_scale: float = 1.0):
    lib = _lib.load()
    with _Timed("adam", "adam_f32", 0.0, 28.0 * w.numel()):     # read w, g, m, v; write w, m, v
        _lib.check(lib.hrv_adam_f32(w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), lr, beta1, beta2, eps,
                                    weight_decay, step, grad_scale, _stream()), "hrv_adam_f32")


def adam_step_dev(w: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step_dev: torch.Tensor, lr_dev: torch.Tensor,
                  hyper: torch.Tensor, beta1: float, beta2: float, eps: float, weight_decay: float, grad_scale: float = 1.0):
    """adam_step with the step count (int32 [1]) and the learning rate (float [1]) on the device: capturable in a hipGraph."""
    lib = _lib.load()
    _lib.check(lib.hrv_adam_hyper_f32(step_dev.data_ptr(), lr_dev.data_ptr(), beta1, beta2, hyper.data_ptr(), _stream()),
               "hrv_adam_hyper_f32")
    with _Timed("adam", "adam_f32", 0.0, 28.0 * w.numel()):
        _lib.check(lib.hrv_adam_dev_f32(w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), hyper.data_ptr(), beta1,
                                        beta2, eps, weight_decay, grad_scale, _stream()), "hrv_adam_dev_f32")


def spectral_sigma(w_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor, power_iterations: int,
                   eps: float = 1e-12) -> torch.Tensor:
    """In-place power iteration on (u, v) + sigma (device scalar tensor [1])."""
    lib = _lib.load()
    R = w_orig.shape[0]
    K = w_orig.numel() // R
    scratch = torch.empty(R, dtype=torch.float32, device=w_orig.device)
    sigma = torch.empty(1, dtype=torch.float32, device=w_orig.device)
    _lib.check(lib.hrv_spectral_norm_f32(w_orig.data_ptr(), R, K, u.data_ptr(), v.data_ptr(), power_iterations, eps,
                                         scratch.data_ptr(), sigma.data_ptr(), _stream()), "hrv_spectral_norm_f32")
    return sigma


class SpectralBatch:
    """Power iteration + sigma of a LIST of spectral-normalised convolutions in four launches
    (hrv_spectral_norm_batched_f32).  ``items``: (weight_orig, u, v) per layer.  ``run`` returns per-layer views
    (sigma [1], u_keep, v_keep) of three flat buffers allocated for THIS call -- the (u, v) that produced sigma, which
    the backward of this forward treats as constants (a later forward advances the module's u, v and gets its own)."""

    def __init__(self, items):
        self.items = list(items)
        self.Rs = [w.shape[0] for w, _, _ in self.items]
        self.Ks = [w.numel() // w.shape[0] for w, _, _ in self.items]
        self.jobs = (_lib.hrv_sn_job_t * len(self.items))()
        for j, (R, K) in enumerate(zip(self.Rs, self.Ks)):
            self.jobs[j].R, self.jobs[j].K = R, K

    def run(self, power_iterations: int, eps: float = 1e-12):
        lib = _lib.load()
        dev = self.items[0][0].device
        n, sR, sK = len(self.items), sum(self.Rs), sum(self.Ks)
        # sigma lives in ONE buffer per batch for the plan's lifetime: the batched weight packs read it by address
        sig = getattr(self, "_sig", None)
        if sig is None or sig.device != dev or sig.numel() != n:
            sig = self._sig = torch.empty(n, dtype=torch.float32, device=dev)
        ub = torch.empty(sR, dtype=torch.float32, device=dev)
        vb = torch.empty(sK, dtype=torch.float32, device=dev)
        scratch = _workspace(dev, 4 * sR)               # bytes: W v of every job
        sp, up, vp = sig.data_ptr(), ub.data_ptr(), vb.data_ptr()
        ro = ko = 0
        for j, (w, u, v) in enumerate(self.items):
            J = self.jobs[j]
            J.w, J.u, J.v = w.data_ptr(), u.data_ptr(), v.data_ptr()
            J.sigma, J.u_keep, J.v_keep = sp + 4 * j, up + 4 * ro, vp + 4 * ko
            ro, ko = ro + self.Rs[j], ko + self.Ks[j]
        _lib.check(lib.hrv_spectral_norm_batched_f32(self.jobs, n, power_iterations, eps, scratch.data_ptr(), _stream()),
                   "hrv_spectral_norm_batched_f32")
        return sig.split(1), ub.split(self.Rs), vb.split(self.Ks)


def prepare_convs(owner, convs, power_iteration: bool, backward: bool = True, extra_weights: Sequence[torch.Tensor] = ()):
    """TConv.prepare for every spectral-normalised convolution of ``convs`` at once (``owner`` caches the batch), then
    the plan's recorded weight packs in one launch (PackBatch, handed to every conv as ``pack_batch``; ``backward`` False:
    a no_grad forward, the data-gradient packs are left alone)."""
    _prepare_sigmas(owner, convs, power_iteration)
    if not convs:
        return
    dev = convs[0].wparam.data.device
    pb = getattr(owner, "_pack_batch", None)
    if pb is None or pb.device != dev:
        pb = owner._pack_batch = PackBatch(dev)
    for c in convs:
        c.pack_batch = pb
    # the records hold raw addresses of the weights: when any weight of the plan lives somewhere else than at the last
    # forward (the fused optimizer moves every parameter into its flat buffer at its FIRST step; load_state_dict on a
    # re-created module; .to(device)), the old records would keep packing from freed storages -- garbage into buffers nobody
    # reads in eager mode, an illegal access once the allocator has returned those storages to the driver (found by replaying
    # a captured iteration: torch.cuda.graph empties the cache before it captures)
    where = tuple(c.wparam.data.data_ptr() for c in convs) + tuple(w.data_ptr() for w in extra_weights)
    if getattr(pb, "where", None) != where:
        if pb.bufs:
            pb.reset()
        pb.where = where
    # (while a hipGraph is being captured the record table cannot be re-uploaded: a plan that still has unrecorded packs
    #  then packs one by one, which captures fine)
    if PACK_BATCHING[0] and not (pb.dirty and torch.cuda.is_current_stream_capturing()):
        pb.run(backward)


def _prepare_sigmas(owner, convs, power_iteration: bool):
    sn = [c for c in convs if c.spectral]
    if not sn:
        return
    batch = getattr(owner, "_sn_batch", None)
    key = tuple(id(c) for c in sn)
    if batch is None or batch[0] != key:
        for c in sn:
            assert c.wparam.data.is_contiguous() and c.conv.weight_u.is_contiguous() and c.conv.weight_v.is_contiguous()
        batch = (key, SpectralBatch([(c.wparam.data, c.conv.weight_u, c.conv.weight_v) for c in sn]))
        owner._sn_batch = batch
    b = batch[1]
    b.items = [(c.wparam.data, c.conv.weight_u, c.conv.weight_v) for c in sn]     # .data may be re-pointed between steps
    sig, us, vs = b.run(1 if power_iteration else 0)
    for c, s_, u_, v_ in zip(sn, sig, us, vs):
        c.sigma, c.u, c.v = s_, u_, v_


def spectral_grad(G: torch.Tensor, w_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor, sigma: torch.Tensor,
                  out: torch.Tensor, accumulate: bool = False):
    lib = _lib.load()
    R = w_orig.shape[0]
    K = w_orig.numel() // R
    ws = _workspace(G.device, 8192)
    _lib.check(lib.hrv_spectral_norm_bwd_f32(G.data_ptr(), w_orig.data_ptr(), u.data_ptr(), v.data_ptr(),
                                             sigma.data_ptr(), R, K, ws.data_ptr(), out.data_ptr(),
                                             1 if accumulate else 0, _stream()), "hrv_spectral_norm_bwd_f32")


def tanh_bwd(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty_like(dy)
    with _Timed("ew", "tanh_bwd", 0.0, 12.0 * dy.numel()):
        _lib.check(lib.hrv_tanh_bwd_f32(dy.data_ptr(), y.data_ptr(), dy.numel(), out.data_ptr(), _stream()), "hrv_tanh_bwd_f32")
    return out


def add_slice(a: Act, out: Act, accumulate: bool):
    """out (+)= a over a's channels (both views may be channel slices; both fp32 or both bf16-stored)."""
    lib = _lib.load()
    assert (a.N, a.H, a.W) == (out.N, out.H, out.W) and a.Cp <= out.cstride - out.coff
    assert a.bf16 == out.bf16, "add_slice: operands share one storage type"
    fn = lib.hrv_add_slice_nhwc_bf16 if a.bf16 else lib.hrv_add_slice_nhwc_f32
    with _Timed("ew", "add_slice", 0.0, ops.act_bytes(a) * (3 if accumulate else 2)):
        _lib.check(fn(a.t.data_ptr(), a.cstride, a.coff, out.t.data_ptr(), out.cstride, out.coff, a.Cp,
                      a.N * a.H * a.W, 1 if accumulate else 0, _stream()), "hrv_add_slice_nhwc")


def concat_nhwc_nchw(a: Act, b: torch.Tensor, out: Act):
    """out = cat((a, b), channels): ``a`` an fp32 NHWC activation, ``b`` fp32 NCHW, ``out`` a dense NHWC activation of
    a.C + b.shape[1] channels (pad channels zeroed) -- hrv_concat_nhwc_nchw_f32."""
    lib = _lib.load()
    N, Cb, H, W = b.shape
    assert (a.N, a.H, a.W) == (N, H, W) == (out.N, out.H, out.W) and not a.bf16 and not out.bf16 and out.coff == 0
    assert b.is_contiguous() and b.dtype == torch.float32 and out.C == a.C + Cb and out.t.is_contiguous()
    with _Timed("layout", "concat_nhwc_nchw", 0.0, 4.0 * N * H * W * (a.C + Cb + out.cstride)):
        _lib.check(lib.hrv_concat_nhwc_nchw_f32(a.t.data_ptr(), a.C, a.cstride, a.coff, b.data_ptr(), Cb, N, H, W, out.t.data_ptr(),
                                                out.cstride, _stream()), "hrv_concat_nhwc_nchw_f32")


def space_to_depth2(a: Act) -> Act:
    """[N,H,W,C] fp32 -> dense [N,H/2,W/2,4*Cp], channel ((y&1)*2 + (x&1))*Cp + c (hrv_space_to_depth2_nhwc_f32)."""
    lib = _lib.load()
    assert not a.bf16 and a.H % 2 == 0 and a.W % 2 == 0
    out = torch.empty((a.N, a.H // 2, a.W // 2, 4 * a.Cp), dtype=torch.float32, device=a.t.device)
    with _Timed("layout", "space_to_depth2", 0.0, 2.0 * ops.act_bytes(a)):
        _lib.check(lib.hrv_space_to_depth2_nhwc_f32(a.t.data_ptr(), a.N, a.H, a.W, a.Cp, a.cstride, a.coff, out.data_ptr(),
                                                    _stream()), "hrv_space_to_depth2_nhwc_f32")
    return Act(out, 4 * a.Cp)


def depth_to_space2(a2: Act, C_: int) -> Act:
    """The inverse: dense [N,Hc,Wc,4*Cp] -> [N,2*Hc,2*Wc,Cp] carrying ``C_`` real channels."""
    lib = _lib.load()
    Cp = a2.C // 4
    assert not a2.bf16 and a2.coff == 0 and a2.cstride == a2.C == 4 * Cp and Cp == _ceil4(C_)
    out = torch.empty((a2.N, 2 * a2.H, 2 * a2.W, Cp), dtype=torch.float32, device=a2.t.device)
    with _Timed("layout", "depth_to_space2", 0.0, 2.0 * ops.act_bytes(a2)):
        _lib.check(lib.hrv_depth_to_space2_nhwc_f32(a2.t.data_ptr(), a2.N, 2 * a2.H, 2 * a2.W, Cp, out.data_ptr(), _stream()),
                   "hrv_depth_to_space2_nhwc_f32")
    return Act(out, C_)


def act_bwd_(d: Act, y: Act, act: int, slope: float = 0.2):
    """d *= act'(y) in place."""
    lib = _lib.load()
    with _Timed("ew", "act_bwd", 0.0, ops.act_bytes(d) * 2 + ops.act_bytes(y, d.C)):
        _lib.check(lib.hrv_act_bwd_nhwc_f32(d.t.data_ptr(), d.cstride, d.coff, y.t.data_ptr(), y.cstride, y.coff, d.Cp,
                                            d.N * d.H * d.W, act, slope, _stream()), "hrv_act_bwd_nhwc_f32")


def scale_(x: torch.Tensor, s_host: float = 1.0, s_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    fn = lib.hrv_scale_bf16 if x.dtype == torch.bfloat16 else lib.hrv_scale_f32      # (a bf16 loss gradient: PatchGAN feature taps)
    _lib.check(fn(x.data_ptr(), x.numel(), s_host, None if s_dev is None else s_dev.data_ptr(), _stream()), "hrv_scale_f32")
    return x


# ---------------------------------------------------------------------------------------------
# condition-generator training kernels (cond_train.hip)
# ---------------------------------------------------------------------------------------------
class BNStats:
    """Batch statistics of one training-mode BatchNorm2d call: per-channel vectors of ceil4(C) floats."""
    __slots__ = ("mean", "rstd", "scale", "shift")

    def __init__(self, mean, rstd, scale, shift):
        self.mean, self.rstd, self.scale, self.shift = mean, rstd, scale, shift


def bn_train_stats(x: Act, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float, momentum: float,
                   running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor]) -> BNStats:
    """nn.BatchNorm2d in training mode over an NHWC Act: batch mean / biased var -> (scale, shift), and the
    in-place running-statistics update (hrv_instnorm_stats + hrv_bn_finalize)."""
    lib = _lib.load()
    mean_nc, rstd_nc = ops.instnorm_stats(x, eps=eps)
    dev, Cp = x.t.device, x.Cp
    vec = torch.zeros((4, Cp), dtype=torch.float32, device=dev)
    _lib.check(lib.hrv_bn_finalize_f32(mean_nc.data_ptr(), rstd_nc.data_ptr(), x.N, x.C, Cp, eps, x.H * x.W,
                                       None if weight is None else weight.data_ptr(),
                                       None if bias is None else bias.data_ptr(), eps, momentum,
                                       None if running_mean is None else running_mean.data_ptr(),
                                       None if running_var is None else running_var.data_ptr(),
                                       vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(),
                                       _stream()), "hrv_bn_finalize_f32")
    return BNStats(vec[0], vec[1], vec[2], vec[3])


def affine_act(x: Act, scale: torch.Tensor, shift: torch.Tensor, act: int = ACT_NONE, residual: Optional[Act] = None,
               slope: float = 0.2, out: Optional[Act] = None) -> Act:
    """out = act(x*scale[c] + shift[c] (+ residual)) -- hrv_affine_act_nhwc_f32."""
    lib = _lib.load()
    if out is None:
        out = ops.alloc(x.N, x.H, x.W, x.C, x.t.device)
    rp, rcs, rco = (None, 0, 0) if residual is None else (residual.t.data_ptr(), residual.cstride, residual.coff)
    npix = x.N * x.H * x.W
    with _Timed("apply", "bn_affine_act", 0.0, 4.0 * npix * x.Cp * (2 if residual is None else 3)):
        _lib.check(lib.hrv_affine_act_nhwc_f32(x.t.data_ptr(), x.cstride, x.coff, x.Cp, npix, scale.data_ptr(),
                                               shift.data_ptr(), rp, rcs, rco, act, slope, out.t.data_ptr(), out.cstride,
                                               out.coff, _stream()), "hrv_affine_act_nhwc_f32")
    return out


def bn_bwd(dy: Act, x: Act, st: BNStats, dgamma: Optional[torch.Tensor], dbeta: Optional[torch.Tensor],
           dx: Optional[Act] = None, dx_accumulate: bool = False) -> Act:
    """Training-mode BatchNorm backward (hrv_bn_bwd_nhwc_f32).  ``x`` = the normalised conv output."""
    lib = _lib.load()
    if dx is None:
        dx = ops.alloc(x.N, x.H, x.W, x.C, x.t.device)
        dx_accumulate = False
    ws = torch.empty(lib.hrv_bn_bwd_workspace_elems(x.C), dtype=torch.float32, device=x.t.device)
    npix = x.N * x.H * x.W
    with _Timed("norm_bwd", "bn_bwd", 0.0, 4.0 * npix * x.Cp * 5):
        _lib.check(lib.hrv_bn_bwd_nhwc_f32(dy.t.data_ptr(), dy.cstride, dy.coff, x.t.data_ptr(), x.cstride, x.coff, x.C,
                                           npix, st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(),
                                           ws.data_ptr(), dx.t.data_ptr(), dx.cstride, dx.coff,
                                           1 if dx_accumulate else 0,
                                           None if dgamma is None else dgamma.data_ptr(),
                                           None if dbeta is None else dbeta.data_ptr(), 0, _stream()),
                   "hrv_bn_bwd_nhwc_f32")
    return dx


def resize_bilinear_bwd(dy: Act, H: int, W: int, rh: float, rw: float, dx: Optional[Act] = None,
                        accumulate: bool = False) -> Act:
    """Adjoint of ops.resize_bilinear: dx[N,H,W,C] (+)= R^T dy."""
    lib = _lib.load()
    if dx is None:
        dx = ops.alloc(dy.N, H, W, dy.C, dy.t.device)
        accumulate = False
    with _Timed("resize", "bilinear_bwd", 0.0, 4.0 * dy.N * dy.H * dy.W * dy.Cp * 2):
        _lib.check(lib.hrv_resize_bilinear_bwd_nhwc_f32(dy.t.data_ptr(), dy.N, dy.H, dy.W, dy.Cp, dy.cstride, dy.coff,
                                                        rh, rw, dx.t.data_ptr(), H, W, dx.cstride, dx.coff,
                                                        1 if accumulate else 0, _stream()),
                   "hrv_resize_bilinear_bwd_nhwc_f32")
    return dx


def resize_bilinear_bwd_dense(dy: torch.Tensor, H: int, W: int, rh: float, rw: float) -> torch.Tensor:
    """Same adjoint on a dense [N,Ho,Wo,C] tensor of any C (the 2-channel flows)."""
    lib = _lib.load()
    N, Ho, Wo, Cc = dy.shape
    dx = torch.empty((N, H, W, Cc), dtype=torch.float32, device=dy.device)
    _lib.check(lib.hrv_resize_bilinear_bwd_nhwc_f32(dy.data_ptr(), N, Ho, Wo, Cc, Cc, 0, rh, rw, dx.data_ptr(), H, W, Cc,
                                                    0, 0, _stream()), "hrv_resize_bilinear_bwd_nhwc_f32")
    return dx


def resize_nearest_bwd(dy: Act, H: int, W: int, dx: Optional[Act] = None, accumulate: bool = False) -> Act:
    """Adjoint of ops.resize_nearest: dx[N,H,W,C] (+)= sum of the output pixels that selected each source pixel."""
    lib = _lib.load()
    if dx is None:
        dx = ops.alloc(dy.N, H, W, dy.C, dy.t.device)
        accumulate = False
    with _Timed("resize", "nearest_bwd", 0.0, 4.0 * dy.N * dy.H * dy.W * dy.Cp * 2):
        _lib.check(lib.hrv_resize_nearest_bwd_nhwc_f32(dy.t.data_ptr(), dy.N, dy.H, dy.W, dy.Cp, dy.cstride, dy.coff, dx.t.data_ptr(), H, W,
                                                       dx.cstride, dx.coff, 1 if accumulate else 0, _stream()),
                   "hrv_resize_nearest_bwd_nhwc_f32")
    return dx


def resize_nearest_bwd_dense(dy: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """Same adjoint on a dense [N,Ho,Wo,C] tensor of any C (the 2-channel flows)."""
    lib = _lib.load()
    N, Ho, Wo, Cc = dy.shape
    dx = torch.empty((N, H, W, Cc), dtype=torch.float32, device=dy.device)
    _lib.check(lib.hrv_resize_nearest_bwd_nhwc_f32(dy.data_ptr(), N, Ho, Wo, Cc, Cc, 0, dx.data_ptr(), H, W, Cc, 0, 0, _stream()),
               "hrv_resize_nearest_bwd_nhwc_f32")
    return dx


def flow_warp_bwd(src: Act, flow_up: torch.Tensor, norm_x: float, norm_y: float, dout: Act,
                  dsrc: Optional[Act], dflow: Optional[torch.Tensor], dflow_accumulate: bool = False):
    """Adjoint of ops.flow_warp.  ``dsrc`` (an accumulator Act, already initialised) receives the atomic
    scatter; ``dflow`` [N,Ho,Wo,2] the gradient w.r.t. the upsampled un-normalised flow."""
    lib = _lib.load()
    d = _lib.hrv_flow_warp_bwd_t()
    d.src, d.N, d.H, d.W, d.C = src.t.data_ptr(), src.N, src.H, src.W, src.Cp
    d.src_cstride, d.src_coff = src.cstride, src.coff
    d.flow_up, d.Ho, d.Wo = flow_up.data_ptr(), dout.H, dout.W
    d.norm_x, d.norm_y = norm_x, norm_y
    d.dout, d.dout_cstride, d.dout_coff = dout.t.data_ptr(), dout.cstride, dout.coff
    if dsrc is not None:
        d.dsrc, d.dsrc_cstride, d.dsrc_coff = dsrc.t.data_ptr(), dsrc.cstride, dsrc.coff
    if dflow is not None:
        d.dflow, d.dflow_accumulate = dflow.data_ptr(), 1 if dflow_accumulate else 0
    with _Timed("warp", "flow_warp_bwd", 0.0, 4.0 * dout.N * dout.H * dout.W * dout.Cp * 6):
        _lib.check(lib.hrv_flow_warp_bwd_nhwc_f32(C.byref(d), _stream()), "hrv_flow_warp_bwd_nhwc_f32")


def mul_(x: Act, m: torch.Tensor) -> Act:
    """x *= m in place over the whole (dense) buffer -- hrv_mul_f32 (dropout mask, forward and backward)."""
    lib = _lib.load()
    assert x.t.is_contiguous() and m.shape == x.t.shape and m.is_contiguous()
    _lib.check(lib.hrv_mul_f32(x.t.data_ptr(), m.data_ptr(), x.t.numel(), x.t.data_ptr(), _stream()), "hrv_mul_f32")
    return x


# ---------------------------------------------------------------------------------------------------------------
# SPADE gamma|beta convolution + modulate, and its data gradient, on the dedicated kernel (csrc/spade_gb.hip)
# ---------------------------------------------------------------------------------------------------------------
def spade_gb_ok(mode: int, C_: int, Cp: int, hid: int, N: int, H: int, W: int) -> bool:
    """The dedicated kernel serves this layer (hrv_spade_gb_supported; HRV_SPADE_GB=0 switches it off for A/B runs)."""
    if os.environ.get("HRV_SPADE_GB", "1") == "0":
        return False
    return bool(_lib.load().hrv_spade_gb_supported(mode, C_, Cp, hid, N, H, W))


def spade_gb_pack(mode: int, w_gamma: torch.Tensor, w_beta: torch.Tensor) -> torch.Tensor:
    """(conv_gamma.weight, conv_beta.weight) [C, hid, 3, 3] fp32 -> the bf16 fragment-order stream of ``mode``
    (0: forward, columns = (gamma | beta) pairs; 1: data gradient over [dgamma | dbeta])."""
    lib = _lib.load()
    ops.require_cuda(w_gamma, "spade_gb_pack")
    assert w_gamma.is_contiguous() and w_beta.is_contiguous() and w_gamma.shape == w_beta.shape
    C_, hid = w_gamma.shape[0], w_gamma.shape[1]
    nbytes = lib.hrv_spade_gb_packed_bytes(mode, C_, C_, hid)
    assert nbytes > 0, (mode, C_, hid)
    buf = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w_gamma.device)
    _lib.check(lib.hrv_spade_gb_pack_dev(mode, w_gamma.data_ptr(), w_beta.data_ptr(), C_, C_, hid, buf.data_ptr(), _stream()),
               "hrv_spade_gb_pack_dev")
    return buf


def spade_gb_forward(actv: Act, x: Act, mean: torch.Tensor, rstd: torch.Tensor, z: Optional[torch.Tensor],
                     noise_scale: Optional[torch.Tensor], packed: torch.Tensor, bias_gamma: torch.Tensor, bias_beta: torch.Tensor,
                     act: int, slope: float, out: Act, g1p: Optional[torch.Tensor], name: str, flops: float, nbytes: float):
    """out = act(IN(x + z*noise_scale) * (1 + conv_gamma(actv)) + conv_beta(actv)); g1p (optional) <- 1 + gamma."""
    lib = _lib.load()
    d = _lib.hrv_spade_gb_t()
    C_ = x.C
    d.mode, d.N, d.H, d.W = 0, x.N, x.H, x.W
    d.src, d.src_cstride, d.src_coff = actv.t.data_ptr(), actv.cstride, actv.coff
    d.w_packed = packed.data_ptr()
    d.C, d.Cp, d.hid = C_, C_, actv.C
    d.x, d.x_cstride, d.x_coff, d.x_f32 = x.t.data_ptr(), x.cstride, x.coff, 0 if x.bf16 else 1
    d.mean, d.rstd, d.stat_stride = mean.data_ptr(), rstd.data_ptr(), C_
    if z is not None:
        d.noise_z, d.noise_scale = z.data_ptr(), noise_scale.data_ptr()
    d.bias_gamma, d.bias_beta = bias_gamma.data_ptr(), bias_beta.data_ptr()
    if g1p is not None:
        d.g1p, d.g1p_bf16 = g1p.data_ptr(), 1 if g1p.dtype == torch.bfloat16 else 0
    d.act, d.act_slope = act, slope
    d.out, d.out_cstride, d.out_coff, d.out_f32 = out.t.data_ptr(), out.cstride, out.coff, 0 if out.bf16 else 1
    with ops._Timed("conv", name + " [spade_gb]", flops, nbytes, "spade_gb_kernel"):      # (the tag: bench.py prices this kernel's launches)
        _lib.check(lib.hrv_spade_gb_bf16(C.byref(d), _stream()), "hrv_spade_gb_bf16[forward]")


def spade_gb_dgrad(dgb: Act, packed: torch.Tensor, C_: int, mask: Optional[Act], slope: float, out: Act, name: str):
    """d(actv) [.., hid] = conv^T([dgamma | dbeta]) * (mask > 0 ? 1 : slope)."""
    lib = _lib.load()
    d = _lib.hrv_spade_gb_t()
    hid = out.C
    d.mode, d.N, d.H, d.W = 1, dgb.N, dgb.H, dgb.W
    d.src, d.src_cstride, d.src_coff = dgb.t.data_ptr(), dgb.cstride, dgb.coff
    d.w_packed = packed.data_ptr()
    d.C, d.Cp, d.hid = C_, C_, hid
    if mask is not None:
        assert mask.bf16
        d.mask, d.mask_cstride, d.mask_coff = mask.t.data_ptr(), mask.cstride, mask.coff
    d.act, d.act_slope = ACT_NONE, slope
    d.out, d.out_cstride, d.out_coff, d.out_f32 = out.t.data_ptr(), out.cstride, out.coff, 0 if out.bf16 else 1
    fl = 2.0 * dgb.N * dgb.H * dgb.W * 2 * C_ * hid * 9
    nb = ops.act_bytes(dgb) + ops.act_bytes(out) + (ops.act_bytes(mask) if mask is not None else 0.0)
    with ops._Timed("conv", name + " [spade_gb]", fl, nb, "spade_gb_kernel"):
        _lib.check(lib.hrv_spade_gb_bf16(C.byref(d), _stream()), "hrv_spade_gb_bf16[dgrad]")


# ---------------------------------------------------------------------------------------------------------------
# SPADENorm forward fused end to end: conv_shared + ReLU computed inside the gamma|beta kernel (csrc/spade_fused.hip)
# ---------------------------------------------------------------------------------------------------------------
def spade_fused_ok(C_: int, hid: int, label_nc: int, N: int, H: int, W: int) -> bool:
    """The fused kernel serves this norm (hrv_spade_fused_supported; HRV_SPADE_FUSED=0 switches it off for A/B runs)."""
    if os.environ.get("HRV_SPADE_FUSED", "1") == "0":
        return False
    return bool(_lib.load().hrv_spade_fused_supported(C_, hid, label_nc, N, H, W))


def spade_fused_pack(w_shared: torch.Tensor, b_shared: torch.Tensor, w_gamma: torch.Tensor, w_beta: torch.Tensor) -> torch.Tensor:
    """conv_shared.weight [128, label_nc, 3, 3] / .bias, conv_gamma.weight / conv_beta.weight [C, 128, 3, 3] (fp32, device) -> the
    bf16 fragment-order stream of hrv_spade_fused_bf16."""
    lib = _lib.load()
    ops.require_cuda(w_gamma, "spade_fused_pack")
    assert w_shared.is_contiguous() and b_shared.is_contiguous() and w_gamma.is_contiguous() and w_beta.is_contiguous()
    assert w_gamma.shape == w_beta.shape and w_shared.shape[0] == w_gamma.shape[1] == 128
    C_, label_nc = w_gamma.shape[0], w_shared.shape[1]
    nbytes = lib.hrv_spade_fused_packed_bytes(C_)
    assert nbytes > 0, C_
    buf = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w_gamma.device)
    _lib.check(lib.hrv_spade_fused_pack_dev(w_shared.data_ptr(), b_shared.data_ptr(), label_nc, w_gamma.data_ptr(), w_beta.data_ptr(),
                                            C_, buf.data_ptr(), _stream()), "hrv_spade_fused_pack_dev")
    return buf


def spade_fused_forward(seg: Act, seg_shift: int, x: Act, mean: torch.Tensor, rstd: torch.Tensor, z: Optional[torch.Tensor],
                        noise_scale: Optional[torch.Tensor], packed: torch.Tensor, bias_gamma: torch.Tensor, bias_beta: torch.Tensor,
                        act: int, slope: float, out: Act, g1p: Optional[torch.Tensor], actv: Optional[Act], name: str):
    """out = act(IN(x + z*noise_scale) * (1 + conv_gamma(a)) + conv_beta(a)), a = ReLU(conv_shared(nearest(seg))) computed in the
    kernel; g1p (optional) <- 1 + gamma; actv (optional, bf16 slice of 128 channels) <- a.  ``seg``: the bf16 label map
    [N, H << seg_shift, W << seg_shift, 8]."""
    lib = _lib.load()
    assert seg.bf16 and seg.cstride == 8 and seg.coff == 0 and out.bf16
    d = _lib.hrv_spade_fused_t()
    C_ = x.C
    d.N, d.H, d.W, d.C = x.N, x.H, x.W, C_
    d.seg, d.seg_H, d.seg_W, d.seg_shift = seg.t.data_ptr(), seg.H, seg.W, seg_shift
    d.w_packed = packed.data_ptr()
    if isinstance(x, ops.ActUp):
        d.x, d.x_cstride, d.x_coff, d.x_f32 = x.lo.t.data_ptr(), x.lo.cstride, x.lo.coff, 1
        d.x_up_channels, d.x2, d.x2_cstride, d.x2_coff = x.lo.C, x.hi.t.data_ptr(), x.hi.cstride, x.hi.coff
        xbytes = 4.0 * x.N * x.H * x.W * (x.lo.C / 4 + x.hi.C)
    else:
        d.x, d.x_cstride, d.x_coff, d.x_f32 = x.t.data_ptr(), x.cstride, x.coff, 0 if x.bf16 else 1
        xbytes = ops.act_bytes(x)
    d.mean, d.rstd = mean.data_ptr(), rstd.data_ptr()
    if z is not None:
        d.noise_z, d.noise_scale = z.data_ptr(), noise_scale.data_ptr()
    d.bias_gamma, d.bias_beta = bias_gamma.data_ptr(), bias_beta.data_ptr()
    if g1p is not None:
        assert g1p.dtype == torch.bfloat16
        d.g1p = g1p.data_ptr()
    d.act, d.act_slope = act, slope
    d.out, d.out_cstride, d.out_coff = out.t.data_ptr(), out.cstride, out.coff
    if actv is not None:
        assert actv.bf16 and actv.C == 128
        d.actv, d.actv_cstride, d.actv_coff = actv.t.data_ptr(), actv.cstride, actv.coff
    px = float(x.N * x.H * x.W)
    fl = 2.0 * px * 2 * C_ * 128 * 9          # the gamma|beta convolution (SURVEY 8d work; conv_shared's 2 * 72 * 128 per pixel rides along)
    nbytes = (px * 16 + xbytes + (px * 2.0 * C_ if g1p is not None else 0.0) + ops.act_bytes(out) + (px * 256 if actv is not None else 0.0) +
              2.0 * C_ * 128 * 9 * 2)
    with ops._Timed("conv", name + " [spade_gb]", fl, nbytes, "spade_fused_kernel"):      # (the tag: bench.py prices this kernel family's launches)
        _lib.check(lib.hrv_spade_fused_bf16(C.byref(d), _stream()), "hrv_spade_fused_bf16")


# ---------------------------------------------------------------------------------------------------------------
# 3x3 'same' convolution over one bf16 source, two blocks per CU (csrc/conv_p2.hip): VGG19 forward / data gradient, the
# data gradient of the SPADE (conv_gamma, conv_beta) pair
# ---------------------------------------------------------------------------------------------------------------
def conv_p2_ok(K: int, cols: int, N: int, H: int, W: int) -> bool:
    """hrv_conv_p2_supported (HRV_CONV_P2=0 switches the kernel off for A/B runs)."""
    if os.environ.get("HRV_CONV_P2", "1") == "0":
        return False
    return bool(_lib.load().hrv_conv_p2_supported(K, cols, N, H, W))


def conv_p2_pack(mode: int, w: torch.Tensor, w2: Optional[torch.Tensor], K: int, cols: int, sigma: Optional[torch.Tensor] = None,
                 wscale: float = 1.0, frozen=None) -> torch.Tensor:
    """The bf16 fragment-order weight stream of hrv_conv_p2_bf16 (mode 0 forward / 1 data gradient / 2 data gradient of a pair);
    ``frozen``: kept across calls like pack_weight_dev's."""
    lib = _lib.load()
    key = None
    if frozen is not None and sigma is None:
        key = ("p2", w.data_ptr(), frozen, ops.LOAD_EPOCH[0], tuple(w.shape), mode, K, cols, wscale)
        hit = _FROZEN_PACKS.get(key)
        if hit is not None:
            return hit[0]
    assert w.is_contiguous() and (w2 is None or w2.is_contiguous())
    nbytes = lib.hrv_conv_p2_packed_bytes(K, cols)
    assert nbytes > 0, (K, cols)
    buf = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.hrv_conv_p2_pack_dev(mode, w.data_ptr(), None if w2 is None else w2.data_ptr(), K, cols,
                                        None if sigma is None else sigma.data_ptr(), wscale, buf.data_ptr(), _stream()),
               "hrv_conv_p2_pack_dev")
    if key is not None:
        if len(_FROZEN_PACKS) > 256:
            _FROZEN_PACKS.clear()
        _FROZEN_PACKS[key] = (buf, ())
    return buf


def conv_p2(src: Act, packed: torch.Tensor, cols: int, out: Act, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: float = 0.2,
            mask: Optional[Act] = None, mask_slope: float = 0.0, name: str = "conv", flops: float = 0.0, tag: str = "",
            residual: Optional[Act] = None, res_after_mask: bool = False):
    """out = act(conv3x3(src) + bias [+ residual]) [* (mask > 0 ? 1 : mask_slope)] [+ residual if res_after_mask] on csrc/conv_p2.hip."""
    lib = _lib.load()
    assert src.bf16 and out.C == cols
    d = _lib.hrv_conv_p2_t()
    d.N, d.H, d.W, d.Cin = src.N, src.H, src.W, src.C
    d.src, d.src_cstride, d.src_coff, d.Cout = src.t.data_ptr(), src.cstride, src.coff, cols
    d.w_packed = packed.data_ptr()
    if bias is not None:
        d.bias = bias.data_ptr()
    d.act, d.act_slope = act, slope
    if mask is not None:
        assert mask.bf16 and mask.C == cols
        d.mask, d.mask_cstride, d.mask_coff, d.mask_slope = mask.t.data_ptr(), mask.cstride, mask.coff, mask_slope
    d.out, d.out_cstride, d.out_coff, d.out_f32 = out.t.data_ptr(), out.cstride, out.coff, 0 if out.bf16 else 1
    if residual is not None:
        assert residual.C == cols
        d.residual, d.res_cstride, d.res_coff, d.res_f32 = residual.t.data_ptr(), residual.cstride, residual.coff, 0 if residual.bf16 else 1
        d.res_after_mask = 1 if res_after_mask else 0
    nb = (ops.act_bytes(src) + ops.act_bytes(out) + (ops.act_bytes(mask) if mask is not None else 0.0) +
          (ops.act_bytes(residual) if residual is not None else 0.0) + 2.0 * src.C * cols * 9)
    with ops._Timed("conv", name + tag, flops, nb, "conv_p2_kernel"):
        _lib.check(lib.hrv_conv_p2_bf16(C.byref(d), _stream()), f"hrv_conv_p2_bf16[{name}]")
    return out


# ---------------------------------------------------------------------------------------------------------------
# csrc/conv_s2.hip -- PatchGAN's 4x4 stride-2 pad-2 convolution over a bf16-stored feature map, and its data gradient
# (NLayerDiscriminator, network_generator.py:263-272)
# ---------------------------------------------------------------------------------------------------------------
S2_FWD, S2_DGRAD, S2_CELLS = 0, 1, 2


def conv_s2_ok(mode: int, K: int, cols: int, Cph: int, N: int, Ho: int, Wo: int) -> bool:
    """hrv_conv_s2_supported (HRV_CONV_S2=0 switches the kernel off for A/B runs)."""
    if os.environ.get("HRV_CONV_S2", "1") == "0":
        return False
    return bool(_lib.load().hrv_conv_s2_supported(mode, K, cols, Cph, N, Ho, Wo))


def conv_s2_pack(mode: int, w: torch.Tensor, K: int, cols: int, Cph: int = 0, sigma: Optional[torch.Tensor] = None,
                 wscale: float = 1.0, split3: bool = False) -> torch.Tensor:
    """The bf16 fragment-order weight stream of hrv_conv_s2_bf16: mode S2_FWD (w = the layer's OIHW [cols][K][4][4]), S2_DGRAD (w = the
    forward OIHW [K][Cph][4][4], cols = 4 Cph) or S2_CELLS (w = [cols][K][2][2] over a space-to-depth source).  ``split3`` (S2_FWD /
    S2_CELLS): K = 3 x w's input channels, for a source from ``split3`` -- HRV_S2_SPLIT3 in the header."""
    lib = _lib.load()
    assert w.is_contiguous() and w.dtype == torch.float32
    nbytes = lib.hrv_conv_s2_packed_bytes(mode, K, cols)
    assert nbytes > 0, (mode, K, cols)
    buf = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=w.device)
    _lib.check(lib.hrv_conv_s2_pack_dev(mode | (4 if split3 else 0), w.data_ptr(), K, cols, Cph, None if sigma is None else sigma.data_ptr(),
                                        wscale, buf.data_ptr(), _stream()), "hrv_conv_s2_pack_dev")
    return buf


def conv_s2_pack_multi(jobs) -> list:
    """conv_s2_pack of several weights in ONE launch (hrv_conv_s2_pack_multi_dev, <= 8 per launch).  ``jobs``: (mode, w, K, cols, Cph,
    sigma, split3) tuples; returns the packed streams (views of one buffer) in order."""
    lib = _lib.load()
    if not jobs:
        return []
    sizes = []
    for mode, w, K, cols, Cph, sigma, sp in jobs:
        assert w.is_contiguous() and w.dtype == torch.float32
        nb = lib.hrv_conv_s2_packed_bytes(mode, K, cols)
        assert nb > 0 and nb % 16 == 0, (mode, K, cols)
        sizes.append(nb // 2)
    buf = torch.empty(sum(sizes), dtype=torch.bfloat16, device=jobs[0][1].device)
    outs, off = [], 0
    for n in sizes:
        outs.append(buf[off:off + n])
        off += n
    for i0 in range(0, len(jobs), 8):
        chunk = jobs[i0:i0 + 8]
        arr = (_lib.hrv_s2_pack_job_t * len(chunk))()
        for q, (mode, w, K, cols, Cph, sigma, sp) in enumerate(chunk):
            arr[q].mode_flags, arr[q].K, arr[q].cols, arr[q].Cph = mode | (4 if sp else 0), K, cols, Cph
            arr[q].w, arr[q].sigma = w.data_ptr(), (None if sigma is None else sigma.data_ptr())
            arr[q].wscale, arr[q].out = 1.0, outs[i0 + q].data_ptr()
        _lib.check(lib.hrv_conv_s2_pack_multi_dev(len(chunk), arr, _stream()), "hrv_conv_s2_pack_multi_dev")
    return outs


def split3(a: Act) -> Act:
    """fp32 activation -> dense bf16 [hi | lo | hi] (3 C channels, hrv_split3_nhwc_bf16): the source of a convolution packed with
    ``split3`` -- hi*hi + lo*hi + hi*lo on the bf16 matrix cores."""
    lib = _lib.load()
    assert not a.bf16 and a.C % 4 == 0
    out = torch.empty((a.N, a.H, a.W, 3 * a.C), dtype=torch.bfloat16, device=a.t.device)
    with _Timed("layout", "split3", 0.0, 2.5 * ops.act_bytes(a)):
        _lib.check(lib.hrv_split3_nhwc_bf16(a.t.data_ptr(), a.N * a.H * a.W, a.C, a.cstride, a.coff, out.data_ptr(), _stream()),
                   "hrv_split3_nhwc_bf16")
    return Act(out, 3 * a.C)


def conv_s2(mode: int, src: Act, packed: torch.Tensor, cols: int, out: Act, Cph: int = 0, bias: Optional[torch.Tensor] = None,
            act: int = ACT_NONE, slope: float = 0.2, residual: Optional[Act] = None, mask: Optional[Act] = None, mask_slope: float = 0.0,
            name: str = "conv", flops: float = 0.0):
    """out = act(conv(src) + bias [+ residual]) [* (mask > 0 ? 1 : mask_slope)] on csrc/conv_s2.hip (see hrv_conv_s2_t)."""
    lib = _lib.load()
    assert src.bf16 and out.C == (Cph if mode == S2_DGRAD else cols), (out.C, cols, Cph)
    d = _lib.hrv_conv_s2_t()
    d.mode, d.N, d.Hs, d.Ws, d.K = mode, src.N, src.H, src.W, src.C
    d.src, d.src_cstride, d.src_coff = src.t.data_ptr(), src.cstride, src.coff
    d.Ho, d.Wo, d.cols, d.Cph = out.H, out.W, cols, Cph
    d.w_packed = packed.data_ptr()
    if bias is not None:
        d.bias = bias.data_ptr()
    d.act, d.act_slope = act, slope
    d.out, d.out_cstride, d.out_coff, d.out_f32 = out.t.data_ptr(), out.cstride, out.coff, 0 if out.bf16 else 1
    if residual is not None:
        assert residual.C == out.C and (residual.N, residual.H, residual.W) == (out.N, out.H, out.W)
        d.residual, d.res_cstride, d.res_coff, d.res_f32 = residual.t.data_ptr(), residual.cstride, residual.coff, 0 if residual.bf16 else 1
    if mask is not None:
        assert mask.bf16 and mask.C == out.C and (mask.N, mask.H, mask.W) == (out.N, out.H, out.W)
        d.mask, d.mask_cstride, d.mask_coff, d.mask_slope = mask.t.data_ptr(), mask.cstride, mask.coff, mask_slope
    nb = (ops.act_bytes(src) + ops.act_bytes(out) + (ops.act_bytes(mask) if mask is not None else 0.0) +
          (ops.act_bytes(residual) if residual is not None else 0.0) + 2.0 * src.C * cols * (16 if mode == S2_FWD else 4))
    with ops._Timed("conv", name, flops, nb, "conv_s2_kernel"):
        _lib.check(lib.hrv_conv_s2_bf16(C.byref(d), _stream()), f"hrv_conv_s2_bf16[{name}]")
    return out


def space_to_depth2_cells(a: Act, Hp: int, Wp: int, split3: bool = False) -> Act:
    """[N,H,W,C] fp32 -> dense bf16 [N,Hp,Wp,4*Cp] (x3 as [hi | lo | hi] with ``split3``), cells / sub-pixels outside the image zero
    (hrv_space_to_depth2_cells_bf16)."""
    lib = _lib.load()
    assert not a.bf16
    K = 4 * a.Cp * (3 if split3 else 1)
    out = torch.empty((a.N, Hp, Wp, K), dtype=torch.bfloat16, device=a.t.device)
    with _Timed("layout", "space_to_depth2", 0.0, (1.5 + (1.0 if split3 else 0.0)) * ops.act_bytes(a)):
        _lib.check(lib.hrv_space_to_depth2_cells_bf16(a.t.data_ptr(), a.N, a.H, a.W, a.Cp, a.cstride, a.coff, Hp, Wp, 1 if split3 else 0,
                                                      out.data_ptr(), _stream()), "hrv_space_to_depth2_cells_bf16")
    return Act(out, K)


def instnorm_apply_bf16(a: Act, mean: torch.Tensor, rstd: torch.Tensor, act: int = ACT_NONE, slope: float = 0.2) -> Act:
    """lrelu((a - mean) * rstd) of an fp32 ``a`` stored in bf16 (a feature map that only matrix cores and the L1 tap read)."""
    lib = _lib.load()
    assert not a.bf16 and a.C % 8 == 0
    out = ops.alloc(a.N, a.H, a.W, a.C, a.t.device, bf16=True)
    with _Timed("apply", "instnorm_apply", 0.0, 6.0 * a.N * a.H * a.W * a.Cp):
        _lib.check(lib.hrv_instnorm_apply_nhwc_bf16out(a.t.data_ptr(), a.N, a.H, a.W, a.Cp, a.cstride, a.coff, mean.data_ptr(), rstd.data_ptr(),
                                                       act, slope, out.t.data_ptr(), out.cstride, out.coff, _stream()),
                   "hrv_instnorm_apply_nhwc_bf16out")
    return out


def pad_width_bf16(a: Act, mult: int = 4) -> Act:
    """A dense bf16 activation with zero columns appended up to a multiple of ``mult`` (conv_wgrad's quad staging of dY)."""
    lib = _lib.load()
    assert a.bf16 and a.coff == 0 and a.cstride == a.C and a.C % 8 == 0 and a.t.is_contiguous()
    Wp = (a.W + mult - 1) // mult * mult
    if Wp == a.W:
        return a
    out = torch.empty((a.N, a.H, Wp, a.C), dtype=torch.bfloat16, device=a.t.device)
    with _Timed("layout", "pad_width", 0.0, 2.0 * ops.act_bytes(a)):
        _lib.check(lib.hrv_pad_width_nhwc_bf16(a.t.data_ptr(), a.N * a.H, a.W, a.C, Wp, out.data_ptr(), _stream()), "hrv_pad_width_nhwc_bf16")
    return Act(out, a.C)
