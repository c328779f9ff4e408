"""GPU parity tests proper: every kernel is called through the C ABI
(hr_viton_amd.ops -> ctypes -> libhrviton_hip.so) and compared with the CPU
oracle on the same seeded inputs.  Tolerances are stated per test; fp32 paths
are held to <=1e-4 relative of the tensor's max (north_star allows 1e-3)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hrviton_oracle as O

pytestmark = pytest.mark.gpu

DIAG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _ops():
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops
    return ops


def _nhwc(ops, x):
    return ops.to_nhwc(x.cuda())


def _diag(name, got, want):
    """On mismatch, dump an error histogram keyed by (pixel%32, channel%32) -- the MFMA
    fragment coordinates -- so a single GPU run localises lane-mapping faults."""
    os.makedirs(DIAG_DIR, exist_ok=True)
    err = (got - want).abs()
    bad = err > 1e-3 * max(1.0, want.abs().max().item())
    lines = [f"{name}: shape {tuple(got.shape)} max_err {err.max().item():.4e} max_ref {want.abs().max().item():.4e} "
             f"bad_frac {bad.float().mean().item():.4f} got_nan {torch.isnan(got).sum().item()}"]
    if got.dim() == 4:  # NCHW
        N, C, H, W = got.shape
        pix = torch.arange(N * H * W).view(N, 1, H, W).expand(N, C, H, W)
        ch = torch.arange(C).view(1, C, 1, 1).expand(N, C, H, W)
        lines.append("bad by channel%32: " + str(torch.bincount((ch[bad] % 32), minlength=32).tolist()))
        lines.append("bad by pixel%32:   " + str(torch.bincount((pix[bad] % 32), minlength=32).tolist()))
        lines.append("bad by pixel//32%8:" + str(torch.bincount((pix[bad] // 32 % 8), minlength=8).tolist()))
        idx = bad.nonzero()[:8].tolist()
        for i in idx:
            lines.append(f"  at {i}: got {got[tuple(i)].item():.6f} want {want[tuple(i)].item():.6f}")
    with open(os.path.join(DIAG_DIR, "diag_" + name.replace("/", "_") + ".txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return "\n".join(lines)


def _assert_close(name, got, want, tol=1e-4):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(1.0, want.abs().max().item())
    if not (err <= tol * ref) or torch.isnan(got).any():
        pytest.fail(_diag(name, got, want))


def test_device_is_gfx950():
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import _lib
    lib = _lib.load()
    assert lib.hrv_device_check() == 0, lib.hrv_last_error()


def test_layout_roundtrip():
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    for C in (4, 13, 16, 3):
        x = torch.randn(2, C, 9, 7, generator=g)
        a = _nhwc(ops, x)
        assert a.t.shape == (2, 9, 7, (C + 3) // 4 * 4)
        assert torch.equal(a.t[..., :C].cpu(), x.permute(0, 2, 3, 1))
        assert (a.t[..., C:] == 0).all()
        assert torch.equal(ops.to_nchw(a).cpu(), x)
    # a slice of a wider concatenation buffer (also at channel offset 0): the neighbours' channels are left alone, fp32 and bf16
    for bf16 in (False, True):
        x = torch.randn(2, 13, 9, 7, generator=g)
        full = torch.full((2, 9, 7, 24), 7.0, device="cuda", dtype=torch.bfloat16 if bf16 else torch.float32)
        for off in (0, 8):
            full.fill_(7.0)
            ops.to_nhwc(x.cuda(), out=ops.Act(full, 13, off))
            want = x.permute(0, 2, 3, 1)
            want = want.to(torch.bfloat16).float() if bf16 else want
            assert torch.equal(full[..., off:off + 13].float().cpu(), want)
            rest = torch.cat([full[..., :off], full[..., off + 13:]], -1)
            assert (rest == 7.0).all()


@pytest.mark.parametrize("shape,out", [((2, 8, 5, 4), (10, 8)), ((1, 12, 6, 9), (12, 18)), ((2, 4, 7, 3), (19, 11))])
def test_resize_bilinear(shape, out):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g)
    N, C, H, W = shape
    Ho, Wo = out
    if (Ho, Wo) == (2 * H, 2 * W):
        want = O.resize_bilinear(x, scale_factor=2)
        rh = rw = 0.5
    else:
        want = O.resize_bilinear(x, size=out)
        rh, rw = H / Ho, W / Wo
    got = ops.to_nchw(ops.resize_bilinear(_nhwc(ops, x), Ho, Wo, rh, rw))
    _assert_close("resize_bilinear", got, want, 1e-6)
    add = torch.randn(N, C, Ho, Wo, generator=g)
    got = ops.to_nchw(ops.resize_bilinear(_nhwc(ops, x), Ho, Wo, rh, rw, addend=_nhwc(ops, add)))
    _assert_close("resize_bilinear_add", got, want + add, 1e-6)


@pytest.mark.parametrize("C,H,W,scale", [(4, 12, 8, 2), (24, 6, 10, 2), (8, 16, 12, 4)])
def test_flow_warp(C, H, W, scale):
    """Fused flow upsample + normalise + base grid + grid_sample vs the oracle's
    step-by-step composition (networks.py:133-135 / test_generator.py:206-213)."""
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    N = 2
    fh, fw = H // scale, W // scale
    src = torch.randn(N, C, H, W, generator=g)
    flow = torch.randn(N, fh, fw, 2, generator=g) * 3.0  # large enough to hit the border clamp
    if scale == 2:
        fup = O.resize_bilinear(flow.permute(0, 3, 1, 2), scale_factor=2).permute(0, 2, 3, 1)
        rh = rw = 0.5
    else:
        fup = O.resize_bilinear(flow.permute(0, 3, 1, 2), size=(H, W)).permute(0, 2, 3, 1)
        rh, rw = fh / H, fw / W
    nx, ny = (W / 2 - 1.0) / 2.0, (H / 2 - 1.0) / 2.0
    fnorm = torch.cat([fup[..., 0:1] / nx, fup[..., 1:2] / ny], 3)
    want = O.grid_sample_bilinear_border(src, fnorm + O.make_grid(N, H, W))
    out, fup_gpu = ops.flow_warp(_nhwc(ops, src), flow.cuda().contiguous(), H, W, rh, rw, nx, ny)
    _assert_close("flow_up", fup_gpu.cpu(), fup, 1e-6)
    _assert_close("flow_warp", ops.to_nchw(out), want, 2e-5)
    # property: bilinear/border sampling is a convex combination of source pixels
    o = ops.to_nchw(out).cpu()
    assert o.max() <= src.max() + 1e-5 and o.min() >= src.min() - 1e-5


CONV_CASES = [
    # name, sources(real C), Cout, k, stride, pad, H, W, extras
    ("first_4ch_s2", [4], 96, 3, 2, 1, 32, 24, {}),
    ("3x3_96", [96], 96, 3, 1, 1, 16, 12, {"bn": True, "act": "relu"}),
    ("3x3_res_relu", [32], 192, 3, 1, 1, 16, 12, {"bn": True, "act": "relu", "res": True}),
    ("1x1_cat3", [96, 16, 4], 13, 1, 1, 0, 16, 12, {"bias": True}),
    ("flow_cat2", [48, 48], 2, 3, 1, 1, 16, 12, {"bias": True, "res": True, "out_cs": 2}),
    ("wide_384", [64], 384, 3, 1, 1, 8, 6, {"bias": True, "act": "relu"}),
    ("cout_128", [20], 128, 3, 1, 1, 12, 8, {"bias": True}),
    ("cin9_pad12", [9], 40, 3, 1, 1, 16, 8, {"bias": True}),
    ("patchgan_4x4_s2", [10], 64, 4, 2, 2, 18, 14, {"bias": True, "act": "lrelu"}),
    ("up_cat_nearest", [32, 16], 64, 3, 1, 1, 16, 12, {"bias": True, "up0": True}),
    ("img_tanh", [32], 3, 3, 1, 1, 16, 12, {"bias": True, "act": "tanh"}),
    ("seg_down4_nearest", [8], 128, 3, 1, 1, 8, 6, {"bias": True, "act": "relu", "down0": 2}),
    ("big_m", [16], 32, 3, 1, 1, 96, 80, {"bias": True}),
]


def _run_conv_case(ops, case, impl, tile=None):
    name, real, cout, k, stride, pad, H, W, ex = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    N = 2
    xs = []
    for i, c in enumerate(real):
        if ex.get("up0") and i == 0:
            xs.append(torch.randn(N, c, H // 2, W // 2, generator=g))
        elif ex.get("down0") and i == 0:
            xs.append(torch.randn(N, c, H << ex["down0"], W << ex["down0"], generator=g))
        else:
            xs.append(torch.randn(N, c, H, W, generator=g))
    w = torch.randn(cout, sum(real), k, k, generator=g) * (1.0 / (sum(real) * k * k) ** 0.5)
    scale = (torch.rand(cout, generator=g) + 0.5) if ex.get("bn") else None
    shift = torch.randn(cout, generator=g) * 0.3 if (ex.get("bn") or ex.get("bias")) else None
    act = {"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH}.get(ex.get("act"), ops.ACT_NONE)
    # ---- oracle (torch CPU fp32)
    full = []
    for i, x in enumerate(xs):
        if ex.get("up0") and i == 0:
            x = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
        if ex.get("down0") and i == 0:
            x = O.resize_nearest(x, (H, W))
        full.append(x)
    xin = torch.cat(full, 1)
    if ex.get("pre"):
        xin = F.leaky_relu(xin, 0.2)
    ref = F.conv2d(xin, w, None, stride=stride, padding=pad)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    if shift is not None:
        ref = ref + shift.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if ex.get("res") else None
    if res is not None:
        ref = ref + res
    ref = {ops.ACT_RELU: F.relu, ops.ACT_LRELU: lambda t: F.leaky_relu(t, 0.2), ops.ACT_TANH: torch.tanh,
           ops.ACT_NONE: lambda t: t}[act](ref)
    # ---- HIP
    layer = ops.ConvLayer(w, real, "cuda", scale=scale, shift=shift, stride=stride, pad=pad, act=act, name=name)
    srcs = []
    for i, x in enumerate(xs):
        a = _nhwc(ops, x)
        up = 1 if (ex.get("up0") and i == 0) else (-ex["down0"] if (ex.get("down0") and i == 0) else 0)
        srcs.append((a, up, ops.ACT_LRELU if ex.get("pre") else ops.ACT_NONE))
    out = None
    Ho, Wo = ref.shape[2:]
    if ex.get("out_cs"):
        out = ops.Act(torch.empty((N, Ho, Wo, ex["out_cs"]), device="cuda"), cout)
    res_act = None
    if res is not None:
        if ex.get("out_cs"):
            res_act = ops.Act(res.permute(0, 2, 3, 1).contiguous().cuda(), cout)
        else:
            res_act = _nhwc(ops, res)
    old = {k_: os.environ.get(k_) for k_ in ("HRV_CONV_IMPL", "HRV_CONV_TILE")}
    try:
        os.environ["HRV_CONV_IMPL"] = impl
        if tile is not None:
            os.environ["HRV_CONV_TILE"] = str(tile)
        else:
            os.environ.pop("HRV_CONV_TILE", None)
        o = layer(srcs, out=out, residual=res_act, H=H, W=W)
        torch.cuda.synchronize()
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    if ex.get("out_cs"):
        got = o.t[..., :cout].permute(0, 3, 1, 2).contiguous()
    else:
        got = ops.to_nchw(o)
        assert (o.t[..., cout:] == 0).all(), "pad channels must stay zero"
    return got, ref


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_naive_device_crosscheck(case):
    ops = _ops()
    got, ref = _run_conv_case(ops, case, "naive")
    _assert_close("conv_naive_" + case[0], got, ref, 2e-5)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_mfma_default_tile(case):
    ops = _ops()
    got, ref = _run_conv_case(ops, case, "mfma")
    _assert_close("conv_mfma_" + case[0], got, ref, 2e-5)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("tile", list(range(8)))
def test_conv_mfma_every_tile_config_and_variant(tile, variant):
    ops = _ops()
    old = os.environ.get("HRV_CONV_VARIANT")
    os.environ["HRV_CONV_VARIANT"] = str(variant)
    from hr_viton_amd import _lib as _hl; _hl.reload_env()
    try:
        for case in (CONV_CASES[2], CONV_CASES[3], CONV_CASES[4], CONV_CASES[9], CONV_CASES[12]):
            got, ref = _run_conv_case(ops, case, "mfma", tile=tile)
            _assert_close(f"conv_mfma_t{tile}_v{variant}_" + case[0], got, ref, 2e-5)
    finally:
        if old is None:
            os.environ.pop("HRV_CONV_VARIANT", None)
        else:
            os.environ["HRV_CONV_VARIANT"] = old


@pytest.mark.parametrize("real,H,W", [([48, 48], 16, 12), ([384, 384], 8, 6), ([8], 5, 7)])
def test_tapconv_small_cout(real, H, W):
    """flow_conv (Cin -> 2, 3x3): taps-as-channels 1x1 on the MFMA engine + hrv_tapsum_nhwc_f32."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    N = 2
    xs = [torch.randn(N, c, H, W, generator=g) for c in real]
    w = torch.randn(2, sum(real), 3, 3, generator=g) * (1.0 / (sum(real) * 9) ** 0.5)
    b = torch.randn(2, generator=g)
    res = torch.randn(N, H, W, 2, generator=g)
    want = F.conv2d(torch.cat(xs, 1), w, b, padding=1).permute(0, 2, 3, 1) + res
    layer = ops.TapConvLayer(w, real, "cuda", bias=b, name="flow")
    out = ops.Act(torch.empty((N, H, W, 2), device="cuda"), 2)
    layer([_nhwc(ops, x) for x in xs], out=out, residual=ops.Act(res.cuda().contiguous(), 2))
    _assert_close("tapconv", out.t.cpu(), want, 2e-5)


@pytest.mark.parametrize("splitk", [0, 2, 5])
def test_conv_split_k(splitk):
    """Small-M / large-K layers split their K range over several blocks (deterministic 2-stage)."""
    ops = _ops()
    old = os.environ.get("HRV_CONV_SPLITK")
    os.environ["HRV_CONV_SPLITK"] = str(splitk)
    try:
        for case in (CONV_CASES[2], CONV_CASES[4], CONV_CASES[5], CONV_CASES[9], CONV_CASES[3]):
            for variant in (0, 1):
                os.environ["HRV_CONV_VARIANT"] = str(variant)
                from hr_viton_amd import _lib as _hl; _hl.reload_env()
                got, ref = _run_conv_case(ops, case, "mfma")
                _assert_close(f"conv_splitk{splitk}_v{variant}_" + case[0], got, ref, 2e-5)
                got2, _ = _run_conv_case(ops, case, "mfma")
                assert torch.equal(got, got2), "split-K must be deterministic"
    finally:
        os.environ.pop("HRV_CONV_VARIANT", None)
        if old is None:
            os.environ.pop("HRV_CONV_SPLITK", None)
        else:
            os.environ["HRV_CONV_SPLITK"] = old


BF16_CASES = [
    # name, cins, cout, k, stride, pad, H, W, extras
    ("bf_3x3", [64], 96, 3, 1, 1, 16, 12, {"bias": True, "act": "relu"}),
    ("bf_res", [32], 128, 3, 1, 1, 16, 12, {"bn": True, "res": True, "act": "lrelu"}),
    ("bf_cat_up", [32, 16], 64, 3, 1, 1, 16, 12, {"bias": True, "up0": True}),
    ("bf_1x1", [80], 32, 1, 1, 0, 12, 8, {"bn": True}),
    ("bf_cin9", [9], 16, 3, 1, 1, 16, 8, {"bias": True}),
    ("bf_img3", [32], 3, 3, 1, 1, 16, 12, {"bias": True, "act": "tanh"}),
    ("bf_down", [8], 128, 3, 1, 1, 8, 6, {"bias": True, "act": "relu", "down0": 2}),
    ("bf_smallM_splitk", [256], 256, 3, 1, 1, 4, 3, {"bias": True}),
]


@pytest.mark.parametrize("cfg", [None, 8, 9, 10, 11, 12, 13, 14, 15],
                         ids=["auto", "cfg8_rb128", "cfg9_rb128", "cfg10_256x128", "cfg11_128x256", "cfg12_8waves",
                              "cfg13_8waves_3stage", "cfg14_loader_waves_256x128", "cfg15_loader_waves_128x128"])
@pytest.mark.parametrize("case", BF16_CASES, ids=[c[0] for c in BF16_CASES])
def test_conv_bf16_engine(case, cfg):
    """bf16 storage / fp32 accumulate engine (v_mfma_f32_32x32x16_bf16) vs the fp32 oracle evaluated on the
    bf16-rounded operands: what remains is accumulation order + the final bf16 rounding (2^-8 relative)."""
    ops = _ops()
    name, real, cout, k, stride, pad, H, W, ex = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    N = 2
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    xs = []
    for i, c in enumerate(real):
        shp = (N, c, H // 2, W // 2) if (ex.get("up0") and i == 0) else \
            ((N, c, H << ex["down0"], W << ex["down0"]) if (ex.get("down0") and i == 0) else (N, c, H, W))
        xs.append(rb(torch.randn(*shp, generator=g)))
    w = rb(torch.randn(cout, sum(real), k, k, generator=g) * (1.0 / (sum(real) * k * k) ** 0.5))
    scale = (torch.rand(cout, generator=g) + 0.5) if ex.get("bn") else None
    shift = torch.randn(cout, generator=g) * 0.3 if (ex.get("bn") or ex.get("bias")) else None
    act = {"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH}.get(ex.get("act"), ops.ACT_NONE)
    full = []
    for i, x in enumerate(xs):
        if ex.get("up0") and i == 0:
            x = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
        if ex.get("down0") and i == 0:
            x = O.resize_nearest(x, (H, W))
        full.append(x)
    ref = F.conv2d(torch.cat(full, 1), w, None, stride=stride, padding=pad)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    if shift is not None:
        ref = ref + shift.view(1, -1, 1, 1)
    res = rb(torch.randn(ref.shape, generator=g)) if ex.get("res") else None
    if res is not None:
        ref = ref + res
    ref = {ops.ACT_RELU: F.relu, ops.ACT_LRELU: lambda t: F.leaky_relu(t, 0.2), ops.ACT_TANH: torch.tanh,
           ops.ACT_NONE: lambda t: t}[act](ref)
    layer = ops.ConvLayer(w, real, "cuda", scale=scale, shift=shift, stride=stride, pad=pad, act=act, name=name, bf16=True)
    srcs = []
    for i, x in enumerate(xs):
        up = 1 if (ex.get("up0") and i == 0) else (-ex["down0"] if (ex.get("down0") and i == 0) else 0)
        srcs.append((ops.to_nhwc(x.cuda(), bf16=True), up, ops.ACT_NONE))
    res_act = ops.to_nhwc(res.cuda(), bf16=True) if res is not None else None
    o = layer(srcs, residual=res_act, H=H, W=W, cfg=cfg)
    got = ops.to_nchw(o)
    assert o.t.dtype == torch.bfloat16 and (o.t[..., cout:].float() == 0).all()
    _assert_close("conv_bf16_" + name, got, ref, 1e-2)


@pytest.mark.parametrize("case", [("p128", 128, 128, 32, 48, {}),
                                  ("p128_res_relu_f32out", 128, 256, 20, 24, {"res": True, "act": "relu", "bias": True}),
                                  ("p256_partial_tiles", 256, 128, 18, 40, {"bn": True}),
                                  ("p384_cout100", 384, 100, 16, 16, {"bias": True, "act": "lrelu"})],
                         ids=lambda c: c[0])
def test_conv_bf16_patch_mode(case):
    """tile_cfg 16 / 17 / 18: 3x3 stride-1 convolutions whose pixel tile (16x16, 8x16 x 128 columns, 8x16 x 64 columns)
    keeps its halo patch resident in LDS (the activation is read from L2 once instead of once per tap).  Same math as the gather tiles: vs the fp32 oracle on
    the bf16-rounded operands, and vs cfg 8 (bit-identical for Cin = 128, where the K order is the same)."""
    ops = _ops()
    name, cin, cout, H, W, ex = case
    g = torch.Generator().manual_seed(cin + cout + H)
    N = 2
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
    x = rb(torch.randn(N, cin, H, W, generator=g))
    w = rb(torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (cin * 9) ** 0.5))
    scale = (torch.rand(cout, generator=g) + 0.5) if ex.get("bn") else None
    shift = torch.randn(cout, generator=g) * 0.3 if (ex.get("bn") or ex.get("bias")) else None
    act = {"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU}.get(ex.get("act"), ops.ACT_NONE)
    ref = F.conv2d(x, w, None, padding=1)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    if shift is not None:
        ref = ref + shift.view(1, -1, 1, 1)
    res = rb(torch.randn(ref.shape, generator=g)) if ex.get("res") else None
    if res is not None:
        ref = ref + res
    ref = {ops.ACT_RELU: F.relu, ops.ACT_LRELU: lambda t: F.leaky_relu(t, 0.2), ops.ACT_NONE: lambda t: t}[act](ref)
    f32out = "f32out" in name
    layer = ops.ConvLayer(w, [cin], "cuda", scale=scale, shift=shift, pad=1, act=act, name=name, bf16=True, out_f32=f32out)
    xa = ops.to_nhwc(x.cuda(), bf16=True)
    ra = ops.to_nhwc(res.cuda(), bf16=True) if res is not None else None
    outs = {}
    for cfg in (8, 16, 17, 18):
        o = layer([(xa, 0, ops.ACT_NONE)], residual=ra, cfg=cfg)
        assert o.t.dtype == (torch.float32 if f32out else torch.bfloat16)
        outs[cfg] = ops.to_nchw(o)
        _assert_close(f"conv_bf16_patch_{name}_cfg{cfg}", outs[cfg], ref, 1e-2 if not f32out else 2e-5)
    assert (outs[8] - outs[17]).abs().max() <= 2 ** -7 * ref.abs().max()
    assert (outs[8] - outs[18]).abs().max() <= 2 ** -7 * ref.abs().max()
    d = (outs[8] - outs[16]).abs()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/patch_mode_diff.txt", "a") as f:
        f.write(f"{name}: max|cfg8-cfg16|={d.max().item():.3e} differing={int((d > 0).sum())}/{d.numel()} "
                f"max|ref|={ref.abs().max().item():.3f}\n")
    assert d.max() <= 2 ** -7 * ref.abs().max()
    # not eligible: patch mode refuses instead of computing something else
    bad = ops.ConvLayer(rb(torch.randn(64, 96, 3, 3, generator=g)), [96], "cuda", pad=1, name="bad", bf16=True)
    with pytest.raises(Exception):
        bad([(ops.to_nhwc(torch.zeros(1, 96, 16, 16).cuda(), bf16=True), 0, ops.ACT_NONE)], cfg=16)


@pytest.mark.parametrize("N,H,W,C", [(2, 37, 29, 80), (1, 130, 70, 144), (1, 9, 7, 1040)])
def test_dual_noise_instnorm_stats_are_bit_identical_to_two_passes(N, H, W, C):
    """ops.instnorm_stats2: the statistics of x + z_a*ns_a and x + z_b*ns_b (norm_s / norm_0 of a learned-shortcut SPADEResBlock,
    network_generator.py:158-166) from ONE pass over x -- same summation order as the single kernel, bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    x = ops.Act((torch.randn(N, H, W, C, generator=g) * 2 + 0.5).cuda(), C)
    za, zb = torch.randn(N, W, H, 1, generator=g).cuda(), torch.randn(N, W, H, 1, generator=g).cuda()
    na, nb = (torch.randn(C, generator=g) * 0.3).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    (ma, ra), (mb, rb) = ops.instnorm_stats2(x, za, na, zb, nb)
    wa, wb = ops.instnorm_stats(x, za, na), ops.instnorm_stats(x, zb, nb)
    torch.cuda.synchronize()
    assert torch.equal(ma, wa[0]) and torch.equal(ra, wa[1]) and torch.equal(mb, wb[0]) and torch.equal(rb, wb[1])
    v = x.t.permute(0, 3, 1, 2) + za.permute(0, 3, 2, 1) * na.view(1, -1, 1, 1)
    assert float((ma - v.mean((2, 3))).abs().max()) < 1e-5 and float((ra - (v.var((2, 3), unbiased=False) + 1e-5).rsqrt()).abs().max()) < 1e-4
