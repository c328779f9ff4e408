"""CPU (no GPU needed): the C-ABI library loads, exports every symbol the header
declares, and its HOST-side logic (tile pick, weight packing / K-tile order) is
right.  No device compute is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from hr_viton_amd import build
        build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    from hr_viton_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "hrviton_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hrv_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hrv_version().startswith(b"hrviton-hip")


def test_struct_layout_matches_c(lib):
    from hr_viton_amd import _lib
    # sizes implied by the C declarations (LP64): see include/hrviton_hip.h
    assert C.sizeof(_lib.hrv_src_t) == 32
    assert C.sizeof(_lib.hrv_conv2d_t) == 40 + 4 * 32 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 16 + 32 + 8
    assert C.sizeof(_lib.hrv_spade_epi_t) == 8 + 16 + 32 + 8
    assert C.sizeof(_lib.hrv_flow_warp_t) == 8 + 24 + 8 + 16 + 16 + 8 + 8 + 8


def test_pick_tile(lib):
    for M, cout in [(3072, 768), (786432, 96), (3145728, 96), (786432, 2), (196608, 384), (100, 13), (50000, 1040)]:
        cfg = lib.hrv_conv2d_pick_tile(M, cout)
        bn, bm = lib.hrv_conv2d_tile_bn(cfg), lib.hrv_conv2d_tile_bm(cfg)
        assert bn in (32, 64, 96, 128) and bm == 128
        padded = -(-cout // bn) * bn
        assert padded - cout < bn
        # never waste more than the minimum achievable over the tile widths
        assert padded == min(-(-cout // b) * b for b in (32, 64, 96, 128))
    assert lib.hrv_conv2d_tile_bn(99) == -1


def _emulate_kernel_gemm(srcs, src_pad, packed, cout, cpad, KH, KW, stride, pad, Ho, Wo):
    """numpy re-enactment of the kernel's K loop: (tap) x (source) x (16-channel chunk)."""
    N, H, W = srcs[0].shape[:3]
    M = N * Ho * Wo
    out = np.zeros((M, cpad), dtype=np.float64)
    chunks = [-(-c // 16) for c in src_pad]
    kt = 0
    ho, wo = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing="ij")
    for kh in range(KH):
        for kw in range(KW):
            hi = ho * stride - pad + kh
            wi = wo * stride - pad + kw
            ok = (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W)
            hic, wic = hi.clip(0, H - 1), wi.clip(0, W - 1)
            for s, x in enumerate(srcs):
                for ch in range(chunks[s]):
                    A = np.zeros((N, Ho, Wo, 16), dtype=np.float64)
                    c0 = ch * 16
                    c1 = min(c0 + 16, src_pad[s])
                    A[..., : c1 - c0] = x[:, hic, wic, c0:c1] * ok[None, :, :, None]
                    out += A.reshape(M, 16) @ packed[kt].astype(np.float64).T
                    kt += 1
    assert kt == packed.shape[0]
    return out[:, :cout]


@pytest.mark.parametrize("case", [
    dict(src=[4], cout=8, k=3, stride=2, pad=1, H=10, W=8),
    dict(src=[96, 16, 4], cout=13, k=1, stride=1, pad=0, H=6, W=5),
    dict(src=[24, 40], cout=2, k=3, stride=1, pad=1, H=7, W=9),
    dict(src=[9], cout=20, k=3, stride=1, pad=1, H=5, W=6),     # real 9 channels padded to 12
    dict(src=[10], cout=70, k=4, stride=2, pad=2, H=9, W=11),   # PatchGAN geometry
])
def test_pack_weight_matches_conv(lib, case):
    g = torch.Generator().manual_seed(0)
    real = case["src"]
    padc = [(c + 3) // 4 * 4 for c in real]
    cout, k = case["cout"], case["k"]
    N, H, W = 2, case["H"], case["W"]
    xs = [torch.randn(N, c, H, W, generator=g) for c in real]
    w = torch.randn(cout, sum(real), k, k, generator=g)
    ref = F.conv2d(torch.cat(xs, 1), w, stride=case["stride"], padding=case["pad"])
    Ho, Wo = ref.shape[2:]
    for cfg in range(8):
        bn = lib.hrv_conv2d_tile_bn(cfg)
        cpad = -(-cout // bn) * bn
        n = len(real)
        srcC = (C.c_int32 * n)(*padc)
        srcR = (C.c_int32 * n)(*real)
        elems = lib.hrv_conv2d_packed_elems(cout, k, k, n, srcC, cfg)
        chunks = sum(-(-c // 16) for c in padc)
        assert elems == k * k * chunks * cpad * 16
        buf = torch.full((elems,), float("nan"))
        wc = w.contiguous()
        rc = lib.hrv_conv2d_pack_weight_f32(wc.data_ptr(), cout, k, k, n, srcC, srcR, cfg, buf.data_ptr())
        assert rc == 0
        assert torch.isfinite(buf).all()
        packed = buf.view(k * k * chunks, cpad, 16).numpy()
        srcs_nhwc = []
        for x, pc in zip(xs, padc):
            a = np.zeros((N, H, W, pc), dtype=np.float32)
            a[..., : x.shape[1]] = x.permute(0, 2, 3, 1).numpy()
            srcs_nhwc.append(a)
        got = _emulate_kernel_gemm(srcs_nhwc, padc, packed, cout, cpad, k, k, case["stride"], case["pad"], Ho, Wo)
        want = ref.permute(0, 2, 3, 1).reshape(-1, cout).double().numpy()
        assert np.abs(got - want).max() < 1e-4, (cfg, np.abs(got - want).max())


def test_bad_args_fail_loudly(lib):
    from hr_viton_amd import _lib
    d = _lib.hrv_conv2d_t()
    assert lib.hrv_conv2d_nhwc_f32(C.byref(d), None) == -1
    assert b"conv2d" in lib.hrv_last_error()
    w = _lib.hrv_flow_warp_t()
    assert lib.hrv_flow_warp_nhwc_f32(C.byref(w), None) == -1


def test_cpu_tensors_raise():
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops
    with pytest.raises(ops.HrvError):
        ops.to_nhwc(torch.zeros(1, 4, 8, 8))


def test_tile_table_and_new_struct_fields(lib):
    """Tile ids 8-18 (128-byte K-tile rows of the bf16 engine incl. the 8-wave, loader-wave and patch-mode tiles):
    column / row extents the packers and the Python planners rely on; the structs that gained fields."""
    from hr_viton_amd import _lib
    want = {8: (128, 128), 9: (128, 64), 10: (256, 128), 11: (128, 256), 12: (256, 128), 13: (256, 128), 14: (256, 128),
            15: (128, 128), 16: (256, 128), 17: (128, 128), 18: (128, 64), 19: (256, 64)}
    for cfg, (bm, bn) in want.items():
        assert (lib.hrv_conv2d_tile_bm(cfg), lib.hrv_conv2d_tile_bn(cfg)) == (bm, bn), cfg
    assert lib.hrv_conv2d_tile_bn(20) == -1
    d = _lib.hrv_norm_bwd_t()
    assert hasattr(d, "dgb_bf16") and hasattr(d, "out_bf16") and d.dgb_bf16 == 0 and d.out_bf16 == 0
    # a descriptor that asks for the patch tile without being a 3x3 'same' bf16 convolution is refused, not rerouted
    c = _lib.hrv_conv2d_t()
    c.tile_cfg = 17
    assert lib.hrv_conv2d_nhwc_bf16(C.byref(c), None) == -1


def test_patch_tile_selection():
    """ops.patch_tile: which layers take the LDS-resident-patch tiles (no GPU needed: pure host logic)."""
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import ops
    pt = ops.patch_tile
    # SPADE gamma|beta convs of the released 1024x768 generator, batch 4: C = 64 -> 128 columns, C = 80 -> 192
    assert pt(True, 3, 3, 1, 1, 1, 0, 128, 128, 4, 1024, 768) == 17
    assert pt(True, 3, 3, 1, 1, 1, 0, 128, 192, 4, 1024, 768) == 18
    assert pt(True, 3, 3, 1, 1, 1, 0, 128, 2112, 4, 8, 6) == 0          # the 8x6 head: too few tiles to fill the chip
    assert pt(True, 3, 3, 1, 1, 1, 0, 128, 2112, 4, 32, 24) == 18       # 33 column tiles x 32 pixel tiles
    assert pt(True, 3, 3, 1, 1, 1, 0, 1024, 1024, 4, 64, 48) == 17
    # not a 3x3 stride-1 'same' conv / fp32 engine / several sources / resampled source / C % 128 / padded columns
    assert pt(True, 1, 1, 1, 0, 1, 0, 128, 128, 4, 1024, 768) == 0
    assert pt(True, 3, 3, 2, 1, 1, 0, 128, 128, 4, 1024, 768) == 0
    assert pt(False, 3, 3, 1, 1, 1, 0, 128, 128, 4, 1024, 768) == 0
    assert pt(True, 3, 3, 1, 1, 2, 0, 128, 128, 4, 1024, 768) == 0
    assert pt(True, 3, 3, 1, 1, 1, 1, 128, 128, 4, 1024, 768) == 0
    assert pt(True, 3, 3, 1, 1, 1, 0, 80, 128, 4, 1024, 768) == 0
    assert pt(True, 3, 3, 1, 1, 1, 0, 128, 13, 4, 1024, 768) == 0
    # wide patch tiles (conv_patchw.hip, tile_cfg 19): only where the caller's epilogue exists there (SPADE sites) and
    # there is at least one 16x16 tile per CU
    assert pt(True, 3, 3, 1, 1, 1, 0, 128, 192, 4, 1024, 768, wide=True) == 18     # opt-in
    os.environ["HRV_CONV_PATCHW"] = "1"
    try:
        assert pt(True, 3, 3, 1, 1, 1, 0, 128, 192, 4, 1024, 768, wide=True) == 19
        assert pt(True, 3, 3, 1, 1, 1, 0, 128, 576, 4, 256, 192, wide=True) == 19
        assert pt(True, 3, 3, 1, 1, 1, 0, 128, 1088, 4, 128, 96, wide=True) == 18     # 192 tiles: the 8x16 tiles
        assert pt(False, 3, 3, 1, 1, 1, 0, 128, 192, 4, 1024, 768, wide=True) == 0
    finally:
        del os.environ["HRV_CONV_PATCHW"]
    os.environ["HRV_CONV_PATCH"] = "0"
    try:
        assert pt(True, 3, 3, 1, 1, 1, 0, 128, 128, 4, 1024, 768) == 0
    finally:
        del os.environ["HRV_CONV_PATCH"]


def test_conv_s2_host_logic(lib):
    """csrc/conv_s2.hip / wgrad_s2.hip, host side only (round 6): descriptor layouts as the C header declares them, the size of the
    packed weight stream per mode ([pass][chunk][4 taps][column tile][2 k-steps][64 lanes][16 B]), which shapes the kernels serve."""
    from hr_viton_amd import _lib
    assert C.sizeof(_lib.hrv_conv_s2_t) == 152          # 5 x i32 + pad | ptr + 2 x i32 | 4 x i32 | 2 ptr | i32 + f32 | ptr + 3 x i32 + pad | x2 | ...
    assert C.sizeof(_lib.hrv_s2_pack_job_t) == 48
    pb = lib.hrv_conv_s2_packed_bytes
    assert pb(0, 64, 128) == 8 * 4 * 4 * 2048            # forward: 4 sub-pixels x 2 chunks, one 4-tile pass
    assert pb(0, 128, 256) == 2 * (16 * 4 * 4 * 2048)    # two passes
    assert pb(1, 128, 256) == 2 * (4 * 4 * 4 * 2048)     # data gradient: K = the forward's 128 outputs, columns = 4 phases x 64
    assert pb(2, 48, 64) == 2 * 4 * 2 * 2048             # model0 over cells: 1.5 chunks -> 2, a 2-tile pass
    assert pb(0, 192, 128) == 3 * pb(0, 64, 128)         # split operands: K triples
    assert pb(0, 60, 128) == -1 and pb(0, 64, 96) == -1 and pb(3, 64, 128) == -1      # K % 8, columns % 64, mode
    sup = lib.hrv_conv_s2_supported
    assert sup(0, 64, 128, 0, 8, 257, 193) == 1 and sup(0, 48, 128, 0, 8, 257, 193) == 0      # a forward chunk lies in one sub-pixel
    assert sup(1, 128, 256, 64, 4, 513, 385) == 1 and sup(1, 128, 256, 48, 4, 513, 385) == 0   # a column tile lies in one phase
    assert sup(0, 64, 128, 0, 1, 9, 9) == 0                                                     # too few tiles for the persistent grid
    ws = lib.hrv_conv2d_wgrad_s2_supported
    assert ws(128, 64, 64, 0, 128, 0, 8, 513, 385) == 1 and ws(256, 128, 128, 0, 256, 0, 8, 257, 193) == 1
    assert ws(128, 64, 192, 0, 128, 0, 8, 257, 193) == 1          # X as the hi third of a split tensor
    assert ws(96, 64, 64, 0, 96, 0, 8, 513, 385) == 0 and ws(128, 64, 64, 0, 128, 0, 1, 33, 33) == 0
