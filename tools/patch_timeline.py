#!/usr/bin/env python3
"""Phase timeline of the persistent patch-tile convolution (diag): every tile logs wall-clock stamps at tile start, main
loop start (patch + first weight tiles landed), main loop end and epilogue end, plus the CU it ran on
(hrv_diag_set_tlog).  Reports the phase durations and, per CU, how much of the co-resident blocks' epilogue / prologue time
is covered by another block's main loop.      python tools/patch_timeline.py [layer_idx] [cfg]      (via gpurun)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import ops  # noqa: E402

LAYERS = [("spade.gb 128->2x64 3x3 @1024x768 N1", 128, 128, 1, 1024, 768), ("gb 128->2x128 @512x384 N4", 128, 256, 4, 512, 384),
          ("blk 256->256 @256x192 N4", 256, 256, 4, 256, 192), ("up_4 gb 128->2x64 @1024x768 N4", 128, 128, 4, 1024, 768)]


def union(iv):
    out = []
    for a, b in sorted(iv):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def inter(A, B):
    i = j = tot = 0
    while i < len(A) and j < len(B):
        lo, hi = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
        tot += max(0, hi - lo)
        if A[i][1] < B[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    li = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 17
    name, cin, cout, N, H, W = LAYERS[li]
    g = torch.Generator().manual_seed(0)
    spade = len(sys.argv) > 3 and sys.argv[3] == "spade"
    if spade:       # the TRAINING SPADE layer: fused gamma|beta conv + modulate, x fp32, bf16 result, (1+gamma) saved
        from argparse import Namespace
        from hr_viton_amd import train_ops as T
        from hr_viton_amd.gen_train import SpadeT
        from hr_viton_amd.network_generator import SPADENorm
        T.MMA_BF16[0] = True
        Cc = cout // 2
        norm = SPADENorm(Namespace(), "aliasinstance", Cc, 7).cuda()
        st_ = SpadeT(norm, ops.ACT_LRELU, name)
        xa = ops.Act(torch.randn(N, H, W, Cc, device="cuda"), Cc)
        if len(sys.argv) > 4 and sys.argv[4] == "slice":      # actv as the middle third of the block-wide tensor, as in the plan
            av = ops.Act(torch.relu(torch.randn(N, H, W, 3 * cin, device="cuda")).to(torch.bfloat16), cin, cin)
            name += " [actv slice of 384]"
        else:
            av = ops.Act(torch.relu(torch.randn(N, H, W, cin, device="cuda")).to(torch.bfloat16), cin)
        z = torch.randn(N, W, H, 1, device="cuda")
        run = lambda: st_.forward(xa, av, z, save=True)         # noqa: E731
        name += " [training SPADE epilogue]"
    elif len(sys.argv) > 3 and sys.argv[3] == "dgrad":      # gamma|beta data gradient: dY = [dgamma|dbeta] bf16, ReLU mask of actv, slices of 384
        from hr_viton_amd import train_ops as T
        T.MMA_BF16[0] = True
        dy = ops.Act(torch.randn(N, H, W, cout, device="cuda").to(torch.bfloat16), cout)
        wg = (torch.randn(cout // 2, cin, 3, 3, device="cuda") * 0.05, torch.randn(cout // 2, cin, 3, 3, device="cuda") * 0.05)
        actv_all = torch.relu(torch.randn(N, H, W, 3 * cin, device="cuda")).to(torch.bfloat16)
        dact_all = torch.empty_like(actv_all)
        run = lambda: T.conv_dgrad(dy, wg, H, W, 1, 1, act_mask=ops.Act(actv_all, cin, cin), slope=0.0,     # noqa: E731
                                   out=ops.Act(dact_all, cin, cin), name="gb.dgrad")
        name += " [gamma|beta data gradient]"
    else:
        x = ops.to_nhwc(torch.randn(N, cin, H, W, generator=g).cuda(), bf16=True)
        layer = ops.ConvLayer(torch.randn(cout, cin, 3, 3, generator=g) * 0.05, [cin], "cuda", shift=torch.randn(cout, generator=g),
                              stride=1, pad=1, act=ops.ACT_RELU, name=name, bf16=True)
        out = ops.alloc(N, H, W, cout, "cuda", bf16=True)
        os.environ["HRV_CONV_TILE"] = str(cfg)
        run = lambda: layer([x], out=out)                       # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    tiles = N * ((H + 7) // 8) * ((W + 15) // 16) * (((cin if (len(sys.argv) > 3 and sys.argv[3] == 'dgrad') else cout) + 127) // 128)
    tlog = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    from hr_viton_amd import _lib
    _lib.check(_lib.load().hrv_diag_set_tlog(tlog.data_ptr(), tiles), "hrv_diag_set_tlog")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run()
    e.record()
    torch.cuda.synchronize()
    _lib.check(_lib.load().hrv_diag_set_tlog(None, 0), "hrv_diag_set_tlog")
    t = tlog.cpu().view(tiles, 8)
    ok = t[:, 3] > 0
    t = t[ok]
    t0 = int(t[:, 0].min())
    us = lambda v: (v - t0).double() / 100.0          # noqa: E731   wall_clock64: 100 MHz
    st, lp, ep, en = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
    print(f"{name} cfg {cfg}: {int(ok.sum())}/{tiles} tiles logged, launch {s.elapsed_time(e) * 1e3:.0f} us (host events), "
          f"device span {float(en.max()):.0f} us")
    for nm, d in (("prologue (patch + first weights)", lp - st), ("main loop", ep - lp), ("epilogue", en - ep), ("tile", en - st)):
        print(f"   {nm:34s} mean {float(d.mean()):7.2f} us  median {float(d.median()):7.2f}  p90 {float(d.quantile(0.9)):7.2f}")
    key = ((t[:, 4] >> 32) << 16) | ((t[:, 4] & 0xFFFFFFFF) >> 8 & 0xFF)
    cus = key.unique()
    tot_main = tot_epi = tot_pro = cov_epi = cov_pro = busy_any = span = 0.0
    per_cu_blocks = []
    for k in cus.tolist():
        m = key == k
        per_cu_blocks.append(int(t[m][:, 5].unique().numel()))
        mains = union(list(zip(lp[m].tolist(), ep[m].tolist())))
        epis = list(zip(ep[m].tolist(), en[m].tolist()))
        pros = list(zip(st[m].tolist(), lp[m].tolist()))
        tot_main += sum(b - a for a, b in mains)
        tot_epi += sum(b - a for a, b in epis)
        tot_pro += sum(b - a for a, b in pros)
        cov_epi += sum(inter([[a, b]], mains) for a, b in epis)
        cov_pro += sum(inter([[a, b]], mains) for a, b in pros)
        span += float(en[m].max() - st[m].min())
    print(f"   {len(cus)} CUs, blocks per CU {min(per_cu_blocks)}..{max(per_cu_blocks)}; per CU: some block in its main loop "
          f"{100 * tot_main / span:.1f} % of the time; epilogue time covered by another block's main loop {100 * cov_epi / tot_epi:.1f} %, "
          f"prologue time covered {100 * cov_pro / tot_pro:.1f} %")
    k0 = cus.tolist()[0]
    m = key == k0
    rows = sorted(zip(st[m].tolist(), lp[m].tolist(), ep[m].tolist(), en[m].tolist(), t[m][:, 5].tolist()))[:10]
    print("   first tiles of one CU (start, loop, epi, end [us]; block):")
    for r in rows:
        print("     %8.2f %8.2f %8.2f %8.2f   b%d" % r)


if __name__ == "__main__":
    main()
