#!/usr/bin/env python3
"""Phase timeline of csrc/conv_p2.hip (two persistent blocks per CU): every (tile, block) logs wall-clock stamps at tile start,
main-loop start, main-loop end, epilogue end and the CU it ran on (hrv_diag_set_tlog).  Reports the phase durations and, per CU,
what the two co-resident blocks do AT THE SAME TIME: both in a main loop, one in a main loop, none (matrix pipes idle).
    python tools/p2_timeline.py            (via gpurun; HRV_* switches apply)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import _lib, ops, train_ops as T  # noqa: E402

# (name, kind, Cin (K), columns, N, H, W)
CASES = [("vgg.12 256->256 @256x192 N8 fwd", "fwd", 256, 256, 8, 256, 192),
         ("vgg.7 128->128 @512x384 N8 fwd", "fwd", 128, 128, 8, 512, 384),
         ("vgg.12 dgrad N4 (ReLU mask)", "dgrad", 256, 256, 4, 256, 192),
         ("up_4.norm_0.gb.dgrad K=160 -> 128 N4 (ReLU mask of actv)", "gb", 160, 128, 4, 1024, 768),
         ("up_3.norm_0.gb.dgrad K=288 -> 128 N4", "gb", 288, 128, 4, 512, 384),
         ("up_4.conv_0 80->32 fp32 out N4", "thin", 80, 32, 4, 1024, 768)]


def measure(name, kind, cin, cols, N, H, W):
    torch.manual_seed(0)
    if kind == "gb":
        dgb = ops.Act(torch.randn(N, H, W, cin, device="cuda").to(torch.bfloat16), cin)
        wg, wb = torch.randn(cin // 2, cols, 3, 3, device="cuda") * 0.05, torch.randn(cin // 2, cols, 3, 3, device="cuda") * 0.05
        actv = ops.Act(torch.relu(torch.randn(N, H, W, cols, device="cuda")).to(torch.bfloat16), cols)
        out = ops.Act(torch.empty(N, H, W, cols, device="cuda", dtype=torch.bfloat16), cols)
        pk = T.conv_p2_pack(2, wg, wb, cin, cols)
        run = lambda: T.conv_p2(dgb, pk, cols, out, mask=actv, mask_slope=0.0, name=name)      # noqa: E731
    elif kind == "thin":
        x = ops.Act(torch.randn(N, H, W, cin, device="cuda").to(torch.bfloat16), cin)
        w = torch.randn(cols, cin, 3, 3, device="cuda") * 0.05
        out = ops.Act(torch.empty(N, H, W, cols, device="cuda"), cols)
        pk = T.conv_p2_pack(0, w, None, cin, cols)
        run = lambda: T.conv_p2(x, pk, cols, out, name=name)      # noqa: E731
    else:
        x = ops.Act(torch.relu(torch.randn(N, H, W, cin, device="cuda")).to(torch.bfloat16), cin)
        w = torch.randn(cols, cin, 3, 3, device="cuda") * 0.03
        out = ops.Act(torch.empty(N, H, W, cols if kind == "fwd" else cin, device="cuda", dtype=torch.bfloat16), cols if kind == "fwd" else cin)
        if kind == "fwd":
            pk = T.conv_p2_pack(0, w, None, cin, cols)
            b = torch.zeros(cols, device="cuda")
            run = lambda: T.conv_p2(x, pk, cols, out, bias=b, act=ops.ACT_RELU, name=name)      # noqa: E731
        else:
            dy = ops.Act(torch.randn(N, H, W, cols, device="cuda").to(torch.bfloat16), cols)
            pk = T.conv_p2_pack(1, w, None, cols, cin)
            run = lambda: T.conv_p2(dy, pk, cin, out, mask=x, mask_slope=0.0, name=name)      # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        s.record(); run(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    tiles = N * ((H + 15) // 16) * ((W + 15) // 16)
    tlog = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    lib = _lib.load()
    _lib.check(lib.hrv_diag_set_tlog(tlog.data_ptr(), tiles), "hrv_diag_set_tlog")
    run()
    torch.cuda.synchronize()
    _lib.check(lib.hrv_diag_set_tlog(None, 0), "hrv_diag_set_tlog")
    t = tlog.cpu().view(tiles, 8)
    t = t[t[:, 3] > 0]
    t0 = int(t[:, 0].min())
    us = lambda v: (v - t0).double() / 100.0          # noqa: E731   wall_clock64: 100 MHz
    st, lp, ep, en, ei = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(t[:, 6])
    npass = (cols if kind != "dgrad" else cin)
    npass = (npass + 127) // 128
    fl = 2.0 * N * H * W * cin * cols * 9
    med = sorted(ts)[len(ts) // 2]
    print(f"{name}: {len(t)}/{tiles} tiles logged, launch {med * 1e3:.0f} us = {fl / med / 1e9:.0f} TF/s (no log), device span with log {float(en.max()):.0f} us, "
          f"{npass} column pass(es) per tile")
    # (with several passes per tile the stamps are: [0] start of pass 0, [1] its loop start, [2] loop end of the LAST pass, [3] tile end)
    for nm, d in (("head (patch 0 + 2 k-tiles landed)", lp - st), ("loop start of first .. loop end of last pass", ep - lp),
                  ("last epilogue: issued (stores in flight)", ei - ep), ("last epilogue: drained (vmcnt 0; log only)", en - ei), ("tile", en - st)):
        print(f"   {nm:46s} mean {float(d.mean()):7.2f} us  median {float(d.median()):7.2f}  p90 {float(d.quantile(0.9)):7.2f}")
    pk = t[:, 7]
    js = [((pk >> (16 * j)) & 0xFFFF).double() / 100.0 for j in range(4)]
    print("   epilogue of wave 0, time since its barrier after column tile j = 0..3 done (stores issued): " +
          "  ".join("%.2f" % float(v.mean()) for v in js) + " us (means)")
    # gaps between a block's consecutive tiles (end of one -> start stamp of the next: the barrier all four waves' epilogues meet at)
    blk = t[:, 5]
    order = torch.argsort(blk * (1 << 40) + t[:, 0])
    sb, ss, se = blk[order], st[order], en[order]
    same = sb[1:] == sb[:-1]
    gaps = (ss[1:] - se[:-1])[same]
    if gaps.numel():
        print(f"   gap between a block's tiles (tile end of wave 0 -> next tile's start, i.e. the slowest wave's epilogue): mean {float(gaps.mean()):.2f} us  "
              f"median {float(gaps.median()):.2f}  p90 {float(gaps.quantile(0.9)):.2f}")
    key = ((t[:, 4] >> 32) << 16) | (((t[:, 4] & 0xFFFFFFFF) >> 8) & 0xFF)       # (XCC id; cu_id | sh_id | se_id = bits 8..15 of HW_ID)
    cus = key.unique().tolist()
    both = one = none = span = 0.0
    for k in cus:
        m = key == k
        ev = []
        for a, b in zip(lp[m].tolist(), ep[m].tolist()):
            ev.append((a, 1)); ev.append((b, -1))
        ev.sort()
        lo, hi = float(st[m].min()), float(en[m].max())
        span += hi - lo
        cur, prev = 0, lo
        for x, d in ev:
            dt = x - prev
            if cur >= 2: both += dt
            elif cur == 1: one += dt
            else: none += dt
            prev, cur = x, cur + d
        none += hi - prev
    nblk = sorted(int(t[key == k][:, 5].unique().numel()) for k in cus)
    print(f"   {len(cus)} CUs (blocks per CU {nblk[0]}..{nblk[-1]}); of each CU's busy span: BOTH blocks in a main loop {100 * both / span:.1f} %, "
          f"ONE {100 * one / span:.1f} %, NONE (matrix pipes idle) {100 * none / span:.1f} %")
    k0 = cus[0]
    m = key == k0
    rows = sorted(zip(st[m].tolist(), lp[m].tolist(), ep[m].tolist(), en[m].tolist(), t[m][:, 5].tolist()))[:8]
    print("   first tiles of one CU (start, loop, loop end, end [us]; block):")
    for r in rows:
        print("     %8.2f %8.2f %8.2f %8.2f   b%d" % r)
    sys.stdout.flush()


def main():
    T.MMA_BF16[0] = True
    sel = [int(a) for a in sys.argv[1:]] or range(len(CASES))
    for i in sel:
        measure(*CASES[i])


if __name__ == "__main__":
    main()
