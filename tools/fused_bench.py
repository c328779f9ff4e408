#!/usr/bin/env python3
"""A/B of the fused SPADENorm forward (csrc/spade_fused.hip: conv_shared inside the gamma|beta kernel, two blocks per CU)
against the unfused pair (thin conv_shared launch writing actv + csrc/spade_gb.hip reading it) on the generator's norm shapes,
training forward (actv and (1 + gamma) saved) and no_grad forward; interleaved rounds in ONE process, median; plus the per-tile
phase timeline of the fused kernel (hrv_diag_set_tlog).      python tools/fused_bench.py [rounds]      (via gpurun)"""
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import _lib, ops, train_ops as T  # noqa: E402
from hr_viton_amd.gen_train import SpadeT  # noqa: E402
from hr_viton_amd.network_generator import SPADENorm  # noqa: E402

SHAPES = [("up_4.norm_0", 80, 4, 1024, 768, 0), ("up_4.norm_1", 32, 4, 1024, 768, 0), ("up_3.norm_0", 144, 4, 512, 384, 1),
          ("up_3.norm_1", 64, 4, 512, 384, 1), ("up_2.norm_0", 272, 4, 256, 192, 2), ("up_2.norm_1", 128, 4, 256, 192, 2)]


def timeline(run, tiles, label):
    tlog = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    _lib.check(_lib.load().hrv_diag_set_tlog(tlog.data_ptr(), tiles), "hrv_diag_set_tlog")
    run()
    torch.cuda.synchronize()
    _lib.check(_lib.load().hrv_diag_set_tlog(None, 0), "hrv_diag_set_tlog")
    t = tlog.cpu().view(tiles, 8)
    t = t[t[:, 3] > 0]
    if t.shape[0] == 0:
        print(f"   timeline {label}: no tiles logged")
        return
    t0 = int(t[:, 0].min())
    us = lambda v: (v - t0).double() / 100.0          # noqa: E731   wall_clock64: 100 MHz
    st, lp, ep, en, iss = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(t[:, 6])
    print(f"   timeline {label}: {t.shape[0]}/{tiles} tiles, device span {float(en.max()):.0f} us")
    for nm, d in (("prologue+csh0", lp - st), ("main loop(s)", ep - lp), ("epilogue", en - ep), ("  stores issued", iss - ep), ("tile", en - st)):
        print(f"      {nm:14s} mean {float(d.mean()):7.2f} us  median {float(d.median()):7.2f}  p90 {float(d.quantile(0.9)):7.2f}")


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    T.MMA_BF16[0] = True
    torch.manual_seed(0)
    for name, Cc, N, H, W, shift in SHAPES:
        norm = SPADENorm(Namespace(), "aliasinstance", Cc, 7).cuda()
        st = SpadeT(norm, ops.ACT_LRELU, name)
        x = ops.Act(torch.randn(N, H, W, Cc, device="cuda"), Cc)
        lab = torch.randint(0, 7, (N, H << shift, W << shift, 1), device="cuda")
        seg = ops.Act(torch.zeros(N, H << shift, W << shift, 8, device="cuda").scatter_(3, lab, 1.0).to(torch.bfloat16), 7)
        z = torch.randn(N, W, H, 1, device="cuda")
        actv_all = torch.empty(N, H, W, 384, device="cuda", dtype=torch.bfloat16)
        actv = ops.Act(actv_all, 128, 128)
        segx = ops.tap_expand(seg, shift, 3)

        def fused(save):
            st.forward(x, actv if save else None, z, save=save, fused=(seg, shift))

        def unfused(save):
            # one norm's share of the block's conv_shared launch (the thin kernel writes 128 of the block's 256 / 384 columns)
            wt, bt_ = T.shared_taps_prep([st.shared.wparam.data], [st.shared.bparam.data], 8)
            a = T.conv_forward_dev(wt, [(segx, 0)], 1, 0, shift=bt_, act=ops.ACT_RELU, out_bf16=True, name="cs")
            st.forward(x, a, z, save=save)
        fl = 2.0 * N * H * W * 2 * Cc * 128 * 9
        print(f"{name}: C={Cc} N={N} {H}x{W} label map x{1 << shift}  ({fl / 1e12:.3f} TFLOP per launch, gamma|beta only)")
        for save in (True, False):
            times = {"fused": [], "unfused": []}
            for rd in range(rounds + 2):
                for key, fn in (("fused", fused), ("unfused", unfused)):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    fn(save)
                    e.record()
                    torch.cuda.synchronize()
                    if rd >= 2:
                        times[key].append(s.elapsed_time(e))
            for key in ("fused", "unfused"):
                ts = sorted(times[key])
                med = ts[len(ts) // 2]
                print(f"   {'train fwd' if save else 'no_grad  '} {key:8s} median {med:7.3f} ms  min {ts[0]:7.3f}  {fl / (med * 1e-3) / 1e12:7.1f} TFLOP/s "
                      "(incl. stats + pack launches" + (", conv_shared launch)" if key == "unfused" else ")"))
        tiles = N * ((H + 15) // 16) * ((W + 15) // 16)
        timeline(lambda: fused(True), tiles, "fused train fwd")
        timeline(lambda: fused(False), tiles, "fused no_grad")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
