#!/usr/bin/env python3
"""A/B of the dedicated SPADE gamma|beta kernel (csrc/spade_gb.hip, HRV_SPADE_GB=1) against the generic patch tiles
(HRV_SPADE_GB=0) on the generator's norm shapes: training forward (SPADE epilogue, (1 + gamma) saved) and data gradient
(ReLU mask), interleaved rounds in ONE process, median; plus the per-tile phase timeline of the new kernel
(hrv_diag_set_tlog).      python tools/gb_bench.py [rounds]      (via gpurun)"""
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import hr_viton_amd  # noqa: E402,F401
from hr_viton_amd import ops, train_ops as T  # noqa: E402
from hr_viton_amd.gen_train import SpadeT  # noqa: E402
from hr_viton_amd.network_generator import SPADENorm  # noqa: E402

SHAPES = [("up_4.norm_0", 80, 4, 1024, 768), ("up_4.norm_1", 32, 4, 1024, 768), ("up_3.norm_0", 144, 4, 512, 384), ("up_3.norm_1", 64, 4, 512, 384),
          ("up_2.norm_0", 272, 4, 256, 192)]


def timeline(run, tiles, label):
    tlog = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    from hr_viton_amd import _lib
    _lib.check(_lib.load().hrv_diag_set_tlog(tlog.data_ptr(), tiles), "hrv_diag_set_tlog")
    run()
    torch.cuda.synchronize()
    _lib.check(_lib.load().hrv_diag_set_tlog(None, 0), "hrv_diag_set_tlog")
    t = tlog.cpu().view(tiles, 8)
    t = t[t[:, 3] > 0]
    t0 = int(t[:, 0].min())
    us = lambda v: (v - t0).double() / 100.0          # noqa: E731   wall_clock64: 100 MHz
    st, lp, ep, en = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
    print(f"   timeline {label}: {t.shape[0]}/{tiles} tiles, device span {float(en.max()):.0f} us")
    iss = us(t[:, 6])
    for nm, d in (("prologue", lp - st), ("main loop(s)", ep - lp), ("epilogue", en - ep), ("  stores issued", iss - ep),
                  ("  drain (vmcnt 0)", en - iss), ("tile", en - st)):
        print(f"      {nm:14s} mean {float(d.mean()):7.2f} us  median {float(d.median()):7.2f}  p90 {float(d.quantile(0.9)):7.2f}")


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    T.MMA_BF16[0] = True
    torch.manual_seed(0)
    for name, Cc, N, H, W in SHAPES:
        norm = SPADENorm(Namespace(), "aliasinstance", Cc, 7).cuda()
        st = SpadeT(norm, ops.ACT_LRELU, name)
        x = ops.Act(torch.randn(N, H, W, Cc, device="cuda"), Cc)
        actv_all = torch.relu(torch.randn(N, H, W, 384, device="cuda")).to(torch.bfloat16)
        actv = ops.Act(actv_all, 128, 128)
        z = torch.randn(N, W, H, 1, device="cuda")
        dgb = ops.Act(torch.randn(N, H, W, 2 * Cc, device="cuda").to(torch.bfloat16), 2 * Cc)
        dact_all = torch.empty(N, H, W, 384, device="cuda", dtype=torch.bfloat16)      # bf16 d(actv), as in the mixed-precision plan
        dact = ops.Act(dact_all, 128, 128)
        wpair = (norm.conv_gamma.weight.data, norm.conv_beta.weight.data)

        def fwd():
            st.forward(x, actv, z, save=True)

        def dgrad():
            if os.environ.get("HRV_SPADE_GB", "1") != "0" and T.spade_gb_ok(1, Cc, Cc, 128, N, H, W):
                T.spade_gb_dgrad(dgb, T.spade_gb_pack(1, *wpair), Cc, actv, 0.0, dact, name + ".gb.dgrad")
            else:
                T.conv_dgrad(dgb, wpair, H, W, 1, 1, act_mask=actv, slope=0.0, out=dact, name=name + ".gb.dgrad")
        fl = 2.0 * N * H * W * 2 * Cc * 128 * 9
        print(f"{name}: C={Cc} N={N} {H}x{W}  ({fl / 1e12:.3f} TFLOP per launch)")
        for what, fn in (("forward", fwd), ("dgrad", dgrad)):
            times = {"1": [], "0": [], "ns": []}
            for rd in range(rounds + 2):
                for flag in ("1", "0", "ns"):
                    os.environ["HRV_SPADE_GB"] = "0" if flag == "0" else "1"
                    if flag == "ns":
                        os.environ["HRV_GB_STAGGER"] = "0"
                    else:
                        os.environ.pop("HRV_GB_STAGGER", None)
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    fn()
                    e.record()
                    torch.cuda.synchronize()
                    if rd >= 2:
                        times[flag].append(s.elapsed_time(e))
            os.environ.pop("HRV_GB_STAGGER", None)
            for flag, lab in (("1", "spade_gb kernel"), ("ns", "spade_gb, no stagger"), ("0", "generic patch tiles")):
                ts = sorted(times[flag])
                med = ts[len(ts) // 2]
                print(f"   {what:8s} {lab:20s} median {med:7.3f} ms  min {ts[0]:7.3f}  {fl / (med * 1e-3) / 1e12:7.1f} TFLOP/s "
                      "(incl. the weight pack launch)")
        os.environ["HRV_SPADE_GB"] = "1"
        tiles = N * ((H + 15) // 16) * ((W + 15) // 16)
        timeline(fwd, tiles, "forward")
        timeline(dgrad, tiles, "dgrad")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
