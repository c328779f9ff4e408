#!/usr/bin/env python3
"""Fill the @@NAME@@ placeholders of README.md / DESIGN.md from the round's committed final measurements (profiles/<round>_final_*):
the documents quote numbers only from files that are in the repo.     usage: python tools/fill_round_numbers.py r05 [--dry]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
    dry = "--dry" in sys.argv
    P = lambda n: os.path.join(ROOT, "profiles", f"{rnd}_{n}")      # noqa: E731
    line = json.loads(open(P("final_bench_default.json")).read())
    det = json.load(open(P("final_bench_detail.json")))
    r = det["roofline"]
    sib = {k["kernel"][5:]: k for k in r["kernels"]}
    v = {"MS": f"{line['ms_per_step']:.1f}", "IPS": f"{line['value']:.1f}", "P2FRAC": f"{r['frac']:.2f}", "P2TF": f"{r['achieved']:.0f}",
         "P2MS": f"{r['ms_per_step']:.1f}", "P2WASTE": f"{r['wasted_traffic_ratio']:.2f}" if r.get("wasted_traffic_ratio") else "n/a",
         "DOMINANT": r["kernel"][5:], "SETFRAC": f"{r['north_star_set_frac']:.3f}",
         "GFFRAC": f"{sib['spade_fused_kernel']['frac']:.2f}" if "spade_fused_kernel" in sib else "n/a"}
    e = det.get("extra") or {}
    v["SERVE"] = f"{e['config5_tryon_infer_bf16_b16']['value']:.0f}" if "config5_tryon_infer_bf16_b16" in e else "n/a"
    v["TOCG"] = f"{e['config2_tocg_infer_f32_b4']['value']:.1f}" if "config2_tocg_infer_f32_b4" in e else "n/a"
    v["COND"] = f"{e['config3_train_condition_f32_b8']['value']:.2f}" if "config3_train_condition_f32_b8" in e else "n/a"
    p = det.get("parity") or {}
    v["DCOS"] = f"{p['discriminator_half_bf16_engine_vs_oracle']['grad_min_cosine']:.3f}" if "discriminator_half_bf16_engine_vs_oracle" in p else "n/a"
    try:
        tj = json.load(open(P("final_pmc_traffic_train_generator.json")))
        v["GB"] = f"{tj['all_kernels']['hbm_bytes_per_step'] / 1e9:.1f}"
    except Exception:
        v["GB"] = "n/a"
    try:
        head = open(P("final_launches_per_iteration.txt")).readline()
        v["LAUNCHES"] = f"{float(re.search(r'([0-9.]+) kernel launches', head).group(1)):.0f}"
    except Exception:
        v["LAUNCHES"] = "n/a"
    v["P2BUSY"] = v["GFBUSY"] = "n/a"
    try:
        for l in open(P("final_mfma_busy_per_kernel.txt")):
            t = l.split()
            if len(t) >= 3 and t[0] == "hrv::conv_p2_kernel":
                v["P2BUSY"] = f"{float(t[2]):.2f}"
            if len(t) >= 3 and t[0] == "hrv::spade_fused_kernel":
                v["GFBUSY"] = f"{float(t[2]):.2f}"
    except Exception:
        pass
    try:
        m = re.search(r"(\d+) passed", open(P("final_pytest_gpu.txt")).read())
        v["NGPU"] = m.group(1) if m else "n/a"
    except Exception:
        v["NGPU"] = "n/a"
    try:      # the PatchGAN family and the per-family table of the iteration (tools/launch_summary.py of the default run's launches)
        rows = [l.rstrip() for l in open(P("final_launch_summary_train_generator_bf16.txt")) if l.strip()]
        dms = sum(float(l.split()[-6]) for l in rows if l.startswith(("conv D fwd", "conv D dgrad", "wgrad D")))
        v["DMS"] = f"{dms:.2f}"
        v["TABLE"] = "```\n" + "\n".join(rows) + "\n```"
    except Exception:
        v["DMS"] = v["TABLE"] = "n/a"
    print(json.dumps({k: (x if k != "TABLE" else "...") for k, x in v.items()}, indent=1))
    for doc in ("README.md", "DESIGN.md"):
        path = os.path.join(ROOT, doc)
        s = open(path).read()
        left = set(re.findall(r"@@([A-Z0-9]+)@@", s))
        for k in left:
            if k in v:
                s = s.replace(f"@@{k}@@", v[k])
        still = set(re.findall(r"@@([A-Z0-9]+)@@", s))
        print(doc, "filled", sorted(left - still), "unfilled", sorted(still))
        if not dry:
            open(path, "w").write(s)


if __name__ == "__main__":
    main()
