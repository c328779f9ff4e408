#!/usr/bin/env python3
"""HBM traffic per convolution launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of one bench.py command.
    python tools/traffic_json.py <fetch.summary.txt> <write.summary.txt> <out.json> "<command>"
Counter values are KiB (rocprofv3 on gfx950); FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes (it reports half of
the bytes of wide coalesced reads)."""
import json
import re
import sys

CONV = re.compile(r"hrv::(conv_mfma_kernel|conv_wgrad|conv_patchw|spade_gb_kernel|spade_fused_kernel|conv_p2_kernel|thin_conv)")
FAMILIES = ("spade_fused_kernel", "conv_p2_kernel", "spade_gb_kernel", "conv_mfma_kernel", "conv_wgrad_tr", "conv_wgrad", "thin_conv", "norm_bwd",
            "instnorm")


def per_kernel(path, counter):
    tot, n = 0.0, 0
    for l in open(path):
        m = re.match(rf"{counter}\s+([\d.]+)\s+n=\s*(\d+)\s+(.*)", l)
        if m and CONV.search(m.group(3)):
            tot += float(m.group(1))
            n += int(m.group(2))
    return tot, n


def per_family(path, counter):
    """{family: (counter sum, dispatches)} -- a dispatch is booked to the FIRST family whose name its kernel contains"""
    out = {}
    for l in open(path):
        m = re.match(rf"{counter}\s+([\d.]+)\s+n=\s*(\d+)\s+(.*)", l)
        if not m:
            continue
        for fam in FAMILIES:
            if fam in m.group(3):
                a = out.setdefault(fam, [0.0, 0])
                a[0] += float(m.group(1))
                a[1] += int(m.group(2))
                break
    return out


def main(fetch, write, out, cmd):
    f, nf = per_kernel(fetch, "FETCH_SIZE")
    w, nw = per_kernel(write, "WRITE_SIZE")
    n = max(nf, nw, 1)
    j = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_traffic.sh) of `{cmd}`; "
                   "all hrv::conv_mfma_kernel / conv_wgrad_* / conv_patchw dispatches of the run",
         "units": "counter values are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced 16 B/lane reads)",
         "fetch_KiB_raw_total": f, "write_KiB_total": w, "conv_dispatches": n,
         "hbm_bytes_total": (2.0 * f + w) * 1024.0, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / n}
    ff, wf = per_family(fetch, "FETCH_SIZE"), per_family(write, "WRITE_SIZE")
    j["per_kernel_family"] = {fam: {"dispatches": max(ff.get(fam, [0, 0])[1], wf.get(fam, [0, 0])[1]),
                                    "fetch_KiB_raw": ff.get(fam, [0.0, 0])[0], "write_KiB": wf.get(fam, [0.0, 0])[0],
                                    "hbm_bytes_per_launch": (2.0 * ff.get(fam, [0.0, 0])[0] + wf.get(fam, [0.0, 0])[0]) * 1024.0 /
                                    max(1, max(ff.get(fam, [0, 0])[1], wf.get(fam, [0, 0])[1]))}
                             for fam in FAMILIES if fam in ff or fam in wf}
    # every kernel of the run (the "over N samples" totals of the two summaries): the whole iteration's HBM traffic
    def total(path, counter):
        for l in open(path):
            m = re.match(rf"{counter}\s+([\d.]+)\s+over\s+(\d+)\s+samples", l)
            if m:
                return float(m.group(1)), int(m.group(2))
        return None, None
    fa, na = total(fetch, "FETCH_SIZE")
    wa, _ = total(write, "WRITE_SIZE")
    if fa is not None and wa is not None:
        j["all_kernels"] = {"dispatches": na, "fetch_KiB_raw": fa, "write_KiB": wa, "hbm_bytes_whole_run": (2.0 * fa + wa) * 1024.0}
    # the profiled command runs warm-up + timed + ONE per-launch-profiled step (bench.py measure()): --steps 1 --warmup 1 -> 3
    m = re.search(r"--steps (\d+) --warmup (\d+)", cmd)
    steps = (int(m.group(1)) + int(m.group(2)) + 1) if m else None
    j["steps_in_the_profiled_run"] = steps
    if steps:
        for fam, v in j["per_kernel_family"].items():
            v["hbm_bytes_per_step"] = v["hbm_bytes_per_launch"] * v["dispatches"] / steps
        if "all_kernels" in j:      # (set-up kernels -- parameter upload, optimizer set-up -- are in the total: an upper bound)
            j["all_kernels"]["hbm_bytes_per_step"] = j["all_kernels"]["hbm_bytes_whole_run"] / steps
    with open(out, "w") as fo:
        json.dump(j, fo, indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main(*sys.argv[1:5])
