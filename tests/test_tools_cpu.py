"""Host-side helpers of the measurement tools (no GPU): interval arithmetic of the collective-overlap report and of the
patch-tile phase timeline, traffic JSON from two PMC summaries."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_overlap_interval_arithmetic():
    d = _load("dp_overlap")
    side = d.union([(0, 10), (5, 20), (30, 40)])
    assert side == [[0, 20], [30, 40]]
    assert d.inter_len(side, d.union([(8, 35)])) == 17          # 8..20 and 30..35
    assert d.inter_len(side, []) == 0


def test_committed_traffic_json_is_consistent():
    with open(os.path.join(ROOT, "profiles", "r02_pmc_traffic_train_generator.json")) as f:
        t = json.load(f)
    total = (2.0 * t["fetch_KiB_raw_total"] + t["write_KiB_total"]) * 1024.0       # FETCH_SIZE doubled (gfx950), KiB units
    assert abs(total - t["hbm_bytes_total"]) < 1e-6 * total
    assert abs(t["hbm_bytes_per_launch"] - t["hbm_bytes_total"] / t["conv_dispatches"]) < 1.0
    with open(os.path.join(ROOT, "profiles", "r02_final_bench_default.json")) as f:
        b = json.load(f)
    assert abs(b["roofline"]["traffic"] - t["hbm_bytes_per_launch"]) < 1.0      # the bench line carries this file's figure


def test_frozen_pack_stamp_is_unique_per_parameter_object():
    """The packed-weight cache of frozen networks (VGG19) must not confuse two Parameter objects whose storage address
    the caching allocator recycled (it made the fp32 gradient parity of a later test depend on an earlier one)."""
    import torch
    import hr_viton_amd  # noqa: F401
    from hr_viton_amd import train_ops as T
    a = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
    b = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
    sa, sb = T.frozen_stamp(a), T.frozen_stamp(b)
    assert sa != sb and sa == T.frozen_stamp(a)
    assert T.frozen_stamp(torch.nn.Parameter(torch.zeros(4))) is None      # trainable: never cached


def test_flat_buffer_order_puts_gradient_mates_next_to_each_other():
    """optim._flat_order: a parameter carrying ``_hrv_flat_after`` follows its mate directly (SPADE conv_beta behind
    conv_gamma, so [dW_gamma ; dW_beta] is written as one matrix); everything else keeps registration order."""
    import torch
    import torch.nn as nn
    from hr_viton_amd.optim import _flat_order
    from hr_viton_amd.network_generator import SPADENorm
    import argparse
    ps = [nn.Parameter(torch.zeros(i + 1)) for i in range(6)]
    ps[4]._hrv_flat_after = ps[1]
    ps[2]._hrv_flat_after = ps[5]          # a mate registered LATER: the pair moves to the follower's place
    got = _flat_order(ps)
    assert [p.numel() for p in got] == [1, 2, 5, 6, 3, 4]
    assert sorted(id(p) for p in got) == sorted(id(p) for p in ps)
    ps[0]._hrv_flat_after = nn.Parameter(torch.zeros(1))     # a mate outside the group is ignored
    assert [p.numel() for p in _flat_order(ps)] == [1, 2, 5, 6, 3, 4]
    opt = argparse.Namespace(norm_G="spectralaliasinstance", num_upsampling_layers="most", gen_semantic_nc=7)
    try:
        n = SPADENorm(opt, "aliasinstance", 8, 7)
    except TypeError:
        return
    names = [k for k, _ in n.named_parameters()]
    order = [names[[id(q) for q in n.parameters()].index(id(p))] for p in _flat_order(list(n.parameters()))]
    assert order.index("conv_beta.weight") == order.index("conv_gamma.weight") + 1
    assert order.index("conv_beta.bias") == order.index("conv_gamma.bias") + 1


def test_per_iteration_launch_table_is_the_difference_of_two_traces(tmp_path):
    """tools/rocprof_per_step.py: calls and time per iteration = (trace B - trace A) / (steps B - steps A); set-up kernels
    (same count in both traces) cancel, kernels that only appear in the longer trace are counted from zero."""
    import subprocess
    import sys
    a = tmp_path / "a.txt"
    b = tmp_path / "b.txt"
    a.write_text("# rocprofv3 summary\n  calls   total_ms     avg_us    pct  kernel\n"
                 "     20     10.000     500.00  50.00  void hrv::conv(hrv::P)\n"
                 "    100      1.000      10.00   5.00  setup_copy\n")
    b.write_text("# rocprofv3 summary\n  calls   total_ms     avg_us    pct  kernel\n"
                 "     60     30.000     500.00  50.00  void hrv::conv(hrv::P)\n"
                 "    100      1.000      10.00   5.00  setup_copy\n"
                 "      8      0.400      50.00   1.00  late_kernel\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_per_step.py"), str(a), "2", str(b), "6"],
                         capture_output=True, text=True, check=True).stdout
    lines = [l for l in out.splitlines() if l and not l.startswith("#")]
    assert out.splitlines()[0].startswith("# per training iteration: 12.0 kernel launches, 5.100 ms")
    assert lines[0].split()[:2] == ["10.0", "5.000"] and "hrv::conv" in lines[0]
    assert lines[1].split()[:2] == ["2.0", "0.100"] and "late_kernel" in lines[1]
    assert not any("setup_copy" in l for l in lines)


def test_committed_traffic_file_carries_the_whole_iteration_total():
    """profiles/r03_pmc_traffic_train_generator.json (tools/traffic_json.py): per-kernel-family bytes and the all-kernels total
    of the two PMC passes; the dominant kernel's bytes per iteration are what bench.py's roofline.traffic divides by its
    launch count; the families sum to less than the whole."""
    import json
    with open(os.path.join(ROOT, "profiles", "r03_pmc_traffic_train_generator.json")) as f:
        j = json.load(f)
    allk, fam = j["all_kernels"], j["per_kernel_family"]
    assert abs(allk["hbm_bytes_whole_run"] - (2.0 * allk["fetch_KiB_raw"] + allk["write_KiB"]) * 1024.0) < 1.0
    assert abs(allk["hbm_bytes_per_step"] - allk["hbm_bytes_whole_run"] / j["steps_in_the_profiled_run"]) < 1.0
    assert 150e9 < allk["hbm_bytes_per_step"] < 300e9                      # measured 244 GB (round 2: 315)
    assert sum(v["hbm_bytes_per_step"] for v in fam.values()) < allk["hbm_bytes_per_step"]
    gb = fam["spade_gb_kernel"]
    assert gb["dispatches"] % j["steps_in_the_profiled_run"] == 0
    with open(os.path.join(ROOT, "profiles", "r03_final_bench_default.json")) as f:
        r = json.load(f)["roofline"]
    assert abs(r["traffic"] - gb["hbm_bytes_per_step"] / r["launches_per_step"]) < 1e-6 * r["traffic"]
