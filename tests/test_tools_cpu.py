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
