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
