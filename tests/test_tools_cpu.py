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
