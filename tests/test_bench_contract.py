"""The JSON line bench.py prints (driver contract) -- checked on the committed output of the round's last default
run (profiles/r01_final_tocg_infer_f32.json, written by tools/round_end_measure.sh on the MI355X box)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


def test_default_bench_line_has_the_contract_fields():
    j = _line("r01_final_tocg_infer_f32.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["unit"] == "images/s" and j["data"] == "synthetic" and j["dtype"] == "f32"
    assert "workload" in j["config"] and "configs[1]" in j["config"]["workload"] and "model" not in j["config"]
    # value is whole-job throughput: global batch * steps / elapsed
    assert abs(j["value"] - j["config"]["global_batch"] * 1e3 / j["ms_per_step"]) < 0.01 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic flops of the step's conv launches / their HIP-event time
    assert abs(r["achieved"] - r["flops_per_step"] / (r["conv_ms_per_step"] * 1e-3) / 1e12) < 0.05
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    p = j["parity"]
    assert p["argmax_mismatch_pixels"] <= 1e-4 * p["pixels"] and p["seg_max_rel_err"] < 1e-3


def test_secondary_workload_lines_share_the_shape():
    for name in ("tryon_infer_bf16", "train_generator_bf16", "train_condition_f32"):
        j = _line(f"r01_final_{name}.json")
        assert j["unit"] == "images/s" and j["scaling"] == "weak" and "workload" in j["config"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(j["roofline"])
        assert abs(j["value"] - j["config"]["global_batch"] * 1e3 / j["ms_per_step"]) < 0.01 * j["value"]
