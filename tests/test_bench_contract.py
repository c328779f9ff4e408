"""The JSON line bench.py prints (driver contract) -- checked on the committed output of the round's last default run
on the MI355X box (profiles/r04_final_bench_default.json, written by tools/round_end_measure.sh)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    for name in ("r04_final_bench_default.json",):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            with open(p) as f:
                return json.load(f)
    raise AssertionError("no committed default bench line under profiles/")


def test_default_bench_line_is_the_headline_config_with_the_contract_fields():
    j = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["unit"] == "images/s" and j["data"] == "synthetic" and j["dtype"].startswith("bf16")
    cfg = j["config"]
    # BASELINE.json's metric is quoted on configs[3] (train_generator.py 1024x768, 4 img/GPU, mixed precision)
    assert "configs[3]" in cfg["workload"] and "train_generator" in cfg["workload"] and "model" not in cfg
    assert cfg["global_batch"] == 4 and (cfg["height"], cfg["width"]) == (1024, 768)
    assert cfg["rccl_ranks"] == 0 and cfg["world_size"] == 1 and cfg["persistent_grid_cus"] == 256      # no process group: nothing ran on RCCL
    # value is whole-job throughput: global batch * steps / elapsed
    assert abs(j["value"] - cfg["global_batch"] * 1e3 / j["ms_per_step"]) < 0.01 * j["value"]


def test_roofline_object_describes_the_dominant_kernel_against_the_dense_bf16_peak():
    r = _line()["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "wasted_traffic_ratio"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and "spade_fused_kernel" in r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic flops per launch / the average HIP-event duration of exactly those launches
    avg_s = r["ms_per_step"] * 1e-3 / r["launches_per_step"]
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / avg_s / 1e12) < 0.01 * r["achieved"]
    # traffic: PMC bytes of the SAME kernel name per launch (passes of the same build, taken just before the line), and its ratio
    # to the model
    assert r["traffic"] is not None and "spade_fused_kernel" in r["traffic_source"] and "r04" in r["traffic_source"]
    assert 0.5 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.5 * r["algorithmic_bytes_per_launch"]
    assert abs(r["wasted_traffic_ratio"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 2e-3
    # the gamma|beta family (fused forwards + pair data gradients), the north-star set (every 3x3 convolution launch of the SPADE
    # generator: forward, data and weight gradients -- conv_shared counted inside the fused launches) and the whole conv family ride along
    for w in (r["spade_gamma_beta_family"], r["spade_3x3_set"], r["whole_step_conv_family"]):
        assert w["launches"] >= r["launches_per_step"] and abs(w["frac"] - w["achieved"] / 2500.0) < 1e-3
    assert r["north_star_set_frac"] == r["spade_3x3_set"]["frac"] and abs(r["north_star_set_achieved"] - r["spade_3x3_set"]["achieved"]) < 0.01
    assert r["frac"] >= 0.40 and r["spade_3x3_set"]["frac"] >= 0.27          # measured 0.418 / 0.2815
    # the HBM-bound kernel families carry bytes and a GB/s figure
    for kind in ("norm_bwd", "stats", "ew", "adam"):
        assert r["hbm_kinds"][kind]["GBps"] > 0


def test_cpu_baseline_parity_and_extra_configs():
    j = _line()
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "images/s" and "train_generator" in c["sample"]
    p = j["parity"]
    f, b = p["fp32_engine_vs_oracle"], p["bf16_engine_vs_oracle"]
    assert f["image_max_rel_err"] < 1e-3 and all(v < 1e-3 for v in f["loss_rel_err"].values())
    assert f["grad_worst_rel_err"] < 2e-2
    # (the mid-round line still shows the 5.4e-3 / 0.985 of the patch-tile packing bug that the final build fixed)
    assert "1024x768" in f["size"] and "1024x768" in b["size"]
    assert b["image_mean_abs_err"] < 3e-3 and all(v < 2e-3 for v in b["loss_rel_err"].values())
    assert b["grad_min_cosine"] > 0.99
    # the discriminator half of the same iteration
    df, db = p["discriminator_half_fp32_engine_vs_oracle"], p["discriminator_half_bf16_engine_vs_oracle"]
    assert "1024x768" in df["size"] and all(v < 1e-3 for v in df["loss_rel_err"].values()) and df["grad_worst_rel_err"] < 2e-2
    assert all(v < 5e-3 for v in db["loss_rel_err"].values())
    assert db["grad_min_cosine"] > db["bf16_rounded_oracle_vs_fp32_oracle"]["grad_min_cosine"] - 0.01
    # the CPU leg is one iteration at the metric's own resolution
    assert "1024x768" in c["sample"] and c["seconds_per_step"] > 1.0
    e = j["extra"]
    for k in ("config3_train_condition_f32_b8", "experimental_train_condition_bf16_operands_b8"):
        assert e[k]["batch"] == 8 and e[k]["value"] > 0 and "parity" in e[k]
    # configs[2] parity at its own resolution: one image 1024x768 against the oracle's autograd
    c3 = e["config3_train_condition_f32_b8"]["parity"]
    assert "1024x768" in c3["size"] and all(v < 1e-3 for v in c3["false"]["loss_rel_err"].values())
    assert c3["false"]["tocg"]["min_cosine"] > 0.9999 and c3["false"]["D"]["min_cosine"] > 0.9999
    assert "NOT a BASELINE configs[2] result" in e["experimental_train_condition_bf16_operands_b8"]["note"]
    t, q = e["config5_tryon_infer_bf16_b16"], e["config2_tocg_infer_f32_b4"]
    assert t["batch"] == 16 and abs(t["value"] - 16e3 / t["ms_per_step"]) < 0.01 * t["value"] and t["value"] > 340      # measured 371
    # configs[4] names the hipGraph-captured decode: the replay is what was timed, bit-identical to eager, and one image is held to
    # the bf16-rounded oracle
    tp = t["parity"]
    assert all(tp["hipgraph_replay_vs_eager"]["bit_identical"].values())
    o1 = tp["one_image_vs_oracle"]
    assert o1["image_mean_abs_err"] <= 2 * o1["oracle_vs_nudged_oracle_mean_abs"] + 1e-4
    assert o1["label_map_mismatch_frac"] <= 2 * o1["label_map_mismatch_frac_oracle_vs_nudged_oracle"] + 1e-3
    assert q["roofline"]["peak"] == 157.3
    qp = q["parity"]
    assert qp["seg_max_rel_err"] < 1e-3 and qp["argmax_mismatch_pixels"] <= 2e-5 * qp["pixels"]      # measured: 6 of 786 432
    # every mismatching pixel is a near-tie: the oracle's top-2 logits are within a few hundred fp32 ulps
    assert all(m <= 1024 for m in qp["mismatch_top2_margin_ulps_of_logit"])      # measured <= 286
