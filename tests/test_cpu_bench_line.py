"""bench.py's driver-facing line must stay small enough for the driver to parse (round 4's 25.7 KB line was not:
BENCH_r04.json parsed = null) and must carry the `roofline` and `cpu_baseline` blocks; the full result goes to
bench_detail.json."""
import json
import os

import bench

HERE = os.path.dirname(os.path.abspath(__file__))


def _canned():
    with open(os.path.join(HERE, "fixtures", "bench_full_result_r04.json")) as f:
        return json.load(f)


def test_compact_line_is_small_and_carries_roofline_and_cpu_baseline():
    full = _canned()
    assert len(json.dumps(full)) > 20000            # the canned result is the line that broke the driver
    line = bench.compact_line(full)
    txt = json.dumps(line)
    assert len(txt) < bench.COMPACT_LIMIT < 6000
    assert "\n" not in txt
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_e2e", "sustained", "library",
              "parity", "detail"):
        assert k in line, k
    assert line["value"] == round(full["value"], 1) or abs(line["value"] - full["value"]) < 1e-3 * full["value"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "arith"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]
    assert "bf16" in line["dtype"]
    # SURVEY 8(d)'s metric is first-class, with its three components, and says which value includes the H2D
    assert line["value_includes_h2d"] is False
    assert {"prepare_data_ms", "train_ms", "weights_ms", "ms_per_update", "host_path"} <= set(line["e2e"])
    assert set(line["secondary"]) == {"breakout_impala", "pong_impala_speedup"}


def test_compact_line_drops_optional_blocks_before_it_outgrows_the_limit(monkeypatch):
    full = _canned()
    full["config"]["workload"] = full["config"]["workload"] + " x" * 1200       # long strings are cut ...
    assert len(json.dumps(bench.compact_line(full))) < bench.COMPACT_LIMIT
    monkeypatch.setattr(bench, "COMPACT_LIMIT", 1900)                           # ... and optional blocks go first
    line = bench.compact_line(full)
    assert len(json.dumps(line)) <= bench.COMPACT_LIMIT
    assert line.get("truncated") is True
    assert "roofline" in line and "cpu_baseline" in line and "value" in line


def test_compact_line_of_a_multi_gpu_result():
    full = _canned()
    full["n_gpus"] = 8
    full.pop("roofline"); full.pop("cpu_baseline"); full.pop("e2e")
    full["dp_variants"] = {"eager": {"valid": True, "value": 1.0e7, "ms_per_step": 13.0},
                           "direct": {"valid": True, "value": 1.5e7, "ms_per_step": 9.0},
                           "hook": {"valid": False, "error": "x" * 500}, "note": "text"}
    full["strict"] = {"value": 2.5e6, "ms_per_step": 6.5, "rows_per_gpu": 40, "scaling": "strong", "dp_variants": {}}
    full["secondary"] = [{"workload": "examples/pong_impala_speedup.yaml ...", "weak": {"value": 1e8}, "strict": {"value": 2e7}}]
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.COMPACT_LIMIT
    assert line["dp_variants"]["direct"] == 1.5e7 and line["dp_variants"]["hook"].startswith("invalid")
    assert line["strict"]["rows_per_gpu"] == 40
    assert line["secondary"]["pong_impala_speedup"] == {"weak": 1e8, "strict": 2e7}
