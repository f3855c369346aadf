"""The driver parses bench.py's LAST stdout line: one JSON object, first key "metric", the contract's keys present, short enough
not to sit on the edge of its parser (round 5's 23.5 KB line beginning with "summary" came back `parsed: null`).  CPU-only: the
emitter is fed a canned measurement (the full dict of a real run, committed under profiles/)."""
import io
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CANNED = os.path.join(ROOT, "profiles", "r05_bench_bf16_line.json")
ORDER = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
         "data", "config", "roofline", "cpu_baseline"]


def _canned():
    with open(CANNED) as f:
        full = json.load(f)
    full.pop("summary_repeated", None)
    return full


def _emit(full, tmp_path):
    buf = io.StringIO()
    extras = tmp_path / "bench_extras.json"
    text = bench.emit(full, str(extras), out=buf)
    lines = buf.getvalue().splitlines()
    return text, lines, extras


def test_last_line_is_the_contract_object(tmp_path):
    full = _canned()
    text, lines, extras = _emit(full, tmp_path)
    assert lines[-1] == text and "\n" not in text
    assert len(text) <= 12_000
    obj = json.loads(text)
    assert next(iter(obj)) == "metric" and list(obj)[:len(ORDER)] == ORDER
    assert json.loads(json.dumps(obj)) == obj
    assert obj["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert obj["value"] == full["value"] and obj["ms_per_step"] == full["ms_per_step"] and obj["steps"] == full["steps"]
    assert obj["dtype"] == "bf16" and obj["data"] == "synthetic" and obj["vs_baseline"] is None and obj["higher_is_better"] is True
    assert "configs[1]" in obj["config"]["workload"] and "model" not in obj["config"]
    r = obj["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and "traffic" in r
    assert 1 <= len(r["mfma_kernels"]) <= 6 and r["symbol"] in r["mfma_kernels"]
    assert all("frac" in v for v in r["hbm_kernels"].values())
    c = obj["cpu_baseline"]
    assert c["value"] > 0 and c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["sample"] and c["unit"]
    assert list(obj)[-1] == "summary" and obj["summary"]["ms_per_step"] == full["ms_per_step"]


def test_long_tables_go_to_an_earlier_line_and_a_side_file(tmp_path):
    full = _canned()
    text, lines, extras = _emit(full, tmp_path)
    assert len(lines) == 2 and lines[0].startswith("BENCH_EXTRAS ")
    assert not lines[0].lstrip().startswith("{")                     # never mistaken for the contract line
    assert json.loads(lines[0][len("BENCH_EXTRAS "):]) == full
    assert json.load(open(extras)) == full
    assert "strong_scaling_ceiling" in full and "strong_scaling_ceiling" not in json.loads(text)


def test_an_oversized_measurement_still_fits(tmp_path):
    full = _canned()
    row = next(iter(full["roofline"]["mfma_kernels"].values()))
    for i in range(200):                                              # a pathological number of symbols / shapes / notes
        full["roofline"]["mfma_kernels"][f"sym_{i}"] = dict(row, what="x" * 500)
    for v in full["roofline"]["hbm_kernels"].values():
        if isinstance(v, dict):
            v["other_shapes"] = {f"shape {i}": {"frac": 0.1, "achieved": 1.0} for i in range(300)}
    full["roofline"]["how"] = "y" * 5000
    full["config"]["schedule"] = "z" * 5000
    text, _, _ = _emit(full, tmp_path)
    obj = json.loads(text)
    assert len(text) <= 12_000 and next(iter(obj)) == "metric"
    assert obj["roofline"]["frac"] is not None and obj["cpu_baseline"]["value"] > 0


def test_multi_gpu_line_without_single_gpu_extras(tmp_path):
    full = _canned()
    for k in ("cpu_baseline", "synthesise", "strong_scaling_ceiling", "transformer_step", "parity_mode_step"):
        full[k] = None
    full["n_gpus"], full["comm_ms_exposed"] = 8, {"generator_ms_per_step": [0.0] * 8, "discriminator_ms_per_step": [0.0] * 8}
    text, _, _ = _emit(full, tmp_path)
    obj = json.loads(text)
    assert obj["n_gpus"] == 8 and obj["cpu_baseline"] is None and len(obj["comm_ms_exposed"]["generator_ms_per_step"]) == 8
