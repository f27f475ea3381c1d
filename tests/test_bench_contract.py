"""The bench line the round ends with (profiles/r03_final_bench.json = stdout of `python bench.py` on the MI355X box) carries
every field of the driver's contract, with the metric and workload BASELINE.json names."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = json.load(open(os.path.join(ROOT, "profiles", "r03_final_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "imgs/sec" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert line["dtype"] == "bf16" and line["data"] == "synthetic" and line["n_gpus"] == 1
    assert "workload" in line["config"] and "model" not in line["config"]
    assert "ResNet50" in line["config"]["workload"] and "B=512" in line["config"]["workload"]
    assert abs(line["value"] - 512 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-3
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "probe", "step_frac", "measured"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1
    # BASELINE cfg 4 / cfg 5 timed in the same process behind the headline: reported beside it, never as `value`
    ex = line["extra_configs"]
    assert [e["workload"].split(":")[0] for e in ex] == ["BASELINE cfg 4", "BASELINE cfg 5", "BASELINE cfg 5"]
    assert [e["dtype"] for e in ex] == ["bf16", "bf16", "fp8-weights"]
    for e in ex:
        assert "error" not in e and e["value"] > 0 and abs(e["value"] - int(e["workload"].split("B=")[1].split(",")[0]) * 1e3 / e["ms_per_step"]) / e["value"] < 1e-3


def test_bench_cli_defaults_match_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src
    assert 'default=1)' in src                               # --gpus defaults to one GPU
