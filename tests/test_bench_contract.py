"""The bench line the driver parses: the committed lines under profiles/ (what bench.py printed on the
GPU box this round) must carry every key of the contract, with the right types and consistent numbers."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = ["r01_bench_line.json", "r01_bench_line_fft_c2.json", "r01_bench_line_hparma_c5.json"]


@pytest.mark.parametrize("name", LINES)
def test_committed_bench_line_has_the_contract_keys(name):
    path = os.path.join(ROOT, "profiles", name)
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert key in line and isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["vs_baseline"] is None        # BASELINE.md has no published number
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["n_gpus"] == 1
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / the kernel's measured duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    if r["traffic"] is not None:                                        # measured traffic is never below the compulsory bytes
        assert r["traffic"] >= r["algorithmic_bytes_per_launch"]
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    # whole-job throughput: frames of all steps / wall time
    assert abs(line["value"] - line["config"]["frames_per_gpu_per_step"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
