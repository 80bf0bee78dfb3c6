"""The bench line the driver parses: the committed lines under profiles/ (what bench.py printed on the
GPU box this round) must carry every key of the contract, with the right types and consistent numbers."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith(("r01_bench_line", "r02_bench_line")))


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


def _run_bench(*argv, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no torch.distributed.run around it): the parent starts
    one child per rank before touching any GPU, rank 0's line comes back on stdout, exit code 0.
    --dry-run keeps the launch path (rendezvous on 127.0.0.1, barrier, max over ranks) and skips
    the GPU work, so this runs on the CPU with gloo."""
    r = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line, nothing else on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["max_over_ranks"] == 2.0 and line["steps"] == 3 and line["warmup"] == 1


def test_bench_gpus_8_dry_run():
    """The launch the driver makes for the scaling curve's last point, dry: eight ranks rendezvous on 127.0.0.1, the line carries
    every rank's own value beside the maximum, and the host-side phase runs on rank 0 alone while the other seven wait at a
    CPU barrier (bench.py host_side_phase) -- the shape of the N > 1 line's host rows."""
    r = _run_bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["max_over_ranks"] == 8.0
    assert line["per_rank"] == [float(i + 1) for i in range(8)]
    assert line["host_phase"] == {"slept_s": 0.2}


def test_bench_under_a_launcher_is_one_rank():
    """Under torch.distributed.run the environment carries WORLD_SIZE: no self-launch, and a
    --gpus that disagrees with it is refused instead of asserting half-way."""
    r = _run_bench("--gpus", "1", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 1
    r = _run_bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_parent_gives_up_when_a_rank_dies():
    """A rank that exits before the rendezvous must not leave the others (and the driver) waiting:
    the parent polls its children, ends the rest and exits with the failing rank's code."""
    import time
    t0 = time.time()
    r = _run_bench("--gpus", "2", "--dry-run", env={"GLFER_BENCH_FAIL_RANK": "1"})
    assert r.returncode == 3 and time.time() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
