"""bench.py end to end at a small particle count (GPU): the one JSON line the driver parses carries every field of
the contract, the step counts it was asked for, a roofline measured live and a CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--device-warmup-ms", "0"], ["--workload", "C5"]], ids=["default", "no_device_warmup", "C5"])
def test_bench_prints_the_contract_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--particles", "2e5",
           "--cpu-sample", "20000", "--cpu-steps", "2", "--device-warmup-ms", "30", *extra]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"] == "particle-steps/s" and d["unit"] == "particle-steps/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert abs(d["value"] - 2e5 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"] and d["value"] > 1e7   # (a floor only: the suite may share the GPU between workers)
    cfg = d["config"]
    assert isinstance(cfg["workload"], str) and cfg["particles_per_gpu"] == 200000 and "model" not in cfg
    assert ("none" in cfg["device_warmup"]) == ("--device-warmup-ms" in extra and extra[extra.index("--device-warmup-ms") + 1] == "0")
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # (2e5 particles touch ~100 MB of meteo records: inside the Infinity Cache, no share of the HBM roof is claimed)
    assert r["kernel_ms"] > 0 and r["frac"] is None and "Infinity Cache" in r["frac_note"] and r["frac_one_launch_per_step"] is None
    assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-9 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02 and "traffic" in r
    # one rank: no communicator; the per-rank kernel times the driver reads are there
    assert cfg["rccl_ranks"] == 0 and len(r["kernel_ms_per_rank"]) == 1
    if "C5" in extra:   # a step of several unlike kernels is priced as a whole (wall time per step)
        assert r["step_kernel_launches_per_step"] == 2 and abs(r["kernel_ms"] - d["ms_per_step"]) <= 1e-9 * r["kernel_ms"]
        assert r["step_kernel_ms_per_step"] < r["kernel_ms"]
    else:
        # (2e5 particles: the K steps go to the library as one call and share launches)
        assert r["step_kernel_launches_per_step"] <= 1 and r["kernel_ms"] == r["step_kernel_ms_per_step"]
        assert "mphip_run_timesteps" in cfg["time_loop"]
    # both bounds stated: the HBM pricing under both launch regimes, the modelled VALU-issue roof (None at
    # an overridden particle count: the committed instruction mix belongs to the workload's own), what pins the oracle
    assert "alu" in r and "algorithmic" in r["basis"]
    assert any("RK4" in m for m in d["parity"]["restatement_only"]) and d["parity"]["pinned_by_reference_goldens"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "particle-steps/s" and c["cores"] >= 1 and c["value"] > 1e4
    assert "sample" in c and c["value_1_thread"] > 0
