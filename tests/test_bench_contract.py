"""bench.py's one JSON line carries what the driver's contract asks for (a small workload; the numbers are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--width", "352", "--height", "288", "--gops", "6", "--steps", "2", "--warmup", "1",
           "--cpu-frames", "2", "--cpu-cif-frames", "3", "--g-sweep", "1,2", "--host-io-steps", "2", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    j = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["unit"] == "MB/s" and j["dtype"] == "u8" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "--subme 7" in j["config"]["workload"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 6 * 396 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert r["kernel"] == "k_analyse_flow_rd" and r["avg_launch_ms"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "MB/s" and "sample" in c
    assert j["extracted_payload_BER"] == 0.0
    assert j["pcie_inclusive"]["value"] > 0 and [g["gops"] for g in j["g_sweep"]] == [1, 2, 6]


def test_strong_mode_gathers_payloads():
    j = _run("--strong", "--cpu-frames", "0", "--host-io-steps", "0", "--g-sweep", "")
    assert j["scaling"] == "strong" and j["gathered_payloads"]["gops"] == 6 and len(j["gathered_payloads"]["sha1"]) == 40
