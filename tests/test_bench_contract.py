"""bench.py's one JSON line carries what the driver's contract asks for (a small workload; the numbers are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--width", "352", "--height", "288", "--gops", "6", "--steps", "2", "--warmup", "1",
           "--cpu-frames", "2", "--cpu-cif-frames", "3", "--g-sweep", "1,2", "--host-io-steps", "3", "--clip-frames", "24", "--clip-keyints", "4", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
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
    assert j["pcie_inclusive"]["value"] > 0 and j["pcie_inclusive"]["steps"] == 3 and [g["gops"] for g in j["g_sweep"]] == [1, 2, 6]
    # what the timed loop computed, against the CPU port over the same chained frames, and GOPs of one content class against each other
    pa = j["parity_at_scale"]
    assert pa["ok"] is True and pa["gops_vs_port"] == [0, 1, 2, 3, 4, 5] and pa["chained_steps"] == 3 and pa["gops_identical_within_class"] is True
    assert j["BER_checked"]["gops"] == 6 and j["BER_checked"]["bits"] > 0
    c6 = j["clip_600"]["runs"][0]
    assert c6["keyint"] == 4 and c6["gops"] == 6 and c6["p_frames"] == 18 and c6["value"] > 0
    assert j["roofline"]["valu_issue_frac"] is None or j["roofline"]["valu_issue_frac"]["frac"] > 0


def test_content_classes_are_compared_within_class():
    """more GOPs than content classes: GOPs 0 and 4, 1 and 5 see the same pictures and must end with the same records"""
    j = _run("--classes", "4", "--cpu-frames", "0", "--host-io-steps", "0", "--g-sweep", "", "--clip-keyints", "")
    assert j["parity_at_scale"]["content_classes"] == 4 and j["parity_at_scale"]["gops_identical_within_class"] is True and j["parity_at_scale"]["ok"] is True


def test_strong_mode_gathers_payloads_and_two_ranks_agree():
    """--strong on one rank, then the N > 1 path as the driver launches it -- two ranks through torch.distributed.run (bench.py starts
    them itself, before anything touches the GPU), sharded GOPs, all_reduce of the time, all_gather + gather of the payloads -- on the
    one GPU of this box (PCAMV_BENCH_REHEARSE=1: both ranks on device 0, collectives over gloo since RCCL refuses two ranks on one
    device): the gathered payloads are byte-identical to the single-rank run's"""
    quiet = ("--strong", "--cpu-frames", "0", "--host-io-steps", "0", "--g-sweep", "", "--clip-keyints", "", "--parity-gops", "0")
    j = _run(*quiet)
    assert j["scaling"] == "strong" and j["gathered_payloads"]["gops"] == 6 and len(j["gathered_payloads"]["sha1"]) == 40
    j2 = _run("--gpus", "2", *quiet, env={"PCAMV_BENCH_REHEARSE": "1"})
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j2["config"]["gops_per_gpu"] == 3
    assert j2["gathered_payloads"]["sha1"] == j["gathered_payloads"]["sha1"] and j2["gathered_payloads"]["bytes"] == j["gathered_payloads"]["bytes"]
