"""bench.py prints ONE JSON line with the fields the driver reads (small configuration, a few seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--total-bp", "20000000", "--contigs", "200",
                        "--bins", "10", "--candidates", "200", "--cpu-bins", "2", "--cpu-procs", "2", "--hbm-round-steps", "2"],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and d["data"] == "synthetic"
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "traffic" in rf and rf["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert d["parity"]["mismatches"] == 0 and d["parity"]["candidates_checked"] > 0


def _bench(extra, nproc=1):
    small = ["--steps", "3", "--warmup", "1", "--total-bp", "20000000", "--contigs", "200", "--bins", "10", "--candidates", "200",
             "--cpu-bins", "0", "--hbm-round-steps", "0"]
    cmd = [sys.executable, "bench.py"] if nproc == 1 else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), "bench.py", "--gpus", str(nproc), "--dist-backend", "gloo", "--force-device", "0"]
    r = subprocess.run(cmd + small + extra, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.strip().split("\n") if l.startswith("{")][-1])


def test_two_rank_bench_weak_and_strong():
    """N = 2 on the one GPU of the test box (gloo): strong scaling reproduces the single-rank count table, weak scaling
    doubles the work and reports it."""
    one = _bench([])
    strong = _bench(["--scaling", "strong"], nproc=2)
    weak = _bench(["--scaling", "weak"], nproc=2)
    assert strong["n_gpus"] == weak["n_gpus"] == 2
    assert strong["scaling"] == "strong" and weak["scaling"] == "weak"
    assert strong["counts_checksum"] == one["counts_checksum"]                     # all-reduced table == unsharded table
    assert weak["counts_checksum"] == one["counts_checksum"]                       # rank 0 holds the seed-1 metagenome
    assert weak["config"]["motif_sites_per_step"] == 2 * one["config"]["motif_sites_per_step"]
    assert strong["config"]["motif_sites_per_step"] == one["config"]["motif_sites_per_step"]
