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
    assert rf["bound"] in ("hbm", "mfma", "valu-int") and rf["unit"] == "GB/s" and abs(rf["hbm_frac"] - rf["frac"]) < 1e-12 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "traffic" in rf and rf["kernel_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["value_t1"] > 0 and cb["host_cores_visible"] >= cb["host_cores_usable"] >= 1
    assert d["parity"]["mismatches"] == 0 and d["parity"]["candidates_checked"] > 0
    # extras of the default run: end to end on the same metagenome, all-bins table, the HBM-bound round
    e = d["e2e"]
    assert e["wall_s"] > 0 and e["search_s"] > 0 and e["gpu_busy_s"] >= 0 and e["rounds"] > 0 and e["wall_s"] >= e["search_s"]
    assert d["cfg5_all"]["agrees_with_per_bin_table"] is True and d["cfg5_all"]["value"] > 0
    assert d["roofline_hbm_bound_round"]["frac"] > 0 and "traffic_frac_of_streaming" in d["roofline_hbm_bound_round"]
    assert d["per_rank"][0]["kernel_ms"] > 0 and d["per_rank"][0]["host_call_ms_per_step"] > 0
    # one GPU: the strict-order pass is the headline, the two-lane pass of the same steps is reported beside it
    assert d["pipelining"]["scoring_lanes"] == 1 and d["two_lanes"]["scoring_lanes"] == 2 and d["two_lanes"]["value"] > 0
    assert d["two_lanes"]["steps"] == 3 and d["prewarm_steps"] == 200


def _bench(extra, nproc=1, launcher=True):
    small = ["--steps", "3", "--warmup", "1", "--total-bp", "20000000", "--contigs", "200", "--bins", "10", "--candidates", "200",
             "--cpu-bins", "0", "--hbm-round-steps", "0"]
    if nproc == 1:
        cmd = [sys.executable, "bench.py", "--extras", "none"]
    elif launcher:      # the way the driver starts N ranks
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "bench.py", "--gpus", str(nproc), "--dist-backend", "gloo", "--force-device", "0"]
    else:               # bench.py starts its own ranks
        cmd = [sys.executable, "bench.py", "--gpus", str(nproc), "--dist-backend", "gloo", "--force-device", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd + small + extra, cwd=ROOT, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_two_rank_bench_strong_default_and_weak():
    """N = 2 on the one GPU of the test box (gloo).  The default is the BASELINE configuration: ONE metagenome, contigs
    sharded, one all-reduce of the count table per step — the all-reduced table equals the single-rank table; the run
    also carries the end-to-end and weak-scaling extras.  ``--scaling weak`` doubles the work and says so."""
    one = _bench([])
    strong = _bench([], nproc=2)
    weak = _bench(["--scaling", "weak"], nproc=2)
    assert strong["n_gpus"] == weak["n_gpus"] == 2
    assert strong["scaling"] == "strong" and weak["scaling"] == "weak"
    assert "sharded over 2 GPUs" in strong["config"]["workload"] and "20,000,000 bp total" in strong["config"]["workload"]
    assert strong["counts_checksum"] == one["counts_checksum"]                     # all-reduced table == unsharded table
    assert weak["counts_checksum"] == one["counts_checksum"]                       # rank 0 holds the seed-1 metagenome
    assert weak["config"]["motif_sites_per_step"] == 2 * one["config"]["motif_sites_per_step"]
    assert strong["config"]["motif_sites_per_step"] == one["config"]["motif_sites_per_step"]
    assert len(strong["per_rank"]) == 2 and strong["allreduce_ms"] > 0
    assert sum(p["contigs"] for p in strong["per_rank"]) == 200
    assert strong["weak_scaling"]["value"] > 0 and strong["e2e"]["motif_rows"] > 0 and strong["e2e"]["wall_s"] > 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher spawns the two ranks itself and prints one n_gpus = 2 line."""
    d = _bench(["--extras", "none"], nproc=2, launcher=False)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and len(d["per_rank"]) == 2
    one = _bench([])
    assert d["counts_checksum"] == one["counts_checksum"]


def test_single_rank_step_with_the_c_abi_allreduce():
    """--force-allreduce: the strong-scaling step (score into table i, nm_allreduce_counts_async on the communication
    stream, nm_comm_wait before table i is reused) with an RCCL world of one rank — same table, all-reduce timed."""
    one = _bench([])
    d = _bench(["--force-allreduce"])
    assert d["counts_checksum"] == one["counts_checksum"] and d["allreduce_ms"] > 0
    assert "nm_allreduce_counts" in d["config"]["sharding"] or d["n_gpus"] == 1
    # the same step on two scoring lanes (what several GPUs run as their headline): same table, strict pass beside it
    two = _bench(["--force-allreduce", "--lanes", "2"])
    assert two["counts_checksum"] == one["counts_checksum"] and two["pipelining"]["scoring_lanes"] == 2
    assert two["strict_order"]["scoring_lanes"] == 1 and two["strict_order"]["value"] > 0 and two["roofline"]["kernel_ms"] > 0
    only = _bench(["--lanes", "1"])
    assert "two_lanes" not in only and "strict_order" not in only and only["counts_checksum"] == one["counts_checksum"]


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], cwd=ROOT, capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
