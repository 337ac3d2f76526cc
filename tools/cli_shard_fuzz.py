"""Sharded CLI runs on the GPU box (not part of the suite): random synthetic metagenomes through `motif_discovery` on ONE
rank and on 2 / 3 ranks sharing the one GPU of the box (gloo carries the exchanges, NANOMOTIF_DIST_BACKEND=gloo), `--shard contigs`
(count tables all-reduced every lock-step round) and `--shard bins` (whole bins per rank, rows gathered at the end):
bin-motifs.tsv must be byte-equal to the one-rank run.   usage: python3 tools/cli_shard_fuzz.py [first_seed [n_seeds]]"""
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from nanomotif_amd import synth

POOL = {"a": [("GATC", 1), ("CCAAAT", 4), ("ACCCA", 4), ("GAAGNNNNNNTAC", 2), ("RGATCY", 2), ("GANTC", 1), ("CAG", 1), ("GTAC", 2)],
        "m": [("CCWGG", 1), ("GGCC", 2), ("GCGC", 1), ("CCGG", 0), ("ACGT", 1), ("GCNGC", 1), ("TCGA", 1)]}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(tmp, out, nproc, shard, env):
    args = ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", out]
    if nproc == 1:
        cmd = [sys.executable, "-m", "nanomotif_amd", "motif_discovery"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
               "-m", "nanomotif_amd", "motif_discovery", "--device", "0", "--shard", shard] + args
    r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (nproc, shard, r.stdout[-1200:], r.stderr[-1200:])
    return open(f"{tmp}/{out}/bin-motifs.tsv").read()


def one(seed):
    rng = np.random.default_rng(3000 + seed)
    mts = [("a", "m"), ("a",), ("m",), ("a", "m")][int(rng.integers(0, 4))]
    fixed = tuple((POOL[mt][k][0], POOL[mt][k][1], mt) for mt in mts for k in rng.choice(len(POOL[mt]), size=int(rng.integers(1, 3)), replace=False))
    n_bins = int(rng.integers(1, 6))
    n_contigs, total_bp = int(rng.integers(n_bins, 5 * n_bins + 1)), int(rng.integers(150_000, 400_000)) * n_bins
    spec = synth.SynthSpec(n_contigs=n_contigs, total_bp=total_bp, n_bins=n_bins, mod_types=mts, seed=int(rng.integers(0, 1 << 30)),
                           min_contig_bp=min(int(rng.choice([2_000, 9_000, 30_000])), total_bp // (2 * n_contigs)), fixed_motifs=fixed)
    mg = synth.make_metagenome(spec)
    tmp = tempfile.mkdtemp(prefix="nm_shardfuzz_")
    try:
        mg.write_fasta(tmp + "/assembly.fasta"); mg.write_contig_bin(tmp + "/contig_bin.tsv"); mg.write_bed(tmp + "/pileup.bed")
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), NANOMOTIF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        single = run(tmp, "out1", 1, None, env)
        nproc = int(rng.choice([2, 2, 3]))
        shard = ["contigs", "bins", "auto"][int(rng.integers(0, 3))]
        if rng.random() < 0.3:
            env["NANOMOTIF_INGEST_PART_ROWS"] = "100000"
        many = run(tmp, "outN", nproc, shard, env)
        assert many == single, (seed, nproc, shard, spec, many, single)
        return f"{nproc} ranks --shard {shard}: {n_contigs} contigs / {n_bins} bins / {mts}: {single.count(chr(10)) - 1} motif rows, equal"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    bad = 0
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(seed)} ({time.time() - t0:.1f} s)", flush=True)
        except (AssertionError, subprocess.TimeoutExpired) as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:2500]}", flush=True)
    print("sharded cli fuzz done, mismatches:", bad)
    sys.exit(1 if bad else 0)
