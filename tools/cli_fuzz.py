"""Differential fuzz of the CLI ON FILES on the GPU box (not part of the suite): random synthetic metagenomes written as
FASTA + bedMethyl — plain text, or bgzip + tabix with random block sizes / compression levels / strategies, the contigs of the
pileup in a shuffled order, some contigs of the assembly in no bin — through `python -m nanomotif_amd motif_discovery`
(device text parser, device inflate, tabix subset, device filters, search, post-processing) against the CPU oracle's
pipeline: bin-motifs.tsv text for text.   usage: python3 tools/cli_fuzz.py [first_seed [n_seeds]]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from helpers import oracle_pipeline_parallel, write_bgzf_tabix
from nanomotif_amd import synth

POOL = {"a": [("GATC", 1), ("CCAAAT", 4), ("ACCCA", 4), ("GAAGNNNNNNTAC", 2), ("RGATCY", 2), ("GANTC", 1), ("CAG", 1), ("GTAC", 2), ("GCAGC", 2)],
        "m": [("CCWGG", 1), ("GGCC", 2), ("GCGC", 1), ("CCGG", 0), ("ACGT", 1), ("GCNGC", 1), ("TCGA", 1), ("RCCGGY", 2)]}


def one(seed):
    rng = np.random.default_rng(1000 + seed)
    mts = [("a", "m"), ("a",), ("m",), ("a", "m")][int(rng.integers(0, 4))]
    fixed = tuple((POOL[mt][k][0], POOL[mt][k][1], mt) for mt in mts for k in rng.choice(len(POOL[mt]), size=int(rng.integers(1, 3)), replace=False))
    n_bins = int(rng.integers(1, 5))
    n_contigs, total_bp = int(rng.integers(n_bins, 5 * n_bins + 1)), int(rng.integers(150_000, 400_000)) * n_bins
    spec = synth.SynthSpec(n_contigs=n_contigs, total_bp=total_bp, n_bins=n_bins, mod_types=mts, seed=int(rng.integers(0, 1 << 30)),
                           min_contig_bp=min(int(rng.choice([2_000, 9_000, 30_000])), total_bp // (2 * n_contigs)), fixed_motifs=fixed)
    mg = synth.make_metagenome(spec)
    tmp = tempfile.mkdtemp(prefix="nm_clifuzz_")
    try:
        mg.write_fasta(tmp + "/assembly.fasta")
        mg.write_contig_bin(tmp + "/contig_bin.tsv")
        mg.write_bed(tmp + "/pileup.bed")
        kind = ["plain", "plain-shuffled", "bgzip", "bgzip"][int(rng.integers(0, 4))]
        text = open(tmp + "/pileup.bed", "rb").read()
        if kind != "plain":
            # the contigs of the pileup in another order than the assembly's (modkit sorts by its own reference order)
            runs, cur, at = {}, None, 0
            for line in text.splitlines(True):
                name = line[:line.index(b"\t")]
                runs.setdefault(name, []).append(line)
            order = list(runs)
            rng.shuffle(order)
            text = b"".join(b"".join(runs[n]) for n in order)
            open(tmp + "/pileup.bed", "wb").write(text)
        bed, how = "pileup.bed", kind
        bgzip = kind == "bgzip"
        if bgzip:
            bs = int(rng.choice([0xFF00, 0xFF00, 20_000, 3_000, 700]))
            level = int(rng.choice([6, 6, 1, 9, 0]))
            strategy = int(rng.choice([0, 0, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY]))
            write_bgzf_tabix(text, tmp + "/pileup.bed.gz", block_size=bs, level=level, strategy=strategy)
            bed, how = "pileup.bed.gz", f"bgzip(block {bs}, level {level}, strategy {strategy})"
        threads = int(rng.choice([1, 1, 4]))
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        if bgzip and rng.random() < 0.3:
            env["NM_BED_INFLATE_SLAB"] = str(int(rng.choice([60_000, 300_000])))
        # other routes through the same pipeline: the pileup ingested in several parts, the host parser, the Python twins of
        # the native search / post-processing, windows planned per task
        extra = []
        if rng.random() < 0.35:
            env["NANOMOTIF_INGEST_PART_ROWS"] = str(int(rng.choice([20_000, 100_000, 400_000]))); extra.append("parts " + env["NANOMOTIF_INGEST_PART_ROWS"])
        if rng.random() < 0.2:
            env["NANOMOTIF_HOST_PARSER"] = "1"; extra.append("host parser")
        if rng.random() < 0.15:
            env["NANOMOTIF_PY_POST"] = "1"; extra.append("python post")
        if rng.random() < 0.1:
            env["NANOMOTIF_PY_SEARCH"] = "1"; extra.append("python search")
        if rng.random() < 0.1:
            env["NANOMOTIF_PLAN_PER_TASK"] = "1"; extra.append("plan per task")
        how += "".join(", " + e for e in extra)
        r = subprocess.run([sys.executable, "-m", "nanomotif_amd", "motif_discovery", "assembly.fasta", bed, "-c", "contig_bin.tsv", "--out", "out", "-t", str(threads)],
                           cwd=tmp, env=env, capture_output=True, text=True)
        assert r.returncode == 0, (seed, how, r.stdout[-1500:], r.stderr[-1500:])
        got = open(tmp + "/out/bin-motifs.tsv").read()
        bins = list(dict.fromkeys(mg.bin_names))
        want = oracle_pipeline_parallel(mg, bins, 12, bgzip_order=bgzip)
        assert got == want, (seed, how, spec, got, want)
        return f"{how}, -t {threads}: {len(mg.names)} contigs / {n_bins} bins / {mts}: {got.count(chr(10)) - 1} motif rows"
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
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:2500]}", flush=True)
    print("cli fuzz done, mismatches:", bad)
    sys.exit(1 if bad else 0)
