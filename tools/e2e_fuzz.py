"""End-to-end differential fuzz ON THE GPU BOX (not part of the suite): random synthetic metagenomes (contigs, bins, mod types,
planted motifs, methylation rates) through the product pipeline (device filters -> windows -> native lock-step search
-> native post-processing) against the CPU oracle's full pipeline (oracle/pipeline.py, one worker per bin): bin-motifs.tsv text
must be equal.   usage: python3 tools/e2e_fuzz.py [first_seed [n_seeds [procs]]]"""
import sys
import time

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch

from helpers import oracle_pipeline_parallel
from nanomotif_amd import e2e_synth, postprocess, synth
from nanomotif_amd.engine import ScanEngine

POOL = {"a": [("GATC", 1), ("CCAAAT", 4), ("ACCCA", 4), ("GAAGNNNNNNTAC", 2), ("RGATCY", 2), ("GANTC", 1), ("CAG", 1), ("TTAA", 3), ("GTAC", 2),
              ("CAMNNNNNNGTG", 1), ("GCAGC", 2), ("AAGNNNNNCTC", 1)],
        "m": [("CCWGG", 1), ("GGCC", 2), ("GCGC", 1), ("CCGG", 0), ("ACGT", 1), ("CCSGG", 1), ("GCNGC", 1), ("TCGA", 1), ("RCCGGY", 2), ("CTAG", 0)]}


def one(seed, procs):
    rng = np.random.default_rng(seed)
    mts = [("a", "m"), ("a",), ("m",), ("a", "m")][int(rng.integers(0, 4))]
    fixed = None
    if rng.random() < 0.6:
        fixed = tuple((POOL[mt][k][0], POOL[mt][k][1], mt) for mt in mts for k in rng.choice(len(POOL[mt]), size=int(rng.integers(0, 3)), replace=False))
    n_bins = int(rng.integers(1, 7))
    n_contigs, total_bp = int(rng.integers(n_bins, 6 * n_bins + 1)), int(rng.integers(150_000, 500_000)) * n_bins
    spec = synth.SynthSpec(n_contigs=n_contigs, total_bp=total_bp, n_bins=n_bins, mod_types=mts,
                           seed=int(rng.integers(0, 1 << 30)), min_contig_bp=min(int(rng.choice([2_000, 9_000, 30_000])), total_bp // (2 * n_contigs)), fixed_motifs=fixed,
                           methylated_fraction=float(rng.choice([0.97, 0.9, 0.75])))
    mg = synth.make_metagenome(spec)
    eng = ScanEngine(0)
    try:
        rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    finally:
        eng.close()
    got = postprocess.format_bin_motifs([r for r in rows if r.n_mod + r.n_nomod >= 50])
    bins = list(dict.fromkeys(mg.bin_names))
    want = oracle_pipeline_parallel(mg, bins, procs)
    assert got == want, (seed, spec, got, want)
    return f"{len(mg.names)} contigs / {n_bins} bins / {spec.total_bp} bp / {mts}: {got.count(chr(10)) - 1} motif rows, {t['rounds']} rounds"


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    bad = 0
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(seed, procs)} ({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:1500]}", flush=True)
    print("e2e fuzz done, mismatches:", bad)
    sys.exit(1 if bad else 0)
