"""Fuzz of binnary's read-methylation table (nm_readstats_upload + nm_contig_methylation, csrc/nmmeth.hip) ON THE GPU BOX against
oracle/contig_methylation.read_methylation: random IUPAC motifs (gaps, degenerate letters, lengths 1..14 and a few that reach
more than 31 / 63 positions from the modified base), both output types, random read filters, a contig-sharded engine.
usage: python3 tools/meth_fuzz.py [first_seed [n_seeds]]"""
import sys
import time

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np

import test_gpu_read_methylation as T
from nanomotif_amd import synth
from nanomotif_amd.contig_methylation import COLUMNS, read_methylation_table
from nanomotif_amd.engine import ScanEngine

LETTERS = "ACGTRYSWKMBDHVN"


def random_motif(rng):
    while True:
        m = _random_motif(rng)
        s, _, pos = m.rsplit("_", 2)
        if max(int(pos), len(s) - 1 - int(pos)) <= 95:        # (the engine reaches 95 positions from the modified base)
            return m


def _random_motif(rng):
    mt = "a" if rng.random() < 0.5 else "m"
    n = int(rng.choice([1, 2, 3, 4, 4, 5, 6, 6, 8, 11, 14]))
    s = list(rng.choice(list(LETTERS), size=n, p=[0.14] * 4 + [0.03] * 10 + [0.14]))
    pos = int(rng.integers(0, n))
    s[pos] = "A" if mt == "a" else "C"
    if rng.random() < 0.12:                                   # far reach: the two- and three-halo-word kernels
        gap = int(rng.choice([30, 40, 62, 70, 90]))
        if rng.random() < 0.5:
            s = s + ["N"] * gap + [str(rng.choice(list("ACGT")))]
        else:
            s = [str(rng.choice(list("ACGT")))] + ["N"] * gap + s
            pos += gap + 1
    while s[0] == "N" and pos > 0:                            # (no leading / trailing gaps: the reference's motifs are stripped)
        s, pos = s[1:], pos - 1
    while s[-1] == "N" and pos < len(s) - 1:
        s = s[:-1]
    return f"{''.join(s)}_{mt}_{pos}"


def one(seed):
    rng = np.random.default_rng(seed)
    spec = synth.SynthSpec(n_contigs=int(rng.integers(1, 9)), total_bp=int(rng.integers(60_000, 500_000)), n_bins=1, mod_types=("a", "m"),
                           seed=int(rng.integers(0, 1 << 30)), min_contig_bp=5_000, fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    rec = T._records(mg, rng)
    motifs = sorted({random_motif(rng) for _ in range(int(rng.integers(4, 40)))})
    min_cov, min_frac = int(rng.choice([1, 3, 3, 8])), float(rng.choice([0.8, 0.8, 0.5, 0.95]))
    eng = ScanEngine(0)
    try:
        n = len(mg.names)
        mine = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist()) if rng.random() < 0.4 else list(range(n))
        eng.upload_assembly([mg.names[i] for i in mine], [mg.contig_ascii(i) for i in mine], [mg.bin_names[i] for i in mine])
        T._upload(eng, mg, rec, min_cov, min_frac, shard=None if len(mine) == n else {g: k for k, g in enumerate(mine)})
        total = 0
        for output_type in ("median", "weighted-mean"):
            got = [tuple(r[c] for c in COLUMNS) for r in read_methylation_table(eng, motifs, output_type)]
            want = T._oracle_rows(mg, rec, motifs, output_type, min_cov, min_frac, contigs=None if len(mine) == n else mine)
            assert len(got) == len(want), (seed, output_type, len(got), len(want))
            for g, w in zip(got, want):
                assert g == w, (seed, output_type, g, w)
            total += len(got)
    finally:
        eng.close()
    return f"{len(motifs)} motifs x {len(mine)} of {len(mg.names)} contigs ({spec.total_bp} bp), reads >= {min_cov} / {min_frac}: {total} rows"


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = 0
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(seed)} ({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:1200]}", flush=True)
    print("read-methylation fuzz done, mismatches:", bad)
    sys.exit(1 if bad else 0)
