"""The scan -> count step against the REFERENCE'S OWN motif_model_contig / motif_model_bin (build container only, like
gen_golden.py's g2): random contigs (N runs, stray IUPAC letters), random pileups (rows on any base and strand, fractions on and
around the thresholds), random motifs (literals, 2- and 3-base sets, gaps, offsets out to +-63, the modified position on a set or
on '.'), random threshold pairs — reference against oracle/scan.py: counts and all four hit-position arrays.  (The HIP engine is
fuzzed against oracle/scan.py on the GPU box: tests/test_gpu_fuzz.py, tools/fuzz_more.py.)
usage: python3 tests/golden/scan_ref_fuzz.py [first_seed [n_seeds]]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np

import refstub
from oracle import scan as osc
from oracle.model import BetaBernoulliModel as OModel
from oracle.motif import Motif as OMotif
from test_gpu_fuzz import _contig, _motif


def one(nm, seed):
    from nanomotif.model import BetaBernoulliModel
    from nanomotif.motif import Motif
    from nanomotif.seq import DNAsequence
    fmb = nm.find_motifs_bin
    rng = np.random.default_rng(7_000 + seed)
    lens = [int(x) for x in rng.choice([1, 2, 31, 33, 100, 1000, 8191, 8193, 12_000, 30_000], size=int(rng.integers(1, 4)))]
    names = [f"c{i}" for i in range(len(lens))]
    seqs = [_contig(rng, n).upper() for n in lens]
    fr_values = np.array([0.0, 0.1, 0.3, 0.30000000000000004, 0.29999999999999993, 0.5, 0.7, 0.7000000000000001, 0.6999999999999999, 0.95, 1.0])
    low, high = [(0.3, 0.7), (0.3, 0.7), (0.1, 0.9), (0.5, 0.5)][int(rng.integers(0, 4))]
    cols = {"contig": [], "position": [], "strand": [], "frac": []}
    opile = {}
    for i, n in enumerate(lens):
        k = int(rng.integers(0, 2 * n + 1))
        flat = np.sort(rng.choice(2 * n, size=k, replace=False))                       # unique (position, strand), in position order
        p, strand = (flat // 2).astype(np.int64), np.where(flat % 2 == 0, ord("+"), ord("-")).astype(np.uint8)
        f = rng.choice(fr_values, size=len(p))
        opile[names[i]] = osc.ContigPileup(p, strand, f)
        cols["contig"] += [names[i]] * len(p); cols["position"].append(p); cols["strand"] += [chr(c) for c in strand.tolist()]; cols["frac"].append(f)
    pile = refstub.make_pileup(cols["contig"], np.concatenate(cols["position"]) if cols["position"] else np.zeros(0, np.int64), cols["strand"],
                               np.concatenate(cols["frac"]) if cols["frac"] else np.zeros(0))
    pl = sys.modules["polars"]
    n_checked = 0
    for k in range(40):
        s, p = _motif(rng, wide=(k % 5 == 0))
        if k % 10 == 7:                                              # reaches of 64..95 positions (the engine's widest kernels)
            gap = int(rng.integers(60, 93))
            s, p = ("ACGT"[int(rng.integers(4))] + "." * gap + s, p + gap + 1) if rng.random() < 0.5 else (s + "." * gap + "ACGT"[int(rng.integers(4))], p)
        # per contig, with positions
        for i, name in enumerate(names):
            sub = pile.filter(pl.col("contig") == name)
            rmodel, rd = fmb.motif_model_contig(sub, seqs[i], BetaBernoulliModel(), Motif(s, p), low_meth_threshold=low, high_meth_threshold=high,
                                                save_motif_positions=True)
            omodel, od = osc.motif_model_contig(opile[name], seqs[i], OModel(), OMotif(s, p), low, high, save_motif_positions=True)
            assert tuple(int(x) for x in rmodel.get_raw_counts()) == tuple(int(x) for x in omodel.get_raw_counts()), (seed, s, p, name, "counts")
            for key in ("index_meth_fwd", "index_nonmeth_fwd", "index_meth_rev", "index_nonmeth_rev"):
                assert np.asarray(rd[key], dtype=np.int64).tolist() == np.asarray(od[key], dtype=np.int64).tolist(), (seed, s, p, name, key)
            n_checked += 1
        # the bin sum
        rmodel = fmb.motif_model_bin(pile, {n: DNAsequence(q) for n, q in zip(names, seqs)}, Motif(s, p), BetaBernoulliModel(), low, high)
        omodel = osc.motif_model_bin(opile, dict(zip(names, seqs)), OMotif(s, p), OModel(), low, high)
        assert tuple(int(x) for x in rmodel.get_raw_counts()) == tuple(int(x) for x in omodel.get_raw_counts()), (seed, s, p, "bin")
    return f"contigs {lens}, thresholds {low} / {high}: {n_checked} (motif, contig) cases + 40 bin sums"


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    nm = refstub.load_reference()
    bad = 0
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(nm, seed)} ({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:2000]}", flush=True)
    print("scan fuzz against the reference done, mismatches:", bad)
    sys.exit(1 if bad else 0)
