"""Scoring batches over random SUBSETS of bins: the per-batch segment table (bins with candidates only) against the static
table of every bin (NM_ALL_SEGMENTS=1), same engine, heavy / light / per-contig batches, many seeds."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from nanomotif_amd import synth
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.motif import Motif
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=600, total_bp=60_000_000, n_bins=60, mod_types=("a", "m"), seed=9))
eng = ScanEngine(0)
eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(len(mg.names))], mg.bin_names)
for mt in ("a", "m"):
    for i in range(len(mg.names)):
        p = mg.contig_pileup(i, mt)
        eng.upload_pileup(mt, np.full(len(p["position"]), i, np.uint32), p["position"], p["strand"], synth.pct_to_fraction(p["pct_hundredths"]), append=i > 0)
bins = sorted(set(mg.bin_names))
rng = np.random.default_rng(4)
bad = n = 0
for seed in range(120):
    k = int(rng.choice([1, 2, 3, 5, 10, 25, 44, 60]))
    some = list(rng.choice(bins, size=k, replace=False))
    style = seed % 4
    if style == 0:        # heavy: many random candidates per bin
        raw = synth.random_candidates(12 * k, seed=500 + seed, mod_types=("a", "m"))
        batch = [(Motif(s, p), mt, some[j % k]) for j, (s, p, mt) in enumerate(raw)]
    elif style == 1:      # light siblings on both mod types
        batch = [(Motif("".join(list("........") + [b] + ["."] + [can] + list("T.G") + list(".......")), 10), mt, bn)
                 for bn in some for mt, can in (("a", "A"), ("m", "C")) for b in "ACG"]
    elif style == 2:      # one candidate per bin, one slot
        batch = [(Motif("GATC", 1), "a", bn) for bn in some]
    else:                 # sets and a general-plane motif
        batch = [(Motif("G[AT]TC", 1), "a", bn) for bn in some] + [(Motif("CC[AT]GG", 1), "m", some[0])]
    os.environ.pop("NM_ALL_SEGMENTS", None)
    own = eng.score(batch)
    per_own = eng.score_per_contig(batch[: min(len(batch), 6)]) if style in (2, 3) else None
    os.environ["NM_ALL_SEGMENTS"] = "1"
    ref = eng.score(batch)
    per_ref = eng.score_per_contig(batch[: min(len(batch), 6)]) if style in (2, 3) else None
    n += 1
    if not np.array_equal(own, ref) or (per_own is not None and any(not np.array_equal(a[1], b[1]) for a, b in zip(per_own, per_ref))):
        bad += 1
        print("MISMATCH seed", seed, "bins", k, "style", style)
print("segment fuzz: %d batches, %d mismatches, nonzero counts in %d" % (n, bad, int((ref.sum(axis=1) > 0).sum())))
eng.close()
