"""Per-contig motif methylation table — the input of binnary's contamination / inclusion calls (reference:
nanomotif/main.py:140-178; one row per (contig, motif) is what ``detect_contamination`` and ``include_contigs`` read,
binnary/data_processing.py:175-189).

The reference fills that table with ``epymetheus.methylation_pattern`` (a Rust crate that is not vendored: per contig
and motif the median or the coverage-weighted mean of the per-site read fractions, the mean coverage and the number
of sites).  This build provides what the SAME scan gives per contig — ``motif_model_contig`` (find_motifs_bin.py:
1285-1331) for every contig of every bin in one launch — i.e. per (contig, motif) the confidently methylated /
unmethylated site counts under the discovery thresholds:

    n_mod, n_nomod          the reference's BetaBernoulli raw counts per contig
    n_motif_obs             n_mod + n_nomod  (sites that carry a confident call)
    methylation_value       n_mod / n_motif_obs  (fraction of confidently called sites that are methylated)

``methylation_value`` is therefore a thresholded stand-in, NOT epymetheus' read-fraction median / weighted mean, and
``mean_read_cov`` is not produced (coverage never reaches the planes).  Stated in DESIGN.md §8; the counts themselves
are bit-exact against the oracle (tests/test_gpu_per_contig.py)."""
from __future__ import annotations

import numpy as np

from .motif import Motif, iupac_to_regex

COLUMNS = ["contig", "motif", "mod_type", "mod_position", "methylation_value", "n_mod", "n_nomod", "n_motif_obs"]


def contig_methylation(engine, motifs, bins=None):
    """motifs: iterable of (IUPAC motif, mod_type, mod_position) — the ``motif_mod`` triples binnary derives from
    bin-motifs.tsv (main.py:129-133).  Every motif is scanned on every resident contig of ``bins`` (default: all bins,
    incl. the contigs of bins that never showed the motif — that is what contamination detection compares).
    Returns a list of dict rows with the COLUMNS above, contigs in engine order per bin; rows without any confident
    site are kept with n_motif_obs = 0 and methylation_value = nan (the reference drops them through its
    n_motif_obs * mean_read_cov filter, main.py:183)."""
    motifs = list(motifs)
    bins = list(engine.bin_names) if bins is None else list(bins)
    cands = [(Motif(iupac_to_regex(m), int(pos)), mt, b) for b in bins for m, mt, pos in motifs]
    if not cands:
        return []
    res = engine.score_per_contig(cands)
    rows = []
    k = 0
    for b in bins:
        for m, mt, pos in motifs:
            names, table = res[k]
            k += 1
            for name, (n_mod, n_nomod) in zip(names, table.tolist()):
                obs = n_mod + n_nomod
                rows.append(dict(contig=name, motif=m, mod_type=mt, mod_position=int(pos),
                                 methylation_value=(n_mod / obs) if obs else float("nan"), n_mod=int(n_mod), n_nomod=int(n_nomod),
                                 n_motif_obs=int(obs)))
    return rows


def write_tsv(rows, path):
    with open(path, "w") as f:
        f.write("\t".join(COLUMNS) + "\n")
        for r in rows:
            f.write("\t".join("" if (isinstance(r[c], float) and np.isnan(r[c])) else str(r[c]) for c in COLUMNS) + "\n")
