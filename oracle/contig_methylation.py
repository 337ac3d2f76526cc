"""Oracle for the per-contig motif methylation counts (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates what nanomotif_amd.contig_methylation provides: ``motif_model_contig`` (reference
nanomotif/find_motifs_bin.py:1285-1331, restated in oracle/scan.py and pinned by fixture g2) applied to every contig on
its own.  Parity status: the COUNTS are pinned (g2 pins motif_model_contig).  What binnary actually consumes at
main.py:167-178 comes from ``epymetheus.methylation_pattern`` (Rust crate epimetheus-py 0.7.5, not vendored under
/root/reference: median / weighted mean of per-site read fractions, mean coverage): that arithmetic is NOT restated here —
**parity unpinned** for methylation_value as binnary defines it."""
from __future__ import annotations

import numpy as np

from .model import BetaBernoulliModel
from .motif import Motif
from .scan import motif_model_contig


def per_contig_counts(pileup: dict, contigs: dict, motif_string: str, mod_position: int, low=0.3, high=0.7):
    """{contig name: (n_mod, n_nomod)} for the contigs of ``contigs`` (name -> str); contigs without pileup rows count 0."""
    from .scan import ContigPileup
    out = {}
    empty = ContigPileup(np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.float64))
    for name, seq in contigs.items():
        m = motif_model_contig(pileup.get(name, empty), seq, BetaBernoulliModel(), Motif(motif_string, mod_position), low, high)
        out[name] = m.get_raw_counts()
    return out
