"""Oracle restatement of the scan -> count step
(reference: nanomotif/utils.py:44-67, find_motifs_bin.py:1234-1331).  Test infrastructure only.

A bin pileup is ``{contig_name: ContigPileup}``; the reference filters one polars frame per contig
(find_motifs_bin.py:1274) — here the split is done once by the caller.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import regex

from .model import BetaBernoulliModel
from .motif import Motif


@dataclass
class ContigPileup:
    position: np.ndarray       # int64, rows in pileup order
    strand: np.ndarray         # uint8 ASCII '+' / '-'
    fraction_mod: np.ndarray   # float64 = percent / 100 (dataload.py:85)

    def __len__(self):
        return len(self.position)


def subseq_indices(subseq: str, seq: str) -> np.ndarray:
    """utils.py:44-67 — start offsets of all (overlapping) regex matches, ascending int64."""
    pattern = regex.compile(subseq)
    return np.fromiter((m.start() for m in pattern.finditer(seq, overlapped=True)), dtype=np.int64)


def methylated_motif_occourances(motif: Motif, sequence: str, methylated_positions, non_methylated_positions):
    """find_motifs_bin.py:1234-1263."""
    assert len(motif) > 0 and len(sequence) > 0 and type(motif) is Motif
    motif_index = subseq_indices(motif.string, sequence) + motif.mod_position
    meth = methylated_positions[np.isin(methylated_positions, motif_index, assume_unique=True)]
    non = non_methylated_positions[np.isin(non_methylated_positions, motif_index, assume_unique=True)]
    return meth, non


def split_positions(p: ContigPileup, low, high):
    """find_motifs_bin.py:1308-1314: meth = frac >= high, nonmeth = frac <= low, per strand."""
    hi = p.fraction_mod >= high
    lo = p.fraction_mod <= low
    plus = p.strand == ord("+")
    minus = p.strand == ord("-")
    return (p.position[hi & plus], p.position[lo & plus], p.position[hi & minus], p.position[lo & minus])


def motif_model_contig(pileup: ContigPileup, contig: str, prior: BetaBernoulliModel, motif: Motif,
                       low_meth_threshold=0.3, high_meth_threshold=0.7, save_motif_positions=False):
    """find_motifs_bin.py:1285-1331."""
    stripped = motif.new_stripped_motif()
    mf, nf, mr, nr = split_positions(pileup, low_meth_threshold, high_meth_threshold)
    i_mf, i_nf = methylated_motif_occourances(stripped, contig, mf, nf)
    i_mr, i_nr = methylated_motif_occourances(stripped.reverse_compliment(), contig, mr, nr)
    prior.update(len(i_mf) + len(i_mr), len(i_nf) + len(i_nr))
    if save_motif_positions:
        return prior, dict(index_meth_fwd=i_mf, index_nonmeth_fwd=i_nf, index_meth_rev=i_mr, index_nonmeth_rev=i_nr)
    return prior


_EMPTY = ContigPileup(np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.float64))


def motif_model_bin(pileup: dict, contigs: dict, motif: Motif, model: BetaBernoulliModel,
                    low_meth_threshold, high_meth_threshold):
    """find_motifs_bin.py:1265-1283 — ``contigs`` maps name -> sequence string; accumulates into ``model``."""
    for name, seq in contigs.items():
        model = motif_model_contig(pileup.get(name, _EMPTY), seq, model, motif,
                                   low_meth_threshold=low_meth_threshold, high_meth_threshold=high_meth_threshold)
    return model


def score_candidates(pileup: dict, contigs: dict, candidates, low=0.3, high=0.7):
    """Counts table int64[n,2] for [(motif_string, mod_position), ...] — the batch form the HIP path exposes.

    Per contig the four position arrays are split once and reused for every candidate; the result is
    identical to calling ``motif_model_bin`` per candidate with a fresh model."""
    out = np.zeros((len(candidates), 2), dtype=np.int64)
    for name, seq in contigs.items():
        mf, nf, mr, nr = split_positions(pileup.get(name, _EMPTY), low, high)
        for k, (s, p) in enumerate(candidates):
            st = Motif(s, p).new_stripped_motif()
            a, b = methylated_motif_occourances(st, seq, mf, nf)
            c, d = methylated_motif_occourances(st.reverse_compliment(), seq, mr, nr)
            out[k, 0] += len(a) + len(c)
            out[k, 1] += len(b) + len(d)
    return out
