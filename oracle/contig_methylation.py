"""Oracle for the per-contig methylation tables (TEST INFRASTRUCTURE, see oracle/__init__.py).

(1) ``per_contig_counts``: ``motif_model_contig`` (reference nanomotif/find_motifs_bin.py:1285-1331, restated in
    oracle/scan.py and pinned by fixture g2) applied to every contig on its own.  Parity: pinned through g2.

(2) ``read_methylation``: what binnary consumes at nanomotif/main.py:167-193 — the output of
    ``epymetheus.methylation_pattern``.  That function lives in the third-party Rust crate epimetheus-py, pinned at
    0.7.5 by the reference (setup.py:38, pixi.toml:20); its source is NOT under /root/reference and the image has no
    network, so the algorithm below is restated from the crate's published behaviour and from what the reference's call
    site and tests fix (main.py:159: columns contig, motif, mod_type, mod_position, methylation_value, mean_read_cov,
    n_motif_obs; tests/binnary/test_utils.py:38-59: two motifs x two contigs -> four rows):
      * a pileup record (bedMethyl columns 1 contig, 2 start, 4 mod code, 6 strand, 10 N_valid_cov, 12 N_mod, 17 N_diff)
        is kept iff N_valid_cov >= min_valid_read_coverage and N_valid_cov / (N_valid_cov + N_diff) >=
        min_valid_cov_to_diff_fraction (the reference's own dataload.py:177-178 uses the same ratio);
      * per contig and motif: sites = start + mod_position of every match of the motif's regex on the contig ('+'
        records) and of its reverse complement with mod_position' = len - mod_position - 1 ('-' records) — matching as in
        utils.py:44-67 (all matches, overlapping included);
      * over the sites that carry a kept record of the motif's mod code: n_motif_obs = their number, mean_read_cov =
        mean N_valid_cov, methylation_value = median of N_mod / N_valid_cov (mean of the two middle values for an even
        count) or, for WeightedMean, sum(N_mod) / sum(N_valid_cov); (contig, motif) pairs without such a site give no row.
    **Parity unpinned** (third-party source absent, no golden vector in the reference): the product is pinned to THIS
    restatement (tests/test_gpu_read_methylation.py), not to epimetheus itself."""
from __future__ import annotations

import numpy as np
import regex

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


def _sites(seq: str, motif: Motif) -> np.ndarray:
    return np.fromiter((m.start() for m in regex.finditer(motif.string, seq, overlapped=True)), dtype=np.int64) + motif.mod_position


def read_methylation(records: dict, contigs: dict, motifs, min_valid_read_coverage=3, min_valid_cov_to_diff_fraction=0.8,
                     output_type="median"):
    """records: {(contig name, mod_type): dict(position, strand (uint8 ASCII), n_valid, n_mod, n_diff)} numpy columns;
    contigs: name -> str; motifs: (regex-style motif string, mod_type, mod_position) triples.  Returns dict rows
    (contig, motif index, n_motif_obs, mean_read_cov, methylation_value), motif-major, contigs in dict order."""
    rows = []
    for k, (string, mod_type, pos) in enumerate(motifs):
        fwd = Motif(string, pos).new_stripped_motif()
        rev = fwd.reverse_compliment()
        for name, seq in contigs.items():
            r = records.get((name, mod_type))
            if r is None or len(r["position"]) == 0:
                continue
            nv, nd = r["n_valid"].astype(np.float64), r["n_diff"].astype(np.float64)
            keep = (r["n_valid"] >= min_valid_read_coverage) & (r["n_valid"] > 0)
            with np.errstate(invalid="ignore", divide="ignore"):
                keep &= nv / (nv + nd) >= min_valid_cov_to_diff_fraction
            cov, mod = [], []
            for strand, motif in ((ord("+"), fwd), (ord("-"), rev)):
                sel = keep & (r["strand"] == strand)
                at = np.isin(r["position"][sel], _sites(seq, motif))
                cov.append(r["n_valid"][sel][at])
                mod.append(r["n_mod"][sel][at])
            cov, mod = np.concatenate(cov).astype(np.int64), np.concatenate(mod).astype(np.int64)
            if len(cov) == 0:
                continue
            frac = np.sort(mod.astype(np.float64) / cov.astype(np.float64))
            n = len(frac)
            median = frac[n // 2] if n % 2 else (frac[n // 2 - 1] + frac[n // 2]) / 2.0
            value = median if output_type == "median" else float(mod.sum()) / float(cov.sum())
            rows.append(dict(contig=name, motif=k, n_motif_obs=n, mean_read_cov=float(cov.sum()) / n, methylation_value=float(value)))
    return rows
