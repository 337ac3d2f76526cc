"""Oracle restatement of the pileup pre-filters (reference: nanomotif/dataload.py:191-247).
Test infrastructure only.  Pinned to OUTPUTS OF THE REFERENCE'S OWN FUNCTIONS run in the build container over the frame stand-in
tests/golden/refframe.py (real polars is absent here): fixture g10 (coverage + frequency filters on every strict bound) and
fixture g14 (the adjacency filter at the production distance 8 and at 1, 3 on gapped, tied, null-bearing, mixed-mod-type rows;
refframe's Expr.rolling restates the polars definition and is itself checked against the two known answers of
tests/test_dataload.py:37-69 before g14 is recorded), plus those two known answers literally (tests/test_oracle_golden.py).

A pileup table is a dict of equal-length numpy columns:
``contig`` (any hashable dtype), ``position`` int64, ``strand`` (uint8 ASCII or str), ``mod_type``,
``fraction_mod`` float64 (NaN = null percentage: polars keeps such a row in ``pl.count()``, every comparison with it
is null = not kept), ``Nvalid_cov`` int.
"""
from __future__ import annotations

import numpy as np


def _take(t, mask_or_idx):
    return {k: v[mask_or_idx] for k, v in t.items()}


def filter_pileup(t, min_coverage=5):
    """dataload.py:191-200 — strict ``Nvalid_cov > 5`` (the CLI flag is not forwarded, main.py:69-83)."""
    return _take(t, t["Nvalid_cov"] > min_coverage)


def filter_pileup_minimummod_frequency(t, methylation_threshold=0.7, min_mod_frequency=0.0001, min_mods_pr_contig=50):
    """dataload.py:202-226 — per (contig, mod_type): #(frac > thr)/#rows > 1e-4 and #(frac > thr) > 50."""
    keys = {}
    key_id = np.empty(len(t["position"]), dtype=np.int64)
    for i, k in enumerate(zip(t["contig"].tolist(), t["mod_type"].tolist())):
        key_id[i] = keys.setdefault(k, len(keys))
    n = np.bincount(key_id, minlength=len(keys))
    n_mod = np.bincount(key_id, weights=(t["fraction_mod"] > methylation_threshold), minlength=len(keys)).astype(np.int64)
    ok = np.zeros(len(keys), dtype=bool)
    nz = n > 0
    ok[nz] = ((n_mod[nz] / n[nz]) > min_mod_frequency) & (n_mod[nz] > min_mods_pr_contig)
    return _take(t, ok[key_id])


def filter_pileup_adjacency_filter(t, methylation_threshold=0.7, adjacency_distance=8):
    """dataload.py:228-247 — per (contig, strand) — mod types mixed — keep a row iff its fraction equals the
    max over rows with position in [p-d, p+d] or it is below the threshold.  Output: groups concatenated in
    (contig, strand) order, each sorted by position (the reference's group order is unspecified)."""
    d = adjacency_distance
    order = np.argsort(t["position"], kind="stable")
    t = _take(t, order)
    groups = {}
    for i, k in enumerate(zip(t["contig"].tolist(), t["strand"].tolist())):
        groups.setdefault(k, []).append(i)
    keep_idx = []
    for k in sorted(groups, key=lambda x: (str(x[0]), str(x[1]))):
        idx = np.array(groups[k])
        pos = t["position"][idx]
        frac = t["fraction_mod"][idx]
        lo = np.searchsorted(pos, pos - d, side="left")
        hi = np.searchsorted(pos, pos + d, side="right")
        # list.max() skips nulls (NaN here); a null row itself satisfies neither comparison
        with np.errstate(invalid="ignore"):
            wmax = np.array([np.fmax.reduce(frac[a:b]) for a, b in zip(lo, hi)]) if len(idx) else frac
            keep = (frac == wmax) | (frac < methylation_threshold)
        keep_idx.append(idx[keep])
    keep_idx = np.concatenate(keep_idx) if keep_idx else np.zeros(0, dtype=np.int64)
    return _take(t, keep_idx)


def prefilter(t):
    """The three filters in the order find_motifs_bin.py:399-414 applies them."""
    return filter_pileup_adjacency_filter(filter_pileup_minimummod_frequency(filter_pileup(t)))
