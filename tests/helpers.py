"""Shared test helpers: fixtures, synthetic inputs, oracle-format conversion."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np

from nanomotif_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def sha1(arr) -> str:
    return hashlib.sha1(np.ascontiguousarray(arr).tobytes()).hexdigest()


def spec_from_json(d) -> synth.SynthSpec:
    kw = dict(d)
    if "mod_types" in kw:
        kw["mod_types"] = tuple(kw["mod_types"])
    if kw.get("fixed_motifs") is not None:
        kw["fixed_motifs"] = tuple(tuple(m) for m in kw["fixed_motifs"])
    return synth.SynthSpec(**kw)


def oracle_bin_inputs(mg, mod_type, contigs=None, min_cov=5):
    """(pileup dict name->ContigPileup, contigs dict name->str) with the coverage filter applied."""
    from oracle.scan import ContigPileup
    idx = range(len(mg.names)) if contigs is None else contigs
    pile, seqs = {}, {}
    for i in idx:
        p = mg.contig_pileup(i, mod_type)
        keep = p["nvalid"] > min_cov
        pile[mg.names[i]] = ContigPileup(p["position"][keep], p["strand"][keep],
                                         synth.pct_to_fraction(p["pct_hundredths"][keep]))
        seqs[mg.names[i]] = mg.contig_str(i)
    return pile, seqs


def oracle_pipeline(mg, min_motifs_bin=50, seed=1, bgzip_order=False, low=0.3, high=0.7):
    """bin-motifs.tsv text computed end to end by the CPU oracle (filters -> search -> post-processing),
    following the task order / seeding of the reference's plain (or bgzip) strategy."""
    import random
    from oracle import pileup as op
    from oracle import postprocess as opp
    from oracle import search as ose
    from oracle.scan import ContigPileup
    mods = ["m", "a", "21839"]
    cols = []
    for mt in mg.spec.mod_types:
        c = mg.pileup_columns(mt)
        c["mod_type"] = np.full(len(c["position"]), mods.index(mt), dtype=np.int8)
        cols.append(c)
    cat = lambda k: np.concatenate([c[k] for c in cols])
    t = dict(contig=cat("contig_id").astype(np.int64), position=cat("position"), strand=cat("strand"),
             mod_type=cat("mod_type"), fraction_mod=cat("fraction_mod"), Nvalid_cov=cat("nvalid").astype(np.int64))
    t = op.prefilter(t)
    rows = []
    bins = []
    for b in mg.bin_names:
        if b not in bins:
            bins.append(b)
    for b in bins:
        idx = [i for i, x in enumerate(mg.bin_names) if x == b]
        seqs = {mg.names[i]: mg.contig_str(i) for i in idx}
        if bgzip_order:
            random.seed(seed)
        for mt_id, mt in enumerate(mods):
            sel = (t["mod_type"] == mt_id) & np.isin(t["contig"], idx)
            if not sel.any():
                continue
            pile = {}
            for i in idx:
                s = sel & (t["contig"] == i)
                if s.any():
                    o = np.argsort(t["position"][s], kind="stable")
                    pile[mg.names[i]] = ContigPileup(t["position"][s][o], t["strand"][s][o], t["fraction_mod"][s][o])
            if not bgzip_order:
                random.seed(seed)
            res = ose.find_best_candidates(pile, seqs, mt, low, high, 20, min_kl=0.05, score_threshold=1.5)
            if res is None:
                continue
            out = opp.process_bin(pile, seqs, b, mt, res[0], res[1], 20)
            if out:
                rows += out
    return opp.format_bin_motifs(rows, min_motifs_bin=min_motifs_bin)
