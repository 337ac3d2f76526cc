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
