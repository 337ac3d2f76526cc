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


def oracle_pipeline(mg, min_motifs_bin=50, seed=1, bgzip_order=False, low=0.3, high=0.7, bins=None):
    """bin-motifs.tsv text computed end to end by the CPU oracle (filters -> search -> post-processing),
    following the task order / seeding of the reference's plain (or bgzip) strategy.  ``bins``: restrict to these
    bins (bins are independent tasks, find_motifs_bin.py:152-171)."""
    from oracle import pipeline as opl
    from oracle import postprocess as opp
    order = []
    for b in mg.bin_names:
        if b not in order and (bins is None or b in bins):
            order.append(b)
    rows = []
    for b in order:
        rows += opl.bin_rows(mg, b, seed=seed, bgzip_order=bgzip_order, low=low, high=high)
    return opp.format_bin_motifs(rows, min_motifs_bin=min_motifs_bin)


def spec_kwargs(spec) -> dict:
    """SynthSpec -> plain dict that survives pickling into a spawn worker (oracle.pipeline workers)."""
    import dataclasses
    return dataclasses.asdict(spec)


def oracle_pipeline_parallel(mg, bins, procs, min_motifs_bin=50, **kw):
    """``oracle_pipeline`` restricted to ``bins`` with one spawn worker per bin (the oracle needs seconds to minutes
    per 2 Mbp bin)."""
    import multiprocessing as mp
    from oracle import pipeline as opl
    from oracle import postprocess as opp
    jobs = [(spec_kwargs(mg.spec), b, kw) for b in bins]
    with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
        res = pool.map(opl.bin_rows_worker, jobs, chunksize=1)
    order = [b for b in dict.fromkeys(mg.bin_names) if b in set(bins)]
    by_bin = {b: rows for b, rows, _ in res}
    rows = [r for b in order for r in by_bin[b]]
    return opp.format_bin_motifs(rows, min_motifs_bin=min_motifs_bin)


# ---------------------------------------------------------------------------------------------
# motif zoo shared by G1/G2 (regex-style strings as the reference uses them)
# ---------------------------------------------------------------------------------------------
def motif_zoo():
    zoo = [
        # literals / palindromes
        ("A", 0), ("C", 0), ("AA", 0), ("AA", 1), ("AAAA", 2), ("GATC", 1), ("GATC", 3), ("CCGG", 1),
        ("GAATTC", 2), ("CTGCAG", 4), ("ACCCA", 4), ("CCAAAT", 4), ("TTCGAA", 5), ("GTAC", 2),
        ("ACGT", 0), ("ACGT", 1), ("ACGT", 2), ("ACGT", 3), ("TTTT", 0), ("CAGAG", 3),
        # gaps and bipartite
        ("GA.TC", 1), ("A.A", 0), ("A.A", 2), ("C..G", 0), ("GCAC......GTT", 2), ("AAC......GTGC", 1),
        ("CAC.....TGG", 1), ("A..........T", 0), ("A...................C", 0),
        ("C....................A....................G", 21),
        # IUPAC sets
        ("CC[AT]GG", 1), ("G[AG].GAAG[CT]", 5), ("[AG]GC[CT]", 2), ("GC.GC", 1), ("[ACG]A[CGT]", 1),
        ("[AC][AC][AC]", 1), ("[CGT]A", 1), ("A[ACT]", 0), ("[AG][CT][AG][CT]", 0), ("[GT]A[AC]..[ACG]C", 1),
        # modified base not canonical / at a bracket (generic API use)
        ("GATC", 0), ("GATC", 2), ("CC[AT]GG", 2), ("TTAA", 0), ("TTAA", 1),
        # flanking dots (search-window form, pad 20)
        ("." * 19 + "GATC" + "." * 18, 20), ("." * 20 + "A" + "." * 20, 20), ("." * 20 + "C" + "." * 20, 20),
        ("." * 18 + "CCAGG" + "." * 18, 19), ("." * 14 + "GCAC......GTT" + "." * 14, 16),
        ("." * 20 + "AATT" + "." * 17, 20), ("..GA.TC..", 3), (".A", 1), ("A.", 0),
    ]
    # seeded extras: random stripped motifs incl. long ones
    for s, p, _ in synth.random_candidates(40, seed=11, mod_types=("a", "m")):
        zoo.append((s, p))
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(20):
        L = int(rng.integers(20, 42))
        chars = ["."] * L
        for q in rng.choice(L, size=int(rng.integers(3, 9)), replace=False):
            chars[int(q)] = "ACGT"[int(rng.integers(4))]
        if chars[0] == "." and chars[-1] == ".":
            chars[0] = "G"
        pos = int(rng.integers(0, L))
        chars[pos] = "AC"[int(rng.integers(2))]
        zoo.append(("".join(chars), pos))
    return zoo
