"""Oracle, whole pipeline per bin + process-pool drivers.  TEST INFRASTRUCTURE (see oracle/__init__.py): used by
tests/ and by the ``cpu_baseline`` leg of bench.py only.

``bin_rows`` chains the restated stages for ONE bin of a synthetic metagenome exactly as the reference's task runs
them (find_motifs_bin.py:399-418 pre-filters, :468-596 ``process_subpileup``: search then post-processing); bins are
independent tasks in the reference (:152-171), each seeded afresh (``random.seed(seed)`` per (bin, mod type) for a
plain pileup, once per bin for the bgzip strategy :219-248), so a bin can be computed alone.

The workers rebuild their inputs from the metagenome SPEC (a few numbers): nothing big crosses the process boundary
and the spawn pool works the same on the GPU box's host cores.
"""
from __future__ import annotations

import random
import time

import numpy as np

MODS = ["m", "a", "21839"]          # constants.py:28-37 order


def bin_table(mg, bin_name):
    """Raw pileup rows (all mod types of the spec) of the contigs of one bin as an oracle.pileup table, plus the
    bin's global contig indices."""
    idx = [i for i, x in enumerate(mg.bin_names) if x == bin_name]
    cols = []
    for mt in mg.spec.mod_types:
        c = mg.pileup_columns(mt, contigs=idx)
        c["mod_type"] = np.full(len(c["position"]), MODS.index(mt), dtype=np.int8)
        cols.append(c)
    cat = lambda k: np.concatenate([c[k] for c in cols])
    t = dict(contig=cat("contig_id").astype(np.int64), position=cat("position"), strand=cat("strand"),
             mod_type=cat("mod_type"), fraction_mod=cat("fraction_mod"), Nvalid_cov=cat("nvalid").astype(np.int64))
    return t, idx


def bin_rows(mg, bin_name, seed=1, bgzip_order=False, low=0.3, high=0.7, padding=20, min_kl=0.05, score_threshold=1.5):
    """Motif rows (oracle.postprocess dicts) of one bin: filters -> search -> post-processing."""
    from . import pileup as op
    from . import postprocess as opp
    from . import search as ose
    from .scan import ContigPileup
    t, idx = bin_table(mg, bin_name)
    t = op.prefilter(t)
    seqs = {mg.names[i]: mg.contig_str(i) for i in idx}
    rows = []
    if bgzip_order:
        random.seed(seed)
    for mt_id, mt in enumerate(MODS):
        sel = t["mod_type"] == mt_id
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
        res = ose.find_best_candidates(pile, seqs, mt, low, high, padding, min_kl=min_kl, score_threshold=score_threshold)
        if res is None:
            continue
        out = opp.process_bin(pile, seqs, bin_name, mt, res[0], res[1], padding)
        if out:
            rows += out
    return rows


def _metagenome(spec_kw):
    from nanomotif_amd import synth
    kw = dict(spec_kw)
    kw["mod_types"] = tuple(kw["mod_types"])
    if kw.get("fixed_motifs") is not None:
        kw["fixed_motifs"] = tuple(tuple(m) for m in kw["fixed_motifs"])
    return synth.make_metagenome(synth.SynthSpec(**kw))


def bin_rows_worker(args):
    """Pool task: (spec kwargs, bin name, keyword arguments of ``bin_rows``) -> (bin name, rows, seconds)."""
    spec_kw, bin_name, kw = args
    t0 = time.perf_counter()
    rows = bin_rows(_metagenome(spec_kw), bin_name, **kw)
    return bin_name, rows, time.perf_counter() - t0


def bin_inputs(mg, bin_name, mod_types, min_cov=5):
    """({mod type: {contig: ContigPileup}} with the coverage filter applied, {contig: str}, bin bp) of one bin — the
    inputs of ``score_candidates`` as the scoring benchmarks / parity checks use them (pre-filtered rows)."""
    from nanomotif_amd import synth
    from .scan import ContigPileup
    idx = [i for i, b in enumerate(mg.bin_names) if b == bin_name]
    seqs = {mg.names[i]: mg.contig_str(i) for i in idx}
    piles = {}
    for mt in mod_types:
        piles[mt] = {}
        for i in idx:
            p = mg.contig_pileup(i, mt)
            keep = p["nvalid"] > min_cov
            piles[mt][mg.names[i]] = ContigPileup(p["position"][keep], p["strand"][keep],
                                                  synth.pct_to_fraction(p["pct_hundredths"][keep]))
    return piles, seqs, int(sum(int(mg.lengths[i]) for i in idx))


def score_table(piles, seqs, cands):
    """cands: [(motif string, mod_position, mod type)] -> [[n_mod, n_nomod]] in the order given."""
    from .scan import score_candidates
    out = {}
    for mt in sorted({c[2] for c in cands}):
        these = [(k, s, p) for k, (s, p, m) in enumerate(cands) if m == mt]
        res = score_candidates(piles[mt], seqs, [(s, p) for _, s, p in these])
        for (k, _, _), r in zip(these, res):
            out[k] = r.tolist()
    return [out[k] for k in range(len(cands))]


def score_worker(args):
    """Pool task: (spec kwargs, bin name, candidates) -> (bin name, count table, scan seconds, generation seconds,
    bin bp).  The scan is ``oracle.scan.score_candidates``: regex overlapped finditer + numpy.isin per contig and
    strand, the reference's own primitives (find_motifs_bin.py:1234-1331)."""
    spec_kw, bin_name, cands = args
    t_gen = time.perf_counter()
    mg = _metagenome(spec_kw)
    piles, seqs, bp = bin_inputs(mg, bin_name, sorted({c[2] for c in cands}))
    t0 = time.perf_counter()
    table = score_table(piles, seqs, cands)
    t1 = time.perf_counter()
    return bin_name, table, t1 - t0, t0 - t_gen, bp


def _barrier_worker(rank, jobs, barrier, queue):
    """One of T concurrent CPU-baseline workers (``timed_pool``): build the inputs of all its jobs first, meet the
    others at the barrier, then scan — so the timed region holds scanning only and every worker is busy in it."""
    prepared = []
    for spec_kw, bin_name, cands in jobs:
        mg = _metagenome(spec_kw)
        prepared.append((bin_name, cands) + bin_inputs(mg, bin_name, sorted({c[2] for c in cands})))
    barrier.wait()
    t0 = time.perf_counter()
    out = []
    for bin_name, cands, piles, seqs, bp in prepared:
        out.append((bin_name, score_table(piles, seqs, cands), bp))
    queue.put((rank, out, time.perf_counter() - t0))


def timed_pool(jobs, procs):
    """Run ``jobs`` ((spec kwargs, bin, candidates) each) on ``procs`` concurrent spawn processes the way the
    reference's ``Pool(threads)`` runs one bin per task (find_motifs_bin.py:330-354), jobs dealt round-robin.
    Returns (results by job order [(bin, table, bp)], wall seconds of the scan phase measured by the parent from the
    moment every worker holds its inputs until the last one has delivered, summed per-worker scan seconds)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = max(1, min(procs, len(jobs)))
    barrier = ctx.Barrier(procs + 1)
    queue = ctx.Queue()
    shares = [jobs[r::procs] for r in range(procs)]
    ps = [ctx.Process(target=_barrier_worker, args=(r, shares[r], barrier, queue), daemon=True) for r in range(procs)]
    for p in ps:
        p.start()
    barrier.wait()
    t0 = time.perf_counter()
    got = {}
    cpu_seconds = 0.0
    for _ in range(procs):
        rank, out, secs = queue.get()
        got[rank] = out
        cpu_seconds += secs
    wall = time.perf_counter() - t0
    for p in ps:
        p.join()
    results = [None] * len(jobs)
    for r in range(procs):
        for k, res in enumerate(got[r]):
            results[r + k * procs] = res
    return results, wall, cpu_seconds
