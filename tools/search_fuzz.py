"""Differential fuzz of the native lock-step search + post-processing (csrc/nmsearch.cpp, nmpost.cpp through their callback
entry points) against the Python coroutines (search.py, postprocess.py) on random small bins: random planted motifs (literal,
bracketed, gapped, palindromes, short), random methylation rates, both mod types, several bins advanced together.  CPU only
(the scorer of both sides is oracle/scan.py).   usage: python3 tools/search_fuzz.py [first_seed [n_seeds]]"""
import random
import sys
import time

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np

from helpers import oracle_bin_inputs
from nanomotif_amd import native_search as ns
from nanomotif_amd import postprocess as pp
from nanomotif_amd import search as ps
from nanomotif_amd import synth
from test_host_search import windows_for
from test_native_search import _backends

PAD = 20
POOL = {"a": [("GATC", 1), ("CCAAAT", 4), ("ACCCA", 4), ("GAAGNNNNNNTAC", 2), ("RGATCY", 2), ("GANTC", 1), ("CAG", 1), ("TTAA", 3), ("GTAC", 2),
              ("CAMNNNNNNGTG", 1), ("AT", 0), ("GCAGC", 2), ("AAGNNNNNCTC", 1)],
        "m": [("CCWGG", 1), ("GGCC", 2), ("GCGC", 1), ("CCGG", 0), ("ACGT", 1), ("CCSGG", 1), ("GCNGC", 1), ("TCGA", 1), ("C", 0), ("RCCGGY", 2), ("CTAG", 0)]}


def row_tuple(r):
    c = r.complement
    return (r.motif, r.mod_position, r.n_mod, r.n_nomod, r.score, None if c is None else (c.motif, c.mod_position, c.n_mod, c.n_nomod, c.score),
            r.motif_iupac, r.mod_position_iupac, r.has_complement_columns)


def one(seed):
    rng = np.random.default_rng(seed)
    n_bins = int(rng.integers(1, 4))
    keys, piles, seqs_by_bin, wins = [], {}, {}, {}
    for b in range(n_bins):
        mts = ("a", "m") if rng.random() < 0.5 else (("a",) if rng.random() < 0.5 else ("m",))
        fixed = []
        for mt in mts:
            for k in rng.choice(len(POOL[mt]), size=int(rng.integers(0, 3)), replace=False):
                fixed.append((POOL[mt][k][0], POOL[mt][k][1], mt))
        spec = synth.SynthSpec(n_contigs=int(rng.integers(1, 4)), total_bp=int(rng.integers(40_000, 140_000)), n_bins=1, mod_types=mts,
                               seed=int(rng.integers(0, 1 << 30)), min_contig_bp=12_000, fixed_motifs=tuple(fixed))
        mg = synth.make_metagenome(spec)
        for mt in mts:
            pile, seqs = oracle_bin_inputs(mg, mt)
            key = (f"bin{b}", mt)
            random.seed(int(rng.integers(0, 1000)))
            w = windows_for(mg, mt, pile)
            if w is None or w[0] is None or len(w[0]) == 0:
                continue
            keys.append(key)
            piles[key], seqs_by_bin[f"bin{b}"] = pile, seqs
            wins[key] = w
    if not keys:
        return "no windows"
    min_kl, thr = float(rng.choice([0.05, 0.02, 0.1])), float(rng.choice([1.5, 1.0, 2.0]))
    # Python coroutines (search, then the post-processing chain)
    store = ps.HostWindowStore()
    for key in keys:
        store.add_task(key, wins[key][0].copy())
    score_fn, window_fn = _backends(keys, piles, seqs_by_bin, store)
    tasks = {key: ps.find_best_candidates_co(wins[key][1], key[1], PAD, min_kl=min_kl, score_threshold=thr) for key in keys}
    want = ps.run_lockstep(tasks, lambda flat: score_fn([(keys.index(k), m) for k, m, _ in flat]), store.execute)
    stages = {k: {} for k in keys}
    ptasks = {}
    for key in keys:
        if want[key] is not None:
            g, best, _ = want[key]
            ptasks[key] = pp.postprocess_co(g, best, key[0], key[1], PAD, on_stage=lambda name, rows, key=key: stages[key].__setitem__(name, list(rows)))
    pwant = ps.run_lockstep(ptasks, lambda flat: score_fn([(keys.index(k), m) for k, m, _ in flat]))
    # native
    store2 = ps.HostWindowStore()
    for key in keys:
        store2.add_task(key, wins[key][0].copy())
    score_fn2, window_fn2 = _backends(keys, piles, seqs_by_bin, store2)
    res = ns.find_best_candidates_custom([(k, store2.totals[k], wins[k][1]) for k in keys], PAD, min_kl, thr, score_fn2, window_fn2)
    for t, key in enumerate(keys):
        r = res.result(t, full_graph=True)
        assert (r is None) == (want[key] is None), (seed, key, "one side found nothing")
        if r is None:
            continue
        graph, best, _ = r
        wg, wbest, _ = want[key]
        assert list(graph.nodes) == list(wg.nodes) and best == wbest, (seed, key, "graph / best differ")
        for n in graph.nodes:
            a, b = graph.nodes[n], wg.nodes[n]
            assert a["model"].get_raw_counts() == b["model"].get_raw_counts() and a["score"] == b["score"] and a["priority"] == b["priority"] \
                and a["depth"] == b["depth"] and a["visited"] == b["visited"], (seed, key, n, a, b)
        assert sorted(graph.edges()) == sorted(wg.edges()), (seed, key, "edges differ")
    post = res.postprocess_custom(score_fn2)
    n_rows = 0
    for t, key in enumerate(keys):
        for s, name in enumerate(ns.PostResults.STAGES):
            got = sorted((row_tuple(r) for r in post.rows(t, s)), key=repr)
            exp = sorted((row_tuple(r) for r in stages[key].get(name, [])), key=repr)
            assert got == exp, (seed, key, name, got, exp)
        if pwant.get(key):
            assert pp.format_bin_motifs(post.final(t)) == pp.format_bin_motifs(pwant[key]), (seed, key)
            n_rows += len(pwant[key])
        else:
            assert post.final(t) is None, (seed, key)
    res.close()
    return f"{len(keys)} tasks, {n_rows} final rows"


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = 0
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(seed)} ({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:600]}", flush=True)
    print("search fuzz done, mismatches:", bad)
    sys.exit(1 if bad else 0)
