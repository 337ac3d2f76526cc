"""Search fuzz against the REFERENCE'S OWN find_best_candidates / MotifSearcher.run (build container only: imports
/root/reference through refstub, like gen_golden.py's g4): random small bins (contigs, planted motifs, methylation rates,
search parameters) searched by the reference with random.seed(seed), and by the product's native lock-step machine
(nm_search_run_custom, scored by the CPU oracle, windows from the product's extraction): every graph node in order (motif, counts,
score, priority, depth, visited), the edges and the best candidates must be equal.
usage: python3 tests/golden/search_ref_fuzz.py [first_seed [n_seeds]]"""
import os
import random
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
if os.environ.get("PYTHONHASHSEED") != "0":                      # (set order inside the reference, as in gen_golden.py)
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)
import numpy as np

import gen_golden as G
import refstub
from helpers import oracle_bin_inputs
from nanomotif_amd import native_search as ns
from nanomotif_amd import search as ps
from nanomotif_amd import synth
from test_host_search import windows_for
from test_native_search import _backends

POOL = {"a": [("GATC", 1), ("CCAAAT", 4), ("ACCCA", 4), ("GAAGNNNNNNTAC", 2), ("RGATCY", 2), ("GANTC", 1), ("CAG", 1), ("TTAA", 3), ("GTAC", 2),
              ("CAMNNNNNNGTG", 1), ("GCAGC", 2), ("AAGNNNNNCTC", 1), ("GRNGAAGY", 5)],
        "m": [("CCWGG", 1), ("GGCC", 2), ("GCGC", 1), ("CCGG", 0), ("ACGT", 1), ("CCSGG", 1), ("GCNGC", 1), ("TCGA", 1), ("RCCGGY", 2), ("CTAG", 0)]}


def make_case(seed):
    """(SynthSpec kwargs, mod type, min_kl, score threshold, random seed) of fuzz case ``seed`` — also what gen_golden.py's g11 records."""
    rng = np.random.default_rng(seed)
    mt = "a" if rng.random() < 0.55 else "m"
    fixed = tuple((POOL[mt][k][0], POOL[mt][k][1], mt) for k in rng.choice(len(POOL[mt]), size=int(rng.integers(0, 4)), replace=False))
    n_contigs, total_bp = int(rng.integers(1, 4)), int(rng.integers(50_000, 260_000))
    kw = dict(n_contigs=n_contigs, total_bp=total_bp, n_bins=1, mod_types=(mt,), seed=int(rng.integers(0, 1 << 30)),
              min_contig_bp=min(12_000, total_bp // (2 * n_contigs)), fixed_motifs=fixed, methylated_fraction=float(rng.choice([0.97, 0.9, 0.8])))
    min_kl, thr, rseed = float(rng.choice([0.05, 0.05, 0.02, 0.1])), float(rng.choice([1.5, 1.5, 1.0, 2.0])), int(rng.choice([1, 1, 7, 123]))
    return kw, mt, min_kl, thr, rseed


def one(nm, seed):
    from nanomotif.seq import DNAsequence
    fmb = nm.find_motifs_bin
    kw, mt, min_kl, thr, rseed = make_case(seed)
    fixed = kw["fixed_motifs"]
    spec = synth.SynthSpec(**kw)
    mg = synth.make_metagenome(spec)
    # ---- the reference
    cols = G.filtered_bin_pileup(mg, mt)
    names = np.array(mg.names, dtype=object)[cols["contig_id"]]
    pile = refstub.make_pileup(names, cols["position"], [chr(c) for c in cols["strand"].tolist()], cols["fraction_mod"], mod_type=[mt] * len(names))
    seqs = {n: DNAsequence(mg.contig_str(i)) for i, n in enumerate(mg.names)}
    tmp = tempfile.mkdtemp()
    random.seed(rseed)
    ref = fmb.find_best_candidates(pile, seqs, mt, "bin0", tmp, low_meth_threshold=0.3, high_meth_threshold=0.7, padding=20, min_kl=min_kl,
                                   max_dead_ends=25, max_rounds_since_new_best=30, score_threshold=thr)
    # ---- the product's native machine on the oracle's scan
    opile, oseqs = oracle_bin_inputs(mg, mt)
    random.seed(rseed)
    windows = windows_for(mg, mt, opile, 0.7, 20)
    if windows is None or windows[0] is None or len(windows[0]) == 0:
        assert ref is None, (seed, "the product found no windows, the reference searched")
        return f"{spec.total_bp} bp {mt} {fixed}: no methylated windows on either side"
    key = ("bin0", mt)
    store = ps.HostWindowStore()
    store.add_task(key, windows[0])
    score_fn, window_fn = _backends([key], {key: opile}, {"bin0": oseqs}, store)
    res = ns.find_best_candidates_custom([(key, store.totals[key], windows[1])], 20, min_kl, thr, score_fn, window_fn)
    got = res.result(0, full_graph=True)
    assert (ref is None) == (got is None), (seed, "one side found nothing", ref is None, got is None)
    if ref is None:
        return f"{spec.total_bp} bp {mt} {fixed}: both found nothing"
    rg, rbest = ref
    graph, best, _ = got
    rnodes = [(n.string, int(n.mod_position)) for n in rg.nodes]
    assert [(n.string, n.mod_position) for n in graph.nodes] == rnodes, (seed, "node order", len(graph.nodes), len(rnodes))
    for (n, d), (rn, rd) in zip(graph.nodes.items(), rg.nodes(data=True)):
        assert list(d["model"].get_raw_counts()) == G.model_counts(rd["model"]), (seed, n, "counts")
        assert abs(d["score"] - float(rd["score"])) <= 1e-9 * max(1.0, abs(d["score"])), (seed, n, "score", d["score"], rd["score"])
        assert abs(d["priority"] - float(rd["priority"])) <= 1e-12 * max(1.0, abs(d["priority"])), (seed, n, "priority")
        assert d["depth"] == int(rd["depth"]) and d["visited"] == bool(rd["visited"]), (seed, n, "depth / visited")
    assert sorted((u.string, v.string) for u, v in graph.edges()) == sorted((u.string, v.string) for u, v in rg.edges()), (seed, "edges")
    assert sorted((m.string, m.mod_position) for m in best) == sorted((m.string, int(m.mod_position)) for m in rbest), (seed, "best")
    res.close()
    return f"{spec.total_bp} bp {mt} kl {min_kl} thr {thr} seed {rseed}: {len(rnodes)} nodes, best {[m.new_stripped_motif().string if hasattr(m, 'new_stripped_motif') else m.string for m in rbest][:6]}"


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    nm = refstub.load_reference()
    bad = 0
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(nm, seed)} ({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:2000]}", flush=True)
    print("search fuzz against the reference done, mismatches:", bad)
    sys.exit(1 if bad else 0)
