"""The native lock-step search (csrc/nmsearch.cpp) on CPU: nm_search_run_custom driven by the oracle's scan and the
numpy window store must reproduce the reference-recorded search traces (tests/golden/g4_search.json) node for node,
agree with the Python coroutine search on several tasks advanced together, and its digamma must be scipy's bit for bit."""
import random

import os

import numpy as np
import pytest

from helpers import load_golden, oracle_bin_inputs, spec_from_json
from nanomotif_amd import native_search as ns
from nanomotif_amd import search as ps
from nanomotif_amd import synth
from nanomotif_amd.find_motifs_bin import LockstepScorer
from test_host_search import oracle_backend, windows_for


def _backends(keys, piles, seqs_by_bin, store):
    from oracle.scan import score_candidates

    def score_fn(reqs):
        out = np.zeros((len(reqs), 2), dtype=np.int64)
        for i, (t, m) in enumerate(reqs):
            key = keys[t]
            out[i] = score_candidates(piles[key], seqs_by_bin[key[0]], [(m.string, m.mod_position)])[0]
        return out

    def window_fn(reqs):
        res = store.execute([(keys[t], ps.WinReq(kind, m)) for t, kind, m in reqs])
        out = np.zeros((len(reqs), 2 + 4 * 64), dtype=np.int32)
        for i, ((t, kind, m), r) in enumerate(zip(reqs, res)):
            if kind == "remove":
                out[i, 0], out[i, 1] = r
            else:
                out[i, 0] = r[0]
                if r[1] is not None:
                    out[i, 2:].reshape(4, 64)[:, :r[1].shape[1]] = r[1]
        return out
    return score_fn, window_fn


@pytest.mark.parametrize("name", ["gatc_single", "ecoli_like_m", "ecoli_like_a", "geobacillus_like", "no_motif"])
def test_native_search_reproduces_reference_trace(name):
    g = load_golden("g4_search.json")[name]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    mt = g["mod_type"]
    pile, seqs = oracle_bin_inputs(mg, mt)
    P = g["params"]
    random.seed(P["seed"])
    windows = windows_for(mg, mt, pile, P["high"], P["padding"])
    key = ("bin0", mt)
    store = ps.HostWindowStore()
    store.add_task(key, windows[0])
    score_fn, window_fn = _backends([key], {key: pile}, {"bin0": seqs}, store)
    res = ns.find_best_candidates_custom([(key, store.totals[key], windows[1])], P["padding"], P["min_kl"], P["score_threshold"],
                                         score_fn, window_fn)
    graph, best, _ = res.result(0, full_graph=True)
    assert [(n.string, n.mod_position) for n in graph.nodes] == [(r["motif"], r["pos"]) for r in g["nodes"]]
    for (n, d), r in zip(graph.nodes.items(), g["nodes"]):
        assert list(d["model"].get_raw_counts()) == r["counts"]
        assert d["score"] == pytest.approx(r["score"], abs=1e-9, rel=1e-9)
        assert d["priority"] == pytest.approx(r["priority"], abs=1e-12, rel=1e-12)
        assert d["depth"] == r["depth"] and d["visited"] == r["visited"]
    assert sorted((u.string, v.string) for u, v in graph.edges()) == sorted(map(tuple, g["edges"]))
    assert sorted((m.string, m.mod_position) for m in best) == sorted(map(tuple, g["best"]))
    # the light result (what post-processing reads) carries the same best candidates with the same models and scores
    light, best2, _ = ns.find_best_candidates_custom([(key, store.totals[key], windows[1])], P["padding"], P["min_kl"], P["score_threshold"],
                                                     *_fresh_backends(mg, mt, pile, seqs, P, key)).result(0)
    assert best2 == best and set(light.nodes) == set(best)
    for m in best:
        assert light.nodes[m]["score"] == graph.nodes[m]["score"] and light.nodes[m]["model"].get_raw_counts() == graph.nodes[m]["model"].get_raw_counts()


def _fresh_backends(mg, mt, pile, seqs, P, key):
    random.seed(P["seed"])
    windows = windows_for(mg, mt, pile, P["high"], P["padding"])
    store = ps.HostWindowStore()
    store.add_task(key, windows[0])
    return _backends([key], {key: pile}, {"bin0": seqs}, store)


@pytest.mark.parametrize("memo", [False, True])
def test_native_lockstep_equals_python_coroutines_bit_for_bit(memo, monkeypatch):
    """Three tasks advanced together: graphs (node order, counts, depth, visited), best lists, and every score and
    priority EXACTLY equal to the Python coroutine path's float64 values.  Without the per-task memo of scored motifs
    (NM_SEARCH_NO_MEMO) also the same number of rounds and candidates; with it (the default) fewer of both."""
    if not memo:
        monkeypatch.setenv("NM_SEARCH_NO_MEMO", "1")
    g4 = load_golden("g4_search.json")
    keys, piles, seqs_by_bin, wins = [], {}, {}, {}
    for bin_name, gname in (("binA", "geobacillus_like"), ("binB", "ecoli_like_m"), ("binC", "ecoli_like_a")):
        g = g4[gname]
        mg = synth.make_metagenome(spec_from_json(g["spec"]))
        mt = g["mod_type"]
        pile, seqs = oracle_bin_inputs(mg, mt)
        key = (bin_name, mt)
        keys.append(key)
        piles[key], seqs_by_bin[bin_name] = pile, seqs
        random.seed(1)
        wins[key] = windows_for(mg, mt, pile)
    # Python coroutines
    store = ps.HostWindowStore()
    tasks = {}
    for key in keys:
        store.add_task(key, wins[key][0].copy())
        tasks[key] = ps.find_best_candidates_co(wins[key][1], key[1], 20, min_kl=0.05, score_threshold=1.5)
    scorer = LockstepScorer(oracle_backend(piles, seqs_by_bin))
    want = ps.run_lockstep(tasks, scorer, store.execute)
    # native
    store2 = ps.HostWindowStore()
    for key in keys:
        store2.add_task(key, wins[key][0].copy())
    score_fn, window_fn = _backends(keys, piles, seqs_by_bin, store2)
    res = ns.find_best_candidates_custom([(k, store2.totals[k], wins[k][1]) for k in keys], 20, 0.05, 1.5, score_fn, window_fn)
    if memo:
        assert res.rounds < scorer.rounds and res.candidates < scorer.candidates
    else:
        assert (res.rounds, res.candidates) == (scorer.rounds, scorer.candidates)
    for t, key in enumerate(keys):
        graph, best, _ = res.result(t, full_graph=True)
        wg, wbest, _ = want[key]
        assert list(graph.nodes) == list(wg.nodes) and best == wbest
        for n in graph.nodes:
            a, b = graph.nodes[n], wg.nodes[n]
            assert a["model"].get_raw_counts() == b["model"].get_raw_counts()
            assert a["score"] == b["score"] and a["priority"] == b["priority"], (n, a["score"], b["score"])
            assert a["depth"] == b["depth"] and a["visited"] == b["visited"]
        assert sorted(graph.edges()) == sorted(wg.edges())
        # the search graph's GML text made straight from the result arrays (what a run with --out writes) = the graph object's
        # (round 6: the text comes from the library, nm_search_result_gml; NANOMOTIF_PY_GML=1: built in Python from the exported arrays)
        assert res.artifacts(t)[0].gml_text() == graph.gml_text() and graph.gml_text().count("node [") == len(graph.nodes)
        assert res.artifacts(t)[2] is res.pssms[t]
        os.environ["NANOMOTIF_PY_GML"] = "1"
        try:
            assert res.artifacts(t)[0].gml_text() == graph.gml_text()
        finally:
            os.environ.pop("NANOMOTIF_PY_GML", None)


def test_native_scores_are_scipy_exact():
    """Counts in, scores out: a one-node search per (alpha, beta) pair exposes the native evaluation score; compare with
    model.predictive_evaluation_score (scipy.special.psi) bit for bit over a grid that covers the harmonic branch
    (alpha + beta <= 10), the asymptotic branch and large counts."""
    from nanomotif_amd.model import BetaBernoulliModel, predictive_evaluation_score
    rng = np.random.default_rng(12)
    grid = [(0, 0), (1, 0), (0, 1), (5, 0), (3, 2), (10, 0), (100, 3), (1545, 5), (7, 100000), (123456, 789), (9999999, 12345678)]
    grid += [tuple(int(x) for x in rng.integers(0, 10 ** rng.integers(1, 8), size=2)) for _ in range(300)]
    key = ("b", "a")
    bg = np.full((4, 41), 0.25)
    for n_mod, n_nomod in grid:
        score_fn = lambda reqs: np.array([[n_mod, n_nomod]] * len(reqs), dtype=np.int64)
        window_fn = lambda reqs: np.zeros((len(reqs), 2 + 4 * 64), dtype=np.int32)          # no active windows: the search stops at the root
        res = ns.find_best_candidates_custom([(key, 100, bg)], 20, 0.05, 1.5, score_fn, window_fn)
        graph, best, _ = res.result(0, full_graph=True)
        m = BetaBernoulliModel.from_counts(n_mod, n_nomod)
        (root, attrs), = graph.nodes.items()
        assert attrs["score"] == predictive_evaluation_score(m, m), (n_mod, n_nomod)
        assert best == []


def test_sample_many_is_a_run_of_random_sample_calls():
    ns_, ks = [30, 5000, 22, 100000, 64], [5, 50, 6, 1000, 64]
    random.seed(77)
    want = [i for n, k in zip(ns_, ks) for i in random.sample(range(n), k)]
    after = random.random()
    random.seed(77)
    with ps.NativeRandom() as rng:
        got = rng.sample_many(ns_, ks).tolist()
    assert got == want and random.random() == after


def test_sample_groups_equal_sequential_streams():
    """nm_py_random_sample_groups: every group is its own generator stream drawn on some thread; results and the final
    state equal the sequential run."""
    import ctypes as C
    from nanomotif_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    groups, init = [], []
    for g in range(37):
        random.seed(1 if g % 3 else 100 + g)
        init.append(np.array(random.getstate()[1], dtype=np.uint32))
        groups.append([(int(n), int(min(n, k))) for n, k in zip(rng.integers(1, 60000, size=rng.integers(0, 9)), rng.integers(0, 700, size=8))])
    want = []
    for st, calls in zip(init, groups):
        random.setstate((3, tuple(st.tolist()), None))
        for n, k in calls:
            want += random.sample(range(n), k)
    final_want = np.array(random.getstate()[1], dtype=np.uint32)
    off = np.zeros(len(groups) + 1, dtype=np.uint64)
    np.cumsum([len(c) for c in groups], out=off[1:])
    ns_ = np.array([n for c in groups for n, _ in c], dtype=np.uint64)
    ks = np.array([k for c in groups for _, k in c], dtype=np.uint64)
    out = np.zeros(int(ks.sum()), dtype=np.uint32)
    final = np.zeros(625, dtype=np.uint32)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    _lib.check(lib.nm_py_random_sample_groups(len(groups), p(np.ascontiguousarray(np.stack(init)), C.c_uint32), p(off, C.c_uint64), p(ns_, C.c_uint64),
                                              p(ks, C.c_uint64), p(out, C.c_uint32), p(final, C.c_uint32)))
    assert out.tolist() == want and np.array_equal(final, final_want)


def test_searches_advance_on_several_threads_with_identical_results(monkeypatch):
    """72 searches (24 copies of three bins) in one nm_search_run_custom call: the state machines advance on a pool of
    host threads between two batches (NM_SEARCH_THREADS, nmsearch.cpp: Workers).  One thread and eight threads must export
    the same graphs, node for node and bit for bit, and every copy must equal the first of its bin."""
    from oracle.scan import score_candidates
    g4 = load_golden("g4_search.json")
    base, piles, seqs_by_bin, wins = [], {}, {}, {}
    for bin_name, gname in (("binA", "gatc_single"), ("binB", "ecoli_like_m"), ("binC", "ecoli_like_a")):
        g = g4[gname]
        mg = synth.make_metagenome(spec_from_json(g["spec"]))
        mt = g["mod_type"]
        pile, seqs = oracle_bin_inputs(mg, mt)
        base.append((bin_name, mt))
        piles[(bin_name, mt)], seqs_by_bin[bin_name] = pile, seqs
        random.seed(1)
        wins[(bin_name, mt)] = windows_for(mg, mt, pile)
    copies = 24
    keys = [(f"{b}#{c}", mt) for c in range(copies) for b, mt in base]
    origin = {k: (k[0].split("#")[0], k[1]) for k in keys}
    cache = {}

    def run(threads):
        monkeypatch.setenv("NM_SEARCH_THREADS", str(threads))
        store = ps.HostWindowStore()
        for k in keys:
            store.add_task(k, wins[origin[k]][0].copy())

        def score_fn(reqs):
            out = np.zeros((len(reqs), 2), dtype=np.int64)
            for i, (t, m) in enumerate(reqs):
                o = origin[keys[t]]
                ck = (o, m.string, m.mod_position)
                if ck not in cache:
                    cache[ck] = score_candidates(piles[o], seqs_by_bin[o[0]], [(m.string, m.mod_position)])[0]
                out[i] = cache[ck]
            return out
        _, window_fn = _backends(keys, piles, seqs_by_bin, store)
        res = ns.find_best_candidates_custom([(k, store.totals[k], wins[origin[k]][1]) for k in keys], 20, 0.05, 1.5, score_fn, window_fn)
        out = []
        for t in range(len(keys)):
            r = res.result(t, full_graph=True)
            if r is None:
                out.append(None)
                continue
            graph, best, _ = r
            out.append(([(n.string, n.mod_position, graph.nodes[n]["model"].get_raw_counts(), graph.nodes[n]["score"], graph.nodes[n]["priority"],
                          graph.nodes[n]["depth"], graph.nodes[n]["visited"]) for n in graph.nodes],
                        sorted((u.string, v.string) for u, v in graph.edges()), [(m.string, m.mod_position) for m in best]))
        return (res.rounds, res.candidates), out
    one = run(1)
    eight = run(8)
    assert one == eight
    for t, k in enumerate(keys):
        assert one[1][t] == one[1][t % len(base)], k
    assert any(x is not None and len(x[2]) > 0 for x in one[1])


@pytest.mark.parametrize("pad", [31, 64, 95])
def test_native_search_on_wide_frames_equals_python_coroutines(pad):
    """Search frames of 63, 129 and 191 columns (the packed motif keys take 3, 7 and 9 words, the window rows 64 / 192 /
    192 columns): the native state machine against the Python coroutines, graphs and scores with ==."""
    g = load_golden("g4_search.json")["ecoli_like_a"]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    mt = g["mod_type"]
    pile, seqs = oracle_bin_inputs(mg, mt)
    key = ("bin0", mt)
    random.seed(1)
    wins = windows_for(mg, mt, pile, 0.7, pad)
    store = ps.HostWindowStore()
    store.add_task(key, wins[0].copy())
    scorer = LockstepScorer(oracle_backend({key: pile}, {"bin0": seqs}))
    want = ps.run_lockstep({key: ps.find_best_candidates_co(wins[1], mt, pad, min_kl=0.05, score_threshold=1.5)}, scorer, store.execute)[key]
    store2 = ps.HostWindowStore()
    store2.add_task(key, wins[0].copy())
    from oracle.scan import score_candidates
    W = 2 * pad + 1
    ws = (W + 63) // 64 * 64

    def score_fn(reqs):
        return np.array([score_candidates(pile, seqs, [(m.string, m.mod_position)])[0] for _, m in reqs], dtype=np.int64).reshape(-1, 2)

    def window_fn(reqs):
        res = store2.execute([(key, ps.WinReq(kind, m)) for _, kind, m in reqs])
        out = np.zeros((len(reqs), 2 + 4 * ws), dtype=np.int32)
        for i, ((_, kind, m), r) in enumerate(zip(reqs, res)):
            if kind == "remove":
                out[i, 0], out[i, 1] = r
            else:
                out[i, 0] = r[0]
                if r[1] is not None:
                    out[i, 2:].reshape(4, ws)[:, :r[1].shape[1]] = r[1]
        return out
    res = ns.find_best_candidates_custom([(key, store2.totals[key], wins[1])], pad, 0.05, 1.5, score_fn, window_fn)
    got = res.result(0, full_graph=True)
    assert (got is None) == (want is None)
    if want is not None:
        graph, best, _ = got
        wg, wbest, _ = want
        assert list(graph.nodes) == list(wg.nodes) and best == wbest and len(graph.nodes) > 3
        for n in graph.nodes:
            a, b = graph.nodes[n], wg.nodes[n]
            assert a["model"].get_raw_counts() == b["model"].get_raw_counts() and a["score"] == b["score"] and a["priority"] == b["priority"]


@pytest.mark.parametrize("seed", [1, 3, 6, 13])
def test_random_bins_native_search_and_postprocessing_equal_the_coroutines(seed):
    """Seeds of tools/search_fuzz.py (68 seeds there: 0 mismatches): random small bins with random planted motifs, thresholds
    and task mixes — graphs node for node, float64 scores bit for bit, all five post-processing stage tables."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import search_fuzz
    assert "tasks" in search_fuzz.one(seed)


def _g11_cases():
    return sorted(load_golden("g11_random_search.json"), key=lambda k: int(k.split("_")[1]))


@pytest.mark.parametrize("name", _g11_cases())
def test_native_search_reproduces_random_reference_traces(name):
    """g11: find_best_candidates of the reference on random bins (random planted motifs, rates, min_kl, score threshold, seed);
    the native lock-step machine must produce every node in order, the edges and the best candidates."""
    g = load_golden("g11_random_search.json")[name]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    mt = g["mod_type"]
    pile, seqs = oracle_bin_inputs(mg, mt)
    P = g["params"]
    random.seed(P["seed"])
    windows = windows_for(mg, mt, pile, P["high"], P["padding"])
    key = ("bin0", mt)
    store = ps.HostWindowStore()
    store.add_task(key, windows[0])
    score_fn, window_fn = _backends([key], {key: pile}, {"bin0": seqs}, store)
    res = ns.find_best_candidates_custom([(key, store.totals[key], windows[1])], P["padding"], P["min_kl"], P["score_threshold"], score_fn, window_fn)
    graph, best, _ = res.result(0, full_graph=True)
    assert [(n.string, n.mod_position) for n in graph.nodes] == [(r["motif"], r["pos"]) for r in g["nodes"]]
    for (n, d), r in zip(graph.nodes.items(), g["nodes"]):
        assert list(d["model"].get_raw_counts()) == r["counts"]
        assert d["score"] == pytest.approx(r["score"], abs=1e-9, rel=1e-9)
        assert d["priority"] == pytest.approx(r["priority"], abs=1e-12, rel=1e-12)
        assert d["depth"] == r["depth"] and d["visited"] == r["visited"]
    assert sorted((u.string, v.string) for u, v in graph.edges()) == sorted(map(tuple, g["edges"]))
    assert sorted((m.string, m.mod_position) for m in best) == sorted(map(tuple, g["best"]))

