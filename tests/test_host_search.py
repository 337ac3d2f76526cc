"""Host-side search / post-processing of the product package, driven on CPU by the oracle's scan as the scoring
backend: must reproduce the reference-recorded search traces (tests/golden/g4_search.json) node for node."""
import random

import numpy as np
import pytest

from helpers import load_golden, oracle_bin_inputs, spec_from_json
from nanomotif_amd import postprocess as ppp
from nanomotif_amd import search as ps
from nanomotif_amd import synth
from nanomotif_amd.find_motifs_bin import LockstepScorer
from nanomotif_amd.motif import Motif


def oracle_backend(piles, seqs):
    """local_counts(flat) -> int64[n,2] computed by the CPU oracle; piles: {(bin, mt): pileup dict}."""
    from oracle.scan import score_candidates

    def local(flat):
        out = np.zeros((len(flat), 2), dtype=np.int64)
        for i, (key, m, tag) in enumerate(flat):
            out[i] = score_candidates(piles[key], seqs[key[0]], [(m.string, m.mod_position)])[0]
        return out
    return local


def windows_for(mg, mod_type, pile, high=0.7, pad=20):
    contigs = {mg.names[i]: mg.contig_ascii(i) for i in range(len(mg.names))}
    plus = {n: p.position[(p.fraction_mod >= high) & (p.strand == ord("+"))] for n, p in pile.items() if len(p)}
    minus = {n: p.position[(p.fraction_mod >= high) & (p.strand == ord("-"))] for n, p in pile.items() if len(p)}
    return ps.extract_windows(contigs, plus, minus, mod_type, pad)


@pytest.mark.parametrize("name", ["gatc_single", "ecoli_like_m", "geobacillus_like", "no_motif"])
def test_product_search_reproduces_reference_trace(name):
    g = load_golden("g4_search.json")[name]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    mt = g["mod_type"]
    pile, seqs = oracle_bin_inputs(mg, mt)
    P = g["params"]
    random.seed(P["seed"])
    windows = windows_for(mg, mt, pile, P["high"], P["padding"])
    key = ("bin0", mt)
    scorer = LockstepScorer(oracle_backend({key: pile}, {"bin0": seqs}))
    store = ps.HostWindowStore()
    store.add_task(key, windows[0])
    co = ps.find_best_candidates_co(windows[1], mt, P["padding"], min_kl=P["min_kl"], max_dead_ends=25,
                                    max_rounds_since_new_best=30, score_threshold=P["score_threshold"])
    graph, best, bin_pssm = ps.run_lockstep({key: co}, scorer, store.execute)[key]
    assert np.allclose(bin_pssm, np.array(g["bin_pssm_4dp"]), atol=5.1e-5, rtol=0)
    assert [(n.string, n.mod_position) for n in graph.nodes] == [(r["motif"], r["pos"]) for r in g["nodes"]]
    for (n, d), r in zip(graph.nodes.items(), g["nodes"]):
        assert list(d["model"].get_raw_counts()) == r["counts"]
        assert d["score"] == pytest.approx(r["score"], abs=1e-9, rel=1e-9)
        assert d["priority"] == pytest.approx(r["priority"], abs=1e-12, rel=1e-12)
        assert d["depth"] == r["depth"] and d["visited"] == r["visited"]
    assert sorted((u.string, v.string) for u, v in graph.edges()) == sorted(map(tuple, g["edges"]))
    assert sorted((m.string, m.mod_position) for m in best) == sorted(map(tuple, g["best"]))
    # far fewer launches than candidates: children and pruning parents are batched per round
    assert scorer.rounds < scorer.candidates or scorer.candidates <= 1


def test_lockstep_equals_sequential_and_matches_oracle_chain():
    """Two bins advanced together give what each gives alone; the post-processing chain agrees with the oracle's."""
    from oracle import postprocess as opp
    from oracle import search as ose
    g4 = load_golden("g4_search.json")
    tasks, piles, seqs_by_bin, expect = {}, {}, {}, {}
    store = ps.HostWindowStore()
    for bin_name, gname in (("binA", "geobacillus_like"), ("binB", "ecoli_like_m")):
        g = g4[gname]
        mg = synth.make_metagenome(spec_from_json(g["spec"]))
        mt = g["mod_type"]
        pile, seqs = oracle_bin_inputs(mg, mt)
        key = (bin_name, mt)
        piles[key], seqs_by_bin[bin_name] = pile, seqs
        random.seed(1)
        windows = windows_for(mg, mt, pile)

        store.add_task(key, windows[0])

        def chain(bin_name=bin_name, mt=mt, windows=windows):
            graph, best, _ = yield from ps.find_best_candidates_co(windows[1], mt, 20, min_kl=0.05, score_threshold=1.5)
            rows = yield from ppp.postprocess_co(graph, best, bin_name, mt, 20)
            return rows
        tasks[key] = chain()
        random.seed(1)
        og, ob, _ = ose.find_best_candidates(pile, seqs, mt, 0.3, 0.7, 20, min_kl=0.05, score_threshold=1.5)
        expect[key] = opp.format_bin_motifs(opp.process_bin(pile, seqs, bin_name, mt, og, ob, 20))
    scorer = LockstepScorer(oracle_backend(piles, seqs_by_bin))
    res = ps.run_lockstep(tasks, scorer, store.execute)
    for key in tasks:
        rows = [r for r in res[key] if r.n_mod + r.n_nomod >= 50]
        assert ppp.format_bin_motifs(rows) == expect[key]
    assert "GATC" in expect[("binA", "a")] and "CCWGG" in expect[("binB", "m")]


def test_postprocess_kats():
    from nanomotif_amd.model import BetaBernoulliModel
    mk = lambda motifs, pos, mod="m": [ppp.MotifRow("ref1", m, mod, p, BetaBernoulliModel(), 1.0) for m, p in zip(motifs, pos)]
    r = ppp.join_motif_complements(mk(["AAGGTT", "AACCTT"], [0, 0]))          # tests/test_postprocess.py:17-36
    assert [x.motif for x in r] == ["AAGGTT"] and [x.complement.motif for x in r] == ["AACCTT"]
    r = ppp.join_motif_complements(mk(["GCGC", "GCGC"], [1, 3]))
    assert len(r) == 4
    motifs = ["AATT", "GATC", "CCA......TGCC", "CAGACG..G", "GGCA......TGG", "GGGAGC", "TTAA", "CTCGAG", "GCAGATG"]
    r = ppp.join_motif_complements(mk(motifs, [1, 1, 2, 3, 3, 3, 3, 4, 4], "a"))  # tests/test_postprocess.py:95-146
    assert [x.motif for x in r] == ["AATT", "GATC", "CAGACG..G", "GGCA......TGG", "GGGAGC", "TTAA", "CTCGAG", "GCAGATG"]
    assert [x.complement.motif if x.complement else None for x in r] == \
        ["AATT", "GATC", None, "CCA......TGCC", None, "TTAA", "CTCGAG", None]
    assert len(ppp.remove_noisy_motifs(mk(["AAGGTT", "AACCTT", "GATCC"], [0, 0, 0]))) == 3
    m = [Motif("ACGT", 0), Motif("ACG", 0), Motif("CGT", 1), Motif("ACGTG", 0), Motif("TGCA", 1)]
    assert set(ppp.get_motif_parental_relationship(m)) == {(m[1], m[0]), (m[1], m[3]), (m[2], m[0]), (m[2], m[3]), (m[0], m[3])}
    text = ppp.format_bin_motifs(mk(["GATC"], [1], "a"))
    assert text.split("\n")[0].split("\t") == ppp.HEADER and text.split("\n")[1].split("\t")[:7] == \
        ["ref1", "GATC", "1", "a", "0", "0", "palindrome"]


def test_native_random_sample_is_cpython_exact():
    """nm_py_random_sample == random.sample(range(n), k), and the interpreter's generator ends in the same state."""
    for seed in (1, 2403, 99):
        for n, k in [(10, 3), (21, 5), (22, 6), (50, 50), (100, 6), (85, 21), (86, 21), (5000, 50), (300_000, 3000),
                     (2 ** 20 + 5, 77), (64, 64), (1, 1), (5, 0)]:
            random.seed(seed)
            want = random.sample(range(n), k)
            after = random.random()
            random.seed(seed)
            got = ps._native_random_sample(n, k).tolist()
            assert got == want and random.random() == after, (seed, n, k)


def test_letter_counts_matches_numpy():
    rng = np.random.default_rng(2)
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=100_000, p=[0.3, 0.2, 0.2, 0.29, 0.01])
    starts = rng.integers(0, 100_000 - 41, size=30_000)
    win = seq[starts[:, None] + np.arange(41)[None, :]]
    want = np.array([(win == ord(b)).sum(axis=0) for b in "ATGC"])
    assert np.array_equal(ps.letter_counts(seq, starts, 41), want)
    assert np.array_equal(ps.letter_pssm(win), want / len(starts))


def test_kl_columns_equals_scipy_entropy():
    from scipy.stats import entropy
    rng = np.random.default_rng(8)
    for _ in range(50):
        p = rng.random((4, 41))
        q = rng.random((4, 41))
        p[:, rng.integers(41)] = [1, 0, 0, 0]
        q[rng.integers(4), rng.integers(41)] = 0.0          # inf where p > 0
        a, b = ps.kl_divergence_columns(p, q), entropy(p, q)
        assert np.array_equal(a, b, equal_nan=True)


def test_present_sorted_ids_equals_present_per_task():
    """The native plan's vectorised presence table (FilteredPileup.present_sorted_ids) against ``present`` task by task: contigs
    unknown to the pileup, empty bins, mod types nobody has."""
    from nanomotif_amd.find_motifs_bin import FilteredPileup
    rng = np.random.default_rng(3)
    names = [f"contig_{i:03d}" for i in rng.permutation(60)]
    kept = (rng.random((60, 8)) < 0.4).astype(np.uint32) * rng.integers(1, 9, (60, 8)).astype(np.uint32)
    kept[:, 2] = 0                                            # a mod type without rows
    z = np.zeros(0)
    f = FilteredPileup(names, z, z, z, z, kept)
    all_names = names + ["not_in_the_pileup_1", "not_in_the_pileup_2"]
    order = rng.permutation(len(all_names))
    bins = {}
    for k, i in enumerate(order.tolist()):
        bins.setdefault(f"bin_{k % 7}", []).append(all_names[i])
    bins["bin_of_strangers"] = ["not_in_the_pileup_3"]
    id_of = {n: 1000 + i for i, n in enumerate(sorted(set(all_names) | {"not_in_the_pileup_3"}))}
    got = f.present_sorted_ids(bins, id_of, 3)
    n_tasks = 0
    for b, members in bins.items():
        for m in range(3):
            want = [id_of[x] for x in sorted(f.present(members, m))]
            if want:
                n_tasks += 1
                assert got[(b, m)].tolist() == want and got[(b, m)].dtype == np.uint32
            else:
                assert (b, m) not in got
    assert n_tasks == len(got) and n_tasks > 10 and ("bin_of_strangers", 0) not in got
