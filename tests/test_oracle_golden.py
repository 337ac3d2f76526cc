"""Pin the CPU oracle against vectors recorded from the real reference (tests/golden/gen_golden.py)
and against the literal known-answer values of the reference's own tests."""
import os
import random

import numpy as np
import pytest

from helpers import load_golden, oracle_bin_inputs, sha1, spec_from_json
from nanomotif_amd import synth
from oracle import motif as om
from oracle import pileup as op
from oracle import postprocess as opp
from oracle import scan as osc
from oracle import search as ose
from oracle.model import BetaBernoulliModel, predictive_evaluation_score
from oracle.motif import Motif

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------ G1: subseq_indices
def _g1_seqs(g):
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    seqs = [mg.contig_str(i) for i in range(3)]
    s2 = list(seqs[2])
    for k, ch in g["iupac_edits"]:
        s2[k] = ch
    seqs[2] = "".join(s2)
    import hashlib
    assert [hashlib.sha1(s.encode()).hexdigest() for s in seqs] == g["seq_sha1"], "synthetic generator drifted"
    return seqs


def test_g1_subseq_indices():
    g = load_golden("g1_subseq_indices.json")
    assert osc.subseq_indices("AATT", g["kat"]["seq"]).tolist() == g["kat"]["AATT"]      # tests/test_fasta.py:95-109
    assert osc.subseq_indices("AA.T", g["kat"]["seq"]).tolist() == g["kat"]["AA.T"]
    seqs = _g1_seqs(g)
    for c in g["cases"]:
        idx = osc.subseq_indices(c["motif"], seqs[c["contig"]])
        assert len(idx) == c["n"] and sha1(idx.astype(np.int64)) == c["sha1"], c["motif"]


# ------------------------------------------------------------------ G2: motif_model_contig
def test_reference_kat_methylated_motif_occurrences():
    # tests/test_motif_find.py:14-39
    m = Motif("ACG", 0)
    a, b = osc.methylated_motif_occourances(m, "TACGGACGCCACG", np.array([1, 5]), np.array([10]))
    assert a.tolist() == [1, 5] and b.tolist() == [10]
    a, b = osc.methylated_motif_occourances(m, "TACGGACGCCACG", np.array([]), np.array([1, 10]))
    assert a.tolist() == [] and b.tolist() == [1, 10]


def test_g2_motif_model_contig():
    g = load_golden("g2_motif_model_contig.json")
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    seq = mg.contig_str(0)
    import hashlib
    assert hashlib.sha1(seq.encode()).hexdigest() == g["seq_sha1"]
    piles = {}
    for mt in ("a", "m"):
        p = mg.contig_pileup(0, mt)
        frac = synth.pct_to_fraction(p["pct_hundredths"])
        assert sha1(frac) == g["pileup_sha1"][mt]["fraction_mod"] and sha1(p["position"]) == g["pileup_sha1"][mt]["position"]
        piles[mt] = osc.ContigPileup(p["position"], p["strand"], frac)
    for c in g["cases"]:
        model, d = osc.motif_model_contig(piles[c["mod_type"]], seq, BetaBernoulliModel(), Motif(c["motif"], c["pos"]),
                                          c["low"], c["high"], save_motif_positions=True)
        assert list(model.get_raw_counts()) == [c["n_mod"], c["n_nomod"]], c
        for k in ("index_meth_fwd", "index_nonmeth_fwd", "index_meth_rev", "index_nonmeth_rev"):
            assert len(d[k]) == c[k]["n"] and sha1(d[k].astype(np.int64)) == c[k]["sha1"], (c["motif"], k)


# ------------------------------------------------------------------ G3: scores
def test_g3_scores():
    g = load_golden("g3_scores.json")
    m = BetaBernoulliModel()
    m.update(*g["kat"]["update"])
    assert m.mean() == g["kat"]["mean"]
    assert abs(m.posterior_predictive_per_obs(m._alpha, m._beta) - g["kat"]["ppo_self"]) < 1e-12
    for c in g["cases"]:
        nxt, cur = BetaBernoulliModel(), BetaBernoulliModel()
        nxt.update(*c["next"])
        cur.update(*c["cur"])
        s = predictive_evaluation_score(nxt, cur)
        assert s == pytest.approx(c["score"], rel=1e-12, abs=1e-12)
        assert nxt.mean() == c["mean_next"]


# ------------------------------------------------------------------ G5: motif algebra
def test_g5_motif_algebra():
    g = load_golden("g5_motif_algebra.json")
    for u in g["unary"]:
        m = Motif(u["motif"], u["pos"])
        st = m.new_stripped_motif()
        rc = st.reverse_compliment()
        assert m.split() == u["split"] and m.length() == u["length"] and m.trimmed_length() == u["trimmed_length"]
        assert [st.string, st.mod_position] == u["stripped"]
        assert [rc.string, rc.mod_position] == u["revcomp_of_stripped"]
        assert m.one_hot().tolist() == u["one_hot"] and st.iupac() == u["iupac"]
        for k, (h, n) in u["isolated"].items():
            assert m.have_isolated_bases(isolation_size=int(k)) == h and m.count_isolated_bases(isolation_size=int(k)) == n
        assert om.motif_type(st.iupac()) == u["motif_type_of_iupac"]
    for b in g["binary"]:
        x, y = Motif(*b["a"]), Motif(*b["b"])
        assert x.sub_motif_of(y) == b["sub_motif_of"], b
        assert x.sub_string_of(y) == b["sub_string_of"], b
        assert x.distance(y) == b["distance"], b
        assert (x == y) == b["eq"]
        if b["merge"] is not None:
            mg = x.merge(y)
            assert [mg.string, mg.mod_position] == b["merge"], b
        if b["merge_no_strip"] is not None:
            mg = x.merge_no_strip(y)
            assert [mg.string, mg.mod_position] == b["merge_no_strip"], b
    for c in g["iupac"]:
        assert om.iupac_to_regex(c["iupac"]) == c["regex"] and om.regex_to_iupac(c["regex"]) == c["roundtrip"]
        assert om.motif_type(c["iupac"]) == c["type"]
    for c in g["align"]:
        al = om.align_motifs([Motif(s, p) for s, p in c["in"]])
        assert [[m.string, m.mod_position] for m in al] == c["out"]
    for c in g["merge_variants"]:
        merged, pre, new = om.merge_and_find_new_variants([Motif(s, p) for s, p in c["in"]])
        assert [merged.string, merged.mod_position] == c["merged"]
        assert sorted([m.string, m.mod_position] for m in pre) == c["pre"]
        assert sorted([m.string, m.mod_position] for m in new) == c["new"]


def test_reference_kat_motif():
    # literal values of tests/test_candidate.py:42-64 and tests/test_motif_find.py:42-84
    assert Motif("ATCG", 0).reverse_compliment() == Motif("CGAT", 3)
    assert Motif("AT[CG]G", 0).reverse_compliment() == Motif("C[CG]AT", 3)
    assert Motif("ATAC.G.", 2).reverse_compliment() == Motif(".C.GTAT", 4)
    assert Motif("....AT..CG..", 4).new_stripped_motif().reverse_compliment() == Motif("CG..AT", 5)
    m = [Motif("ACGT", 0), Motif("ACG", 0), Motif("CGT", 1), Motif("ACGTG", 0), Motif("TGCA", 1)]
    rel = set(opp.get_motif_parental_relationship(m))
    assert rel == {(m[1], m[0]), (m[1], m[3]), (m[2], m[0]), (m[2], m[3]), (m[0], m[3])}
    m = [Motif("GGCA[AT]", 2), Motif("GGCAAT", 2), Motif("GGCAAT", 4), Motif("AATTT", 0), Motif("AATTT", 1), Motif("AATTTT", 0)]
    rel = set(opp.get_motif_parental_relationship(m))
    assert rel == {(m[0], m[1]), (m[0], m[2]), (m[3], m[5]), (m[4], m[5])}
    m = [Motif("TTAAGGAG", 6), Motif("TTAA", 3)]
    assert set(opp.get_motif_parental_relationship(m)) == {(m[1], m[0])}


# ------------------------------------------------------------------ G6: background sampling
def test_g6_background_windows():
    import hashlib
    g = load_golden("g6_background.json")
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    random.seed(1)
    for c in g["cases"]:
        s = mg.contig_str(c["contig"])
        w = ose.sample_n_subsequences_unique(s, 41, c["n"], c["base"])
        assert hashlib.sha1("".join(w).encode()).hexdigest() == c["windows_sha1"] and w[:2] == c["first"]
        assert np.array_equal(ose.letter_pssm(w), np.array(c["pssm"]))


# ------------------------------------------------------------------ G4: search traces
@pytest.mark.parametrize("name", ["gatc_single", "ecoli_like_a", "ecoli_like_m", "geobacillus_like", "no_motif"])
def test_g4_search_trace(name):
    g = load_golden("g4_search.json")[name]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    pile, seqs = oracle_bin_inputs(mg, g["mod_type"])
    assert sum(len(p) for p in pile.values()) == g["n_rows"]
    P = g["params"]
    random.seed(P["seed"])
    res = ose.find_best_candidates(pile, seqs, g["mod_type"], P["low"], P["high"], P["padding"], min_kl=P["min_kl"],
                                   max_dead_ends=25, max_rounds_since_new_best=30, score_threshold=P["score_threshold"])
    assert res is not None
    graph, best, bin_pssm = res
    assert np.allclose(bin_pssm, np.array(g["bin_pssm_4dp"]), atol=5.1e-5, rtol=0)  # fixture is the %.4f text dump
    got_nodes = {(n.string, n.mod_position): d for n, d in graph.nodes.items()}
    assert len(got_nodes) == len(g["nodes"])
    assert [(n.string, n.mod_position) for n in graph.nodes] == [(r["motif"], r["pos"]) for r in g["nodes"]]
    for r in g["nodes"]:
        d = got_nodes[(r["motif"], r["pos"])]
        assert list(d["model"].get_raw_counts()) == r["counts"], r["motif"]
        assert d["score"] == pytest.approx(r["score"], abs=1e-9, rel=1e-9), r["motif"]
        assert d["priority"] == pytest.approx(r["priority"], abs=1e-12, rel=1e-12)
        assert d["depth"] == r["depth"] and d["visited"] == r["visited"]
    assert sorted((u.string, v.string) for u, v in graph.edges()) == sorted(map(tuple, g["edges"]))
    assert sorted((m.string, m.mod_position) for m in best) == sorted(map(tuple, g["best"]))
    assert [(m.string, m.mod_position) for m in best][:1] == [tuple(x) for x in g["best"]][:1]


# ------------------------------------------------------------------ G7: parents + merge
def test_g7_parents_and_merge():
    g = load_golden("g7_parents_merge.json")
    gs = load_golden("g4_search.json")[g["bin"]]
    mg = synth.make_metagenome(spec_from_json(gs["spec"]))
    pile, seqs = oracle_bin_inputs(mg, gs["mod_type"])
    for c in g["parents"]:
        ps = ose.get_parent_scores(Motif(*c["motif"]), pile, seqs, 0.3, 0.7)
        assert [k.string for k in ps] == [p["motif"] for p in c["parents"]]
        for (k, v), p in zip(ps.items(), c["parents"]):
            assert v["motif_position"] == p["motif_position"]
            assert list(v["parent_model"].get_raw_counts()) == p["parent_counts"]
            assert list(v["child_model"].get_raw_counts()) == p["child_counts"]
            assert v["score"] == pytest.approx(p["score"], abs=1e-9, rel=1e-9)
    for c in g["merge"]:
        res = om.merge_motifs([Motif(s, p) for s, p in c["in"]])
        got = sorted(({"merged": [m.string, m.mod_position],
                       "cluster": sorted([x.string, x.mod_position] for x in cl),
                       "pre": sorted([x.string, x.mod_position] for x in pre),
                       "new": sorted([x.string, x.mod_position] for x in new)} for m, cl, pre, new in res),
                     key=lambda r: r["merged"])
        assert got == c["out"]


# ------------------------------------------------------------------ pileup filters (reference KATs)
def test_adjacency_filter_kat():
    # tests/test_dataload.py:37-69
    t = dict(contig=np.array(["contig1"] * 10, dtype=object), position=np.arange(10, dtype=np.int64),
             mod_type=np.array(["m6A"] * 10, dtype=object), strand=np.array(["+"] * 10, dtype=object),
             fraction_mod=np.array([0.8, 0.9, 0.1, 0.95, 0.85, 0.2, 0.75, 0.9, 0.05, 0.8]),
             Nvalid_cov=np.full(10, 10))
    f = op.filter_pileup_adjacency_filter(t, methylation_threshold=0.7, adjacency_distance=1)
    assert f["position"].tolist() == [1, 2, 3, 5, 7, 8, 9]
    t = dict(contig=np.array(["contig1"] * 5 + ["contig2"] * 5, dtype=object),
             position=np.array(list(range(5)) + list(range(5)), dtype=np.int64),
             mod_type=np.array(["m6A", "5mC"] * 5, dtype=object), strand=np.array(["+"] * 5 + ["-"] * 5, dtype=object),
             fraction_mod=np.array([0.8, 0.9, 0.1, 0.95, 0.85, 0.2, 0.75, 0.9, 0.05, 0.8]), Nvalid_cov=np.full(10, 10))
    f = op.filter_pileup_adjacency_filter(t, methylation_threshold=0.7, adjacency_distance=1)
    assert f["position"][f["contig"] == "contig1"].tolist() == [1, 2, 3]
    assert f["position"][f["contig"] == "contig2"].tolist() == [0, 2, 3, 4]


def test_frequency_and_coverage_filters():
    n = 200_000
    t = dict(contig=np.array(["c1"] * n + ["c2"] * 100, dtype=object), position=np.arange(n + 100, dtype=np.int64),
             mod_type=np.array(["a"] * (n + 100), dtype=object), strand=np.array(["+"] * (n + 100), dtype=object),
             fraction_mod=np.zeros(n + 100), Nvalid_cov=np.full(n + 100, 6))
    t["fraction_mod"][:51] = 0.71       # c1: 51 > 50 and 51/200000 > 1e-4
    t["fraction_mod"][n:n + 50] = 0.9    # c2: 50 is not > 50
    t["Nvalid_cov"][60] = 5             # strict > 5
    f = op.filter_pileup_minimummod_frequency(op.filter_pileup(t))
    assert set(f["contig"].tolist()) == {"c1"} and len(f["position"]) == n - 1
    t["fraction_mod"][50] = 0.7          # not > 0.7 -> only 50 modified rows left
    assert len(op.filter_pileup_minimummod_frequency(t)["position"]) == 0


# ------------------------------------------------------------------ postprocess (reference KATs)
def _rows(motifs, pos, mod="m", models=None, scores=None):
    models = models or [BetaBernoulliModel() for _ in motifs]
    scores = scores or [1.0] * len(motifs)
    return [opp.derive(dict(reference="ref1", motif=m, mod_type=mod, mod_position=p, model=mo, score=s))
            for m, p, mo, s in zip(motifs, pos, models, scores)]


def test_join_motif_complements_kats():
    # tests/test_postprocess.py:17-146
    r = opp.join_motif_complements(_rows(["AAGGTT", "AACCTT"], [0, 0]))
    assert [x["motif"] for x in r] == ["AAGGTT"] and [x["motif_complement"] for x in r] == ["AACCTT"]
    r = opp.join_motif_complements(_rows(["GATCC", "GATCG"], [0, 0]))
    assert {x["motif"] for x in r} == {"GATCC", "GATCG"} and [x["motif_complement"] for x in r] == [None, None]
    r = opp.join_motif_complements(_rows(["GCGC", "GATC"], [1, 2]))
    assert {x["motif"] for x in r} == {"GCGC", "GATC"} and {x["motif_complement"] for x in r} == {"GCGC", "GATC"}
    r = opp.join_motif_complements(_rows(["GCGC", "GCGC"], [1, 3]))
    assert [x["motif"] for x in r] == ["GCGC"] * 4 and [x["motif_complement"] for x in r] == ["GCGC"] * 4
    motifs = ["AATT", "GATC", "CCA......TGCC", "CAGACG..G", "GGCA......TGG", "GGGAGC", "TTAA", "CTCGAG", "GCAGATG"]
    r = opp.join_motif_complements(_rows(motifs, [1, 1, 2, 3, 3, 3, 3, 4, 4], mod="a"))
    assert [x["motif"] for x in r] == ["AATT", "GATC", "CAGACG..G", "GGCA......TGG", "GGGAGC", "TTAA", "CTCGAG", "GCAGATG"]
    assert [x["motif_complement"] for x in r] == ["AATT", "GATC", None, "CCA......TGCC", None, "TTAA", "CTCGAG", None]


def test_remove_noisy_motifs_kat():
    rows = _rows(["AAGGTT", "AACCTT", "GATCC"], [0, 0, 0])
    assert len(opp.remove_noisy_motifs(rows)) == 3         # tests/test_postprocess.py:147-162
    rows = _rows(["A....T", "GATC"], [0, 1])
    assert [r["motif"] for r in opp.remove_noisy_motifs(rows)] == ["GATC"]
    rows = _rows(["A....T"], [0])
    assert len(opp.remove_noisy_motifs(rows)) == 1         # all noisy -> unchanged (postprocess.py:21-22)


def test_full_chain_runs_and_formats():
    g = load_golden("g4_search.json")["geobacillus_like"]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    pile, seqs = oracle_bin_inputs(mg, "a")
    random.seed(1)
    graph, best, _ = ose.find_best_candidates(pile, seqs, "a", 0.3, 0.7, 20, min_kl=0.05, score_threshold=1.5)
    rows = opp.process_bin(pile, seqs, "bin0", "a", graph, best, 20)
    text = opp.format_bin_motifs(rows)
    lines = text.strip().split("\n")
    assert lines[0].split("\t") == opp.HEADER
    found = {l.split("\t")[1] for l in lines[1:]}
    assert {"GATC", "ACCCA", "CCAAAT"} <= found and any(m.startswith("G") and m.endswith("GAAGY") for m in found), found


def _iupac_to_regex(iupac):
    sets = {"R": "[AG]", "Y": "[CT]", "S": "[CG]", "W": "[AT]", "K": "[GT]", "M": "[AC]", "B": "[CGT]", "D": "[AGT]", "H": "[ACT]",
            "V": "[ACG]", "N": "."}
    return "".join(sets.get(ch, ch) for ch in iupac)


def test_bin_motifs_text_equals_a_reference_output_file():
    """tests/golden/ref_e_coli_bin-motifs.tsv is nanomotif/datasets/e_coli_bin-motifs.tsv — a bin-motifs.tsv the reference
    wrote (polars write_csv of motif.py:899-926's frame; data, copied as is).  Feeding its own rows — primary and complement
    as separate search results — through join_motif_complements and the formatter must give the file back byte for byte:
    pins the header, the column order, the sort key, the null formatting, motif_type and which partner of a
    complementary pair is kept as the primary.  Both the oracle's chain and the product's."""
    from nanomotif_amd import postprocess as npp
    from nanomotif_amd.model import BetaBernoulliModel as ProductModel
    want = open(os.path.join(GOLDEN, "ref_e_coli_bin-motifs.tsv")).read()
    if not want.endswith("\n"):
        want += "\n"                                       # the data set file was saved without the final newline
    cells = [l.split("\t") for l in want.strip("\n").split("\n")[1:]]
    found = []                                             # (reference, iupac, pos, mod type, n_mod, n_nomod) as the search finds them
    for c in cells:
        found.append((c[0], c[1], int(c[2]), c[3], int(c[4]), int(c[5])))
        if c[7] and c[7] != c[1]:
            found.append((c[0], c[7], int(c[8]), c[3], int(c[9]), int(c[10])))
    for order in (found, found[::-1]):
        rows = [opp.derive(dict(reference=r, motif=_iupac_to_regex(m), mod_type=mt, mod_position=p, model=_model(a, b), score=1.0))
                for r, m, p, mt, a, b in order]
        assert opp.format_bin_motifs(opp.join_motif_complements(rows)) == want
        prows = [npp.MotifRow(r, _iupac_to_regex(m), mt, p, ProductModel.from_counts(a, b), 1.0) for r, m, p, mt, a, b in order]
        assert npp.format_bin_motifs(npp.join_motif_complements(prows)) == want
    # motif_type on the motifs of the reference's other shipped output (older header, same classification)
    for line in open(os.path.join(GOLDEN, "ref_geobacillus-plasmids.bin-motifs.tsv")).read().strip().split("\n")[1:]:
        c = line.split("\t")
        from nanomotif_amd.motif import motif_type as product_motif_type
        assert opp.motif_type(c[2]) == c[6] and product_motif_type(c[2]) == c[6]


def _model(a, b):
    m = BetaBernoulliModel()
    m.update(a, b)
    return m


def test_read_methylation_restatement_on_a_hand_worked_case():
    """oracle/contig_methylation.read_methylation (epimetheus' methylation_pattern restated; parity unpinned — there is no
    reference vector): a case small enough to work by hand.  GATC@1 on GATCAGATCTGATC: '+' sites 1, 6, 11 (the A), '-'
    sites 2, 7, 12 (the T opposite the A of the reverse-complement GATC)."""
    from oracle.contig_methylation import read_methylation
    seq = "GATCAGATCTGATC"
    rec = {("c", "a"): dict(position=np.array([1, 6, 11, 2, 7, 12, 4]), strand=np.frombuffer(b"+++---+", np.uint8).copy(),
                            n_valid=np.array([10, 20, 2, 8, 10, 40, 50]), n_mod=np.array([9, 10, 2, 8, 1, 30, 50]),
                            n_diff=np.array([0, 0, 0, 0, 3, 0, 0]))}
    # kept: pos 1 (0.9, cov 10), 6 (0.5, cov 20), 2 (1.0, cov 8), 12 (0.75, cov 40); dropped: 11 (coverage 2 < 3), 7 (10 / 13 < 0.8);
    # position 4 is no GATC site
    rows = read_methylation(rec, {"c": seq}, [("GATC", "a", 1)], 3, 0.8, "median")
    assert rows == [dict(contig="c", motif=0, n_motif_obs=4, mean_read_cov=(10 + 20 + 8 + 40) / 4, methylation_value=(0.75 + 0.9) / 2)]
    rows = read_methylation(rec, {"c": seq}, [("GATC", "a", 1)], 3, 0.8, "weighted-mean")
    assert rows[0]["methylation_value"] == (9 + 10 + 8 + 30) / (10 + 20 + 8 + 40)
    assert read_methylation(rec, {"c": seq}, [("GATC", "m", 3)], 3, 0.8) == []        # no record of that mod code
    rows = read_methylation(rec, {"c": seq}, [("GATC", "a", 1)], 3, 0.0, "median")    # odd count: the middle value
    assert rows[0]["n_motif_obs"] == 5 and rows[0]["methylation_value"] == 0.75


# ------------------------------------------------------------------ G8 - G10: post-processing glue and pre-filters, recorded from the
# reference's OWN functions run on the row-list frame of tests/golden/refframe.py (round 4)
def post_table(rows):
    """Oracle / product rows -> the sorted plain table g8 / g9 record (tests/golden/gen_golden.py: _table)."""
    comp = any("motif_complement" in r for r in rows)
    out = []
    for r in rows:
        row = [r["reference"], r["motif"], r["mod_type"], int(r["mod_position"]), int(r["n_mod"]), int(r["n_nomod"]),
               float(r["score"]), r["motif_iupac"], int(r["mod_position_iupac"])]
        if comp:
            none = r.get("motif_complement") is None
            row += [None if none else r["motif_complement"], None if none else int(r["mod_position_complement"]),
                    None if none else int(r["n_mod_complement"]), None if none else int(r["n_nomod_complement"]),
                    None if none else r["motif_iupac_complement"], None if none else int(r["mod_position_iupac_complement"])]
        out.append(row)
    return sorted(out, key=lambda x: [("" if v is None else str(v)) for v in x])


def assert_tables_equal(got, exp, what):
    assert len(got) == len(exp), (what, got, exp)
    for g, e in zip(got, exp):
        assert len(g) == len(e), (what, g, e)
        for i, (a, b) in enumerate(zip(g, e)):
            if i == 6:                                        # the score: float64, same operation order
                assert a == pytest.approx(b, rel=0, abs=1e-9), (what, g, e)
            else:
                assert a == b, (what, g, e)


def g8_case_inputs(g, case):
    """(oracle pileups per mod type, sequences, input rows) of one g8 case."""
    mg = synth.make_metagenome(spec_from_json(g["bins"][case["bin"]]))
    piles, seqs = {}, None
    for mt in case["mod_types"]:
        piles[mt], seqs = oracle_bin_inputs(mg, mt)
    rows = []
    for motif, pos, mt, counts, score in case["input"]:
        model = BetaBernoulliModel()
        model.update(*counts)
        rows.append(opp.derive(dict(reference="bin0", motif=motif, mod_type=mt, mod_position=pos, model=model, score=score)))
    return piles, seqs, rows


def oracle_post_stages(rows, piles, seqs):
    """The chain of find_motifs_bin.py:555-593 on a frame that may hold several (reference, mod_type) groups."""
    stages = {}
    rows = opp.remove_noisy_motifs(rows)
    stages["noise"] = rows
    groups = {}
    for r in rows:
        groups.setdefault((r["reference"], r["mod_type"]), []).append(r)
    merged = []
    for (_, mt), grp in groups.items():
        merged += opp.merge_motifs_in_rows(grp, piles[mt], seqs)
    rows = opp.unique_rows(merged)
    stages["merge"] = rows
    groups = {}
    for r in rows:
        groups.setdefault((r["reference"], r["mod_type"]), []).append(r)
    kept = []
    for grp in groups.values():
        kept += opp.remove_sub_motifs(grp)
    rows = opp.unique_rows(kept)
    stages["sub"] = rows
    stages["complement"] = opp.join_motif_complements(rows)
    return stages


def test_g8_postprocess_glue_equals_the_reference_functions():
    g = load_golden("g8_postprocess_glue.json")
    assert len(g["cases"]) >= 12
    fired = set()
    for case in g["cases"]:
        assert not case["a_stage_held_one_motif_twice"]      # no recorded value rests on polars' Object-cell equality
        piles, seqs, rows = g8_case_inputs(g, case)
        # the input counts themselves were computed by the reference's motif_model_bin
        for r, (motif, pos, mt, counts, _) in zip(rows, case["input"]):
            m = osc.motif_model_bin(piles[mt], seqs, Motif(motif, pos), BetaBernoulliModel(), 0.3, 0.7)
            assert list(m.get_raw_counts()) == counts, (case["name"], motif)
        stages = oracle_post_stages(rows, piles, seqs)
        for name in ("noise", "merge", "sub", "complement"):
            assert_tables_equal(post_table(stages[name]), case["stages"][name], (case["name"], name))
        n_in, n_noise, n_merge, n_sub = len(case["input"]), len(case["stages"]["noise"]), len(case["stages"]["merge"]), len(case["stages"]["sub"])
        if n_noise < n_in:
            fired.add("noise")
        if any("[" in r[1] for r in case["stages"]["merge"]):
            fired.add("degenerate merge")
        if n_merge == n_noise and "rejected" in case["name"]:
            fired.add("rejected merge")
        if n_sub < n_merge:
            fired.add("sub-motif")
        if len(case["stages"]["complement"]) != n_sub:
            fired.add("complement join changes the row count")
    assert fired == {"noise", "degenerate merge", "rejected merge", "sub-motif", "complement join changes the row count"}


def test_g13_duplicate_merged_motifs_follow_the_reference():
    """g13 (round 5): families in which two merge clusters produce the SAME merged motif.  merge_motifs_in_df gives each accepted
    cluster its own row and model (find_motifs_bin.py:1497-1504); `motifs.unique()` (:570, 579, 588) compares the Object `model`
    cells by identity and keeps both; remove_sub_motifs drops every row of a discarded motif; join_motif_complements pairs every
    row with every partner row, so a duplicated palindrome comes out four times (seed 625: 3 rows -> 5)."""
    g = load_golden("g13_duplicate_merged_motifs.json")
    assert len(g["cases"]) >= 4
    grew = 0
    for case in g["cases"]:
        assert case["a_stage_held_one_motif_twice"]
        keys = [(r[0], r[2], r[1], r[3]) for r in case["stages"]["merge"]]
        assert len(set(keys)) < len(keys)
        piles, seqs, rows = g8_case_inputs(g, case)
        stages = oracle_post_stages(rows, piles, seqs)
        for name in ("noise", "merge", "sub", "complement"):
            assert_tables_equal(post_table(stages[name]), case["stages"][name], (case["name"], name))
        grew += len(case["stages"]["complement"]) > len(case["stages"]["sub"])
    assert grew >= 2


def _g11_cases():
    return sorted(load_golden("g11_random_search.json"), key=lambda k: int(k.split("_")[1]))


@pytest.mark.parametrize("name", _g11_cases())
def test_g11_random_search_traces(name):
    """find_best_candidates of the reference on RANDOM bins (tests/golden/search_ref_fuzz.py cases: random planted motifs,
    methylation rates, min_kl, score threshold, random seed) against the oracle's search: nodes in order, edges, best."""
    g = load_golden("g11_random_search.json")[name]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    pile, seqs = oracle_bin_inputs(mg, g["mod_type"])
    P = g["params"]
    random.seed(P["seed"])
    res = ose.find_best_candidates(pile, seqs, g["mod_type"], P["low"], P["high"], P["padding"], min_kl=P["min_kl"],
                                   max_dead_ends=25, max_rounds_since_new_best=30, score_threshold=P["score_threshold"])
    assert res is not None
    graph, best, _ = res
    assert [(n.string, n.mod_position) for n in graph.nodes] == [(r["motif"], r["pos"]) for r in g["nodes"]]
    for (n, d), r in zip(graph.nodes.items(), g["nodes"]):
        assert list(d["model"].get_raw_counts()) == r["counts"], r["motif"]
        assert d["score"] == pytest.approx(r["score"], abs=1e-9, rel=1e-9), r["motif"]
        assert d["priority"] == pytest.approx(r["priority"], abs=1e-12, rel=1e-12)
        assert d["depth"] == r["depth"] and d["visited"] == r["visited"]
    assert sorted((u.string, v.string) for u, v in graph.edges()) == sorted(map(tuple, g["edges"]))
    assert sorted((m.string, m.mod_position) for m in best) == sorted(map(tuple, g["best"]))


def test_g12_random_process_subpileup_stage_tables():
    """process_subpileup of the reference on 20 RANDOM bins (tests/golden/subpileup_ref_fuzz.py cases) against the oracle's
    search + post-processing chain: all five stage tables and the return value."""
    g = load_golden("g12_random_process_subpileup.json")
    assert len(g) >= 20
    for name, rec in g.items():
        mg = synth.make_metagenome(spec_from_json(rec["spec"]))
        pile, seqs = oracle_bin_inputs(mg, rec["mod_type"])
        p = rec["params"]
        random.seed(p["seed"])
        res = ose.find_best_candidates(pile, seqs, rec["mod_type"], p["low"], p["high"], p["padding"], min_kl=p["min_kl"],
                                       score_threshold=p["score_threshold"])
        rows = opp.graph_to_rows(res[0], res[1], "bin0", rec["mod_type"], p["padding"])
        assert_tables_equal(post_table(rows), rec["stages"]["motifs"], (name, "motifs"))
        stages = oracle_post_stages(rows, {rec["mod_type"]: pile}, seqs)
        for ours, theirs in (("noise", "motifs-noise"), ("merge", "motifs-noise-merge"), ("sub", "motifs-noise-merge-sub"),
                             ("complement", "motifs-noise-merge-sub-complement")):
            assert_tables_equal(post_table(stages[ours]), rec["stages"][theirs], (name, theirs))
        final = opp.process_bin(pile, seqs, "bin0", rec["mod_type"], res[0], res[1], p["padding"])
        assert_tables_equal(post_table(final or []), rec["final"] or [], (name, "final"))


def test_g9_process_subpileup_stage_tables():
    """process_subpileup itself (find_motifs_bin.py:468-596) run from the reference on six bins: every stage table it writes
    and its return value against the oracle's search + post-processing chain."""
    g = load_golden("g9_process_subpileup.json")
    for name, rec in g.items():
        mg = synth.make_metagenome(spec_from_json(rec["spec"]))
        pile, seqs = oracle_bin_inputs(mg, rec["mod_type"])
        p = rec["params"]
        random.seed(p["seed"])
        res = ose.find_best_candidates(pile, seqs, rec["mod_type"], p["low"], p["high"], p["padding"], min_kl=p["min_kl"],
                                       score_threshold=p["score_threshold"])
        if rec["final"] is None and not rec["stages"]:
            assert res is None or not opp.graph_to_rows(res[0], res[1], "bin0", rec["mod_type"], p["padding"]), name
            continue
        rows = opp.graph_to_rows(res[0], res[1], "bin0", rec["mod_type"], p["padding"])
        assert_tables_equal(post_table(rows), rec["stages"]["motifs"], (name, "motifs"))
        stages = oracle_post_stages(rows, {rec["mod_type"]: pile}, seqs)
        for ours, theirs in (("noise", "motifs-noise"), ("merge", "motifs-noise-merge"), ("sub", "motifs-noise-merge-sub"),
                             ("complement", "motifs-noise-merge-sub-complement")):
            assert_tables_equal(post_table(stages[ours]), rec["stages"][theirs], (name, theirs))
        final = opp.process_bin(pile, seqs, "bin0", rec["mod_type"], res[0], res[1], p["padding"])
        assert_tables_equal(post_table(final or []), rec["final"] or [], (name, "final"))


def g10_table(g):
    """The input table of g10, rebuilt from its seed (tests/golden/gen_golden.py: g10) as an oracle.pileup table."""
    rng = np.random.default_rng(g["seed"])
    cols = {"contig": [], "mod_type": [], "fraction_mod": [], "Nvalid_cov": [], "position": []}
    for contig, mt, n, n_mod, _ in g["groups"][:-1]:
        frac = rng.integers(0, 7000, n).astype(np.float64) / 10000.0
        frac[rng.choice(n, n_mod, replace=False)] = rng.integers(7001, 10001, n_mod) / 10000.0
        if contig == "c_ok" and mt == "a":
            frac[np.flatnonzero(frac <= 0.7)[:5]] = 0.7
        if contig.startswith("c_null"):
            frac[np.flatnonzero(frac <= 0.7)[:100 if contig == "c_null" else 100_002]] = np.nan
        cols["contig"] += [contig] * n
        cols["mod_type"] += [mt] * n
        cols["fraction_mod"].append(frac)
        cols["Nvalid_cov"].append(rng.integers(6, 60, n))
        cols["position"].append(np.arange(n, dtype=np.int64))
    n = 10_000
    frac = rng.integers(0, 7000, n).astype(np.float64) / 10000.0
    hot = rng.choice(n, 55, replace=False)
    frac[hot] = 0.9
    cov = rng.integers(6, 60, n)
    cov[hot[:5]] = 5
    cov[hot[5:8]] = 6
    cols["contig"] += ["c_cov"] * n
    cols["mod_type"] += ["a"] * n
    cols["fraction_mod"].append(frac)
    cols["Nvalid_cov"].append(cov)
    cols["position"].append(np.arange(n, dtype=np.int64))
    t = dict(contig=np.array(cols["contig"], dtype=object), mod_type=np.array(cols["mod_type"], dtype=object),
             fraction_mod=np.concatenate(cols["fraction_mod"]), Nvalid_cov=np.concatenate(cols["Nvalid_cov"]).astype(np.int64),
             position=np.concatenate(cols["position"]))
    t["strand"] = np.full(len(t["position"]), ord("+"), dtype=np.uint8)
    assert len(t["position"]) == g["n_rows"]
    for k in ("fraction_mod", "Nvalid_cov", "position"):
        assert sha1(t[k]) == g["input_sha1"][k], f"g10 input column {k} drifted"
    t["row"] = np.arange(len(t["position"]), dtype=np.int64)
    return t


def test_g10_coverage_and_frequency_filters():
    """dataload.filter_pileup + filter_pileup_minimummod_frequency (dataload.py:191-226) run from the reference: the strict
    bounds (> 5, > 50, > 1e-4, fraction > 0.7) and a null percentage counting as a position of its group."""
    g = load_golden("g10_frequency_filter.json")
    t = g10_table(g)
    a = op.filter_pileup(t)
    assert len(a["row"]) == g["after_coverage"] and sha1(a["row"]) == g["after_coverage_rows_sha1"]
    b = op.filter_pileup_minimummod_frequency(a)
    assert len(b["row"]) == g["after_frequency"] and sha1(b["row"]) == g["after_frequency_rows_sha1"]
    kept = {}
    for c, m in zip(b["contig"].tolist(), b["mod_type"].tolist()):
        kept[f"{c}|{m}"] = kept.get(f"{c}|{m}", 0) + 1
    assert kept == g["kept_groups"]
    assert "c_null_tip|m" not in kept and "c_ratio_eq|a" not in kept and "c_ok|m" not in kept and "c_cov|a" not in kept


def test_g14_adjacency_filter_run_from_the_reference():
    """dataload.filter_pileup_adjacency_filter (dataload.py:228-247) EXECUTED by tests/golden/gen_golden.py at the production
    distance 8 (and 1, 3) on gapped positions, tied fractions, nulls and 'm' / '21839' rows sharing positions: the oracle's
    restatement and the product's host filter keep exactly the recorded rows."""
    from helpers import g14_table
    from nanomotif_amd import pileup as pp
    g = load_golden("g14_adjacency_filter.json")
    t = g14_table(g)
    assert g["counts"]["nulls"] > 100 and g["counts"]["nulls_kept"] == 0 and g["counts"]["confident_dropped"] > 1000
    names = sorted(set(t["contig"].tolist()))
    codes = {"m": 0, "a": 1, "21839": 2}
    table = pp.PileupTable(names, np.array([names.index(c) for c in t["contig"].tolist()], dtype=np.int32), t["position"],
                           np.array([codes[m] for m in t["mod_type"].tolist()], dtype=np.int8),
                           np.array([ord(s) for s in t["strand"].tolist()], dtype=np.uint8), t["fraction_mod"], t["Nvalid_cov"])
    for d, kept in g["kept_rows"].items():
        got = op.filter_pileup_adjacency_filter(t, methylation_threshold=g["methylation_threshold"], adjacency_distance=int(d))
        assert sorted(got["row"].tolist()) == kept, d
        # the product's host filter returns a table, not row numbers: compare the rows themselves
        host = pp.filter_pileup_adjacency_filter(table, methylation_threshold=g["methylation_threshold"], adjacency_distance=int(d))
        key = lambda c, p, s, m, f: sorted(zip(c, p, s, m, [None if np.isnan(x) else x for x in f]))
        k = np.array(kept, dtype=np.int64)
        assert key(host.contig.tolist(), host.position.tolist(), host.strand.tolist(), host.mod_type.tolist(), host.fraction_mod.tolist()) == \
            key(table.contig[k].tolist(), table.position[k].tolist(), table.strand[k].tolist(), table.mod_type[k].tolist(), table.fraction_mod[k].tolist()), d
    # all three filters in the pipeline's order leave the same rows (every group passes the first two: asserted at recording time)
    assert sorted(op.prefilter(t)["row"].tolist()) == g["kept_rows"]["8"]
