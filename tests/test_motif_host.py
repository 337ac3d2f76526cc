"""Product-side host types (nanomotif_amd.motif / .model) against the reference-recorded vectors."""
import numpy as np
import pytest

from helpers import load_golden
from nanomotif_amd import motif as pm
from nanomotif_amd.motif import Motif


def test_motif_algebra_matches_reference_vectors():
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
        assert pm.motif_type(st.iupac()) == u["motif_type_of_iupac"]
        sets, pos = m.stripped_sets()
        assert len(sets) == len(st.split()) and pos == st.mod_position
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
        assert pm.iupac_to_regex(c["iupac"]) == c["regex"] and pm.regex_to_iupac(c["regex"]) == c["roundtrip"]
        assert pm.motif_type(c["iupac"]) == c["type"]
    for c in g["align"]:
        al = pm.align_motifs([Motif(s, p) for s, p in c["in"]])
        assert [[m.string, m.mod_position] for m in al] == c["out"]
    for c in g["merge_variants"]:
        merged, pre, new = pm.merge_and_find_new_variants([Motif(s, p) for s, p in c["in"]])
        assert [merged.string, merged.mod_position] == c["merged"]
        assert sorted([m.string, m.mod_position] for m in pre) == c["pre"]
        assert sorted([m.string, m.mod_position] for m in new) == c["new"]


def test_merge_motifs_matches_reference_vectors():
    g = load_golden("g7_parents_merge.json")
    for c in g["merge"]:
        res = pm.merge_motifs([Motif(s, p) for s, p in c["in"]])
        got = sorted(({"merged": [m.string, m.mod_position],
                       "cluster": sorted([x.string, x.mod_position] for x in cl),
                       "pre": sorted([x.string, x.mod_position] for x in pre),
                       "new": sorted([x.string, x.mod_position] for x in new)} for m, cl, pre, new in res),
                     key=lambda r: r["merged"])
        assert got == c["out"]


def test_reverse_complement_of_sets_matches_string_form():
    for s, p in [("GATC", 1), ("G[AG].GAAG[CT]", 5), ("CC[AT]GG", 1), ("[ACG]A[CGT]", 1)]:
        m = Motif(s, p)
        sets, _ = m.stripped_sets()
        rc_sets, _ = m.reverse_compliment().stripped_sets()
        assert [pm.complement_set(int(x)) for x in sets[::-1]] == rc_sets.tolist()
