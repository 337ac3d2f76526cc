"""The literal known-answer values of the reference's own unit tests for the Motif algebra (tests/test_candidate.py:42-460
there; inputs and expected outputs only, as tables), applied to BOTH restatements: the oracle's ``oracle.motif`` and the
product's ``nanomotif_amd.motif``.  The recorded fixtures g5 / g7 cover the same operations on 46 motifs and all their
pairs; these are the cases the reference's authors chose themselves."""
import pytest

import nanomotif_amd.motif as product
import oracle.motif as oracle

IMPLS = [pytest.param(oracle, id="oracle"), pytest.param(product, id="product")]

REVERSE_COMPLEMENT = [(("ATCG", 0), ("CGAT", 3)), (("AT[CG]G", 0), ("C[CG]AT", 3)), (("ATC.G.", 0), (".C.GAT", 5)),
                      (("ATAC.G.", 2), (".C.GTAT", 4))]
STRIPPED = [(("....ATCG", 4), ("ATCG", 0)), (("ATCG....", 0), ("ATCG", 0)), (("....AT..CG", 4), ("AT..CG", 0))]
SUB_MOTIF_OF = [(("ATCG", 2), ("ATCG", 2), False), (("ATCG", 0), ("AT", 0), True), (("A[TCG]CG", 0), ("A.C", 0), True),
                (("ATCG", 2), ("CG", 0), True), (("ATCG", 2), ("CG", 2), False), (("CG", 1), ("ATCG", 2), False)]
SUB_STRING_OF = [(("ATCG", 2), ("ATCG", 0), False), (("AGCG", 2), ("ATCG", 0), False), (("ATCG", 0), ("AT", 0), True),
                 (("ACC", 0), ("A[CG]C", 2), True), (("ATCG", 0), ("AT.G", 2), True), (("AT.G", 0), ("ATCG", 2), False),
                 (("AT[ACG]G", 0), ("ATCG", 0), False), (("T.....ATCG...C", 0), ("ATCG", 2), True),
                 (("ATCG", 2), ("T.....ATCG...C", 0), False), (("[AT]..AT..CG[TA]C..C", 2), ("[TA]T..CG[AT][GC]", 0), True)]
DISTANCE = [(("A[TCG]CG", 0), ("A.C", 0), 2), (("ATCG", 0), ("AT", 0), 2), (("ATCG", 2), ("CG", 0), 2), (("CG", 0), ("ATCG", 2), 2),
            (("C...ATCG", 6), ("CG", 0), 3), (("CG", 0), ("C...ATCG", 6), 3), (("C..CG...G", 3), ("CG", 0), 2),
            (("CG", 0), ("C..CG...G", 3), 2), (("CG", 0), ("C..CG...[GC]", 3), 2)]
MERGE = [(("..A..", 2), ("..C..", 2), ("[AC]", 0)), ((".ATGC.", 1), (".TAAG", 2), ("A[AT]G", 0)), (("AGG", 0), ("VAAG", 1), ("A[AG]G", 0)),
         (("GATC...R", 1), ("A...TATC", 5), ("[GT]ATC", 1))]
IUPAC = [(("A[TCG]CG", 0), "ABCG"), (("A.C", 0), "ANC"), (("ATCG", 0), "ATCG"), (("ATCG", 2), "ATCG"), (("CG", 0), "CG"),
         (("C...ATCG", 6), "CNNNATCG"), (("C..CG...G", 3), "CNNCGNNNG"), (("C..CG...[GC]", 3), "CNNCGNNNS")]
MERGE_NO_STRIP = [(("..A..", 2), ("..C..", 2), "..[AC].."), (("G.A..", 2), ("..C..", 2), "..[AC].."), ((".A..", 1), ("..C..", 2), "..[AC].."),
                  ((".A......", 1), ("..C..", 2), "..[AC]......"), (("G.GATC", 3), ("C.GATC", 3), "[CG].GATC"),
                  (("[CG].GATC", 3), ("G.GATG", 3), "[CG].GAT[CG]")]
ALIGN = [([("G.GATC", 3), ("C.GATC", 3), ("G.GATG", 3)], [("G.GATC", 3), ("C.GATC", 3), ("G.GATG", 3)]),
         ([("..A..", 2), ("..C..", 2)], [("..A..", 2), ("..C..", 2)]),
         ([("G.GATC", 3), ("GAT.", 1)], [("G.GATC", 3), ("..GAT.", 3)])]
_X_GAT_Y = lambda firsts, lasts: {(f + ".GAT" + l, 3) for f in firsts for l in lasts}
# (inputs, merged, pre-variants or their number or None, new variants / their number / (must contain, must not contain))
MERGE_VARIANTS = [
    ([("G.GATC", 3), ("G.GATC", 3)], ("G.GATC", 3), {("G.GATC", 3)}, 0),
    ([("G.GATC", 3), ("C.GATG", 3)], ("[CG].GAT[CG]", 3), 2, {("C.GATC", 3), ("G.GATG", 3)}),
    ([("G.GATC", 3), ("..GAT.", 3)], ("GAT", 1), _X_GAT_Y("ACGT", "ACGT"), 0),
    ([("G.GATC", 3), ("C.GAT.", 3), ("A.GATC", 3), ("T.GAT.", 3)], ("GAT", 1),
     {("A.GATC", 3), ("G.GATC", 3)} | _X_GAT_Y("CT", "ACGT"), _X_GAT_Y("AG", "AGT")),
    ([(".G.GATC", 4), ("C.GAT.", 3), ("A.GATC", 3), ("T.GAT.", 3)], ("GAT", 1),
     {("A.GATC", 3), ("G.GATC", 3)} | _X_GAT_Y("CT", "ACGT"), _X_GAT_Y("AG", "AGT")),
    ([("G.GATC", 3), ("GAT.", 1)], ("GAT", 1), None, 0),
    ([("G.GATC", 3), ("C.GATC", 3), ("G.GATG", 3)], ("[CG].GAT[CG]", 3), None, (("C.GATG", 3), ("G.GATC", 3))),
    ([(".....", 2), (".....", 2)], (".....", 2), {(".....", 2)}, 0),
]


def _pair(m):
    return (str(m), m.mod_position)


@pytest.mark.parametrize("impl", IMPLS)
def test_unary_known_answers(impl):
    M = impl.Motif
    for a, want in REVERSE_COMPLEMENT:
        assert M(*a).reverse_compliment() == M(*want) and _pair(M(*a).reverse_compliment()) == want
    for a, want in STRIPPED:
        assert M(*a).new_stripped_motif() == M(*want) and _pair(M(*a).new_stripped_motif()) == want
    assert _pair(M("....AT..CG..", 4).new_stripped_motif().reverse_compliment()) == ("CG..AT", 5)
    for a, want in IUPAC:
        assert M(*a).iupac() == want


@pytest.mark.parametrize("impl", IMPLS)
def test_binary_known_answers(impl):
    M = impl.Motif
    for a, b, want in SUB_MOTIF_OF:
        assert M(*a).sub_motif_of(M(*b)) == want, (a, b)
    for a, b, want in SUB_STRING_OF:
        assert M(*a).sub_string_of(M(*b)) == want, (a, b)
    for a, b, want in DISTANCE:
        assert M(*a).distance(M(*b)) == want, (a, b)
    for a, b, want in MERGE:
        got = M(*a).merge(M(*b))
        assert got == M(*want) and got.mod_position == want[1], (a, b, _pair(got))
    for a, b, want in MERGE_NO_STRIP:
        assert str(M(*a).merge_no_strip(M(*b))) == want, (a, b)


@pytest.mark.parametrize("impl", IMPLS)
def test_align_and_variant_known_answers(impl):
    M = impl.Motif
    for inp, want in ALIGN:
        got = impl.align_motifs([M(*x) for x in inp])
        assert [_pair(m) for m in got] == want and got == [M(*x) for x in want]
    for inp, merged, pre, new in MERGE_VARIANTS:
        got_merged, got_pre, got_new = impl.merge_and_find_new_variants([M(*x) for x in inp])
        assert got_merged == M(*merged) and str(got_merged) == merged[0], (inp, _pair(got_merged))
        pre_pairs, new_pairs = {_pair(m) for m in got_pre}, {_pair(m) for m in got_new}
        if isinstance(pre, int):
            assert len(pre_pairs) == pre
        elif pre is not None:
            assert pre_pairs == pre, inp
        if isinstance(new, int):
            assert len(new_pairs) == new, inp
        elif isinstance(new, set):
            assert (new_pairs == new) if len(new) > 2 else (new <= new_pairs and len(new_pairs) == 2), inp
        else:
            assert new[0] in new_pairs and new[1] not in new_pairs, inp


def test_motif_rows_carry_the_reference_columns(tmp_path):
    """tests/test_motif.py there: derived columns (n_mod 50 / n_nomod 100 of a (5, 5) prior updated with (50, 100); motif_iupac,
    mod_position_iupac), the model survives pickling (alpha 55), and write_motifs writes REQUIRED_COLUMNS + DERIVED_COLUMNS
    (motif.py:656-670) in that order without the object column, complementary columns (:672-681) behind them when present."""
    import pickle
    from nanomotif_amd import postprocess as pp
    from nanomotif_amd.model import BetaBernoulliModel
    model = BetaBernoulliModel(5, 5)
    model.update(50, 100)
    row = pp.MotifRow("contig1", "GATC", "m6A", 2, model, 1.23)
    assert (row.n_mod, row.n_nomod, row.motif_iupac, row.mod_position_iupac) == (50, 100, "GATC", 2)
    back = pickle.loads(pickle.dumps(row))
    assert back.model._alpha == 55 and back.model._beta == 105 and (back.n_mod, back.n_nomod) == (50, 100)
    wide = pp.MotifRow("Enterococcus_faecalis_complete_genome", "C[AG]AA......[AG]TTG", "a", 3, BetaBernoulliModel.from_counts(15, 25), 16.22159133144396)
    assert (wide.motif_iupac, wide.mod_position_iupac) == ("CRAANNNNNNRTTG", 3)
    out = tmp_path / "motifs.tsv"
    pp.write_motifs([row], str(out))
    lines = out.read_text().strip().split("\n")
    assert lines[0].split("\t") == ["reference", "motif", "mod_type", "mod_position", "score", "n_mod", "n_nomod", "motif_iupac", "mod_position_iupac"]
    assert lines[1].split("\t") == ["contig1", "GATC", "m6A", "2", "1.23", "50", "100", "GATC", "2"]
    comp = BetaBernoulliModel(5, 5)
    comp.update(30, 70)
    paired = pp.MotifRow("contig1", "GATC", "m6A", 2, model, 1.23, complement=pp.MotifRow("contig1", "CTAG", "m6A", 3, comp, 0.98),
                         has_complement_columns=True)
    pp.write_motifs([paired], str(out))
    lines = out.read_text().strip().split("\n")
    assert lines[0].split("\t")[9:] == ["motif_complement", "mod_position_complement", "score_complement", "n_mod_complement",
                                        "n_nomod_complement", "motif_iupac_complement", "mod_position_iupac_complement"]
    assert lines[1].split("\t")[9:] == ["CTAG", "3", "0.98", "30", "70", "CTAG", "3"]
