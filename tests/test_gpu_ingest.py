"""Device-side pre-filters + classification (nm_ingest_pileup) against the CPU oracle's filters
(oracle/pileup.py, pinned by the reference's adjacency known-answer cases)."""
import os

import numpy as np
import pytest

from nanomotif_amd import synth
from nanomotif_amd.motif import Motif

pytestmark = pytest.mark.gpu


def _raw_table(mg, rng=None):
    cols = []
    for code, mt in ((0, "m"), (1, "a")):
        if mt not in mg.spec.mod_types:
            continue
        c = mg.pileup_columns(mt)
        c["mod"] = np.full(len(c["position"]), code, np.int8)
        cols.append(c)
    cat = lambda k: np.concatenate([c[k] for c in cols])
    t = dict(contig=cat("contig_id").astype(np.int64), position=cat("position"), strand=cat("strand"), mod_type=cat("mod"),
             fraction_mod=cat("fraction_mod").copy(), Nvalid_cov=cat("nvalid").astype(np.int64))
    if rng is not None:      # adversarial edits: ties inside windows, exact thresholds, coverage edge, a third mod code
        n = len(t["position"])
        idx = rng.choice(n, size=n // 50, replace=False)
        t["fraction_mod"][idx] = rng.choice([0.7, 0.7000000000000001, 0.6999999999999999, 0.9, 0.9, 1.0, 0.3], size=len(idx))
        idx = rng.choice(n, size=n // 200, replace=False)
        t["Nvalid_cov"][idx] = rng.choice([5, 6, 0], size=len(idx))
        idx = rng.choice(n, size=n // 300, replace=False)
        t["mod_type"][idx] = 3                                       # e.g. 'h': takes part in the filters only
    return t


def test_device_filters_match_oracle_filters():
    from nanomotif_amd.engine import ScanEngine
    from oracle import pileup as op
    from oracle.scan import ContigPileup, score_candidates
    spec = synth.SynthSpec(n_contigs=6, total_bp=600_000, n_bins=2, mod_types=("a", "m"), seed=81, min_contig_bp=30_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m"), ("GAATTC", 2, "a")))
    mg = synth.make_metagenome(spec)
    t = _raw_table(mg, np.random.default_rng(4))
    # contig 5 gets too few methylated rows for mod 'm' -> dropped by the frequency filter
    sel = (t["contig"] == 5) & (t["mod_type"] == 0)
    t["fraction_mod"][sel] = np.minimum(t["fraction_mod"][sel], 0.5)
    exp = op.prefilter({k: v.copy() for k, v in t.items()})
    eng = ScanEngine(0)
    eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(6)], mg.bin_names)
    res = eng.ingest_pileup(t["contig"].astype(np.uint32), t["position"], t["mod_type"], t["strand"], t["fraction_mod"],
                            t["Nvalid_cov"], {0: ("m", "C"), 1: ("a", "A")})
    assert res["n_kept"] == len(exp["position"])
    kept = np.zeros((6, 8), dtype=np.int64)
    np.add.at(kept, (exp["contig"], exp["mod_type"]), 1)
    assert np.array_equal(res["kept"].astype(np.int64), kept) and kept[5, 0] == 0
    # confident rows (fraction >= 0.7) of the scored mod codes
    conf = (exp["fraction_mod"] >= 0.7) & (exp["mod_type"] < 2)
    want = sorted(zip(exp["contig"][conf].tolist(), exp["position"][conf].tolist(), exp["strand"][conf].tolist(), exp["mod_type"][conf].tolist()))
    cc, cp, cs, cm = res["confident"]
    assert sorted(zip(cc.tolist(), cp.tolist(), cs.tolist(), cm.tolist())) == want
    # scoring on the device-filtered planes == oracle scoring on the oracle-filtered rows
    motifs = [("GATC", 1), ("A", 0), ("GAATTC", 2), ("CC[AT]GG", 1), ("C", 0), ("G[AG].GAAG[CT]", 5)]
    for code, mt in ((0, "m"), (1, "a")):
        for b in sorted(set(mg.bin_names)):
            idx = [i for i, x in enumerate(mg.bin_names) if x == b]
            pile = {}
            for i in idx:
                s = (exp["contig"] == i) & (exp["mod_type"] == code)
                pile[mg.names[i]] = ContigPileup(exp["position"][s], exp["strand"][s], exp["fraction_mod"][s])
            seqs = {mg.names[i]: mg.contig_str(i) for i in idx}
            these = [(s, p) for s, p in motifs if Motif(s, p).split()[p] in ("A" if mt == "a" else "C", "[AG]")]
            expc = score_candidates(pile, seqs, these)
            got = eng.score([(Motif(s, p), mt, b) for s, p in these])
            assert np.array_equal(got, expc), (mt, b)
    # a shard: rows of absent contigs are ignored
    eng2 = ScanEngine(0)
    mine = [0, 2, 4]
    eng2.upload_assembly([mg.names[i] for i in mine], [mg.contig_ascii(i) for i in mine], [mg.bin_names[i] for i in mine],
                         bin_names=sorted(set(mg.bin_names)))
    lut = np.full(6, 0xFFFFFFFF, dtype=np.uint32)
    lut[mine] = np.arange(3, dtype=np.uint32)
    res2 = eng2.ingest_pileup(lut[t["contig"]], t["position"], t["mod_type"], t["strand"], t["fraction_mod"], t["Nvalid_cov"],
                              {0: ("m", "C"), 1: ("a", "A")})
    assert np.array_equal(res2["kept"], res["kept"][mine])
    eng.close(); eng2.close()


def test_adjacency_kat_on_device():
    """tests/test_dataload.py:37-69 shapes, scaled to the fixed d = 8 of the pipeline."""
    from nanomotif_amd.engine import ScanEngine
    from oracle import pileup as op
    rng = np.random.default_rng(9)
    L = 4000
    seq = "".join(rng.choice(list("ACGT"), size=L))
    pos = np.arange(0, L, 3, dtype=np.int64)
    frac = rng.choice([0.0, 0.2, 0.69, 0.7, 0.75, 0.8, 0.8, 0.95, 1.0], size=len(pos))
    t = dict(contig=np.zeros(len(pos), np.int64), position=pos, strand=np.where(rng.random(len(pos)) < 0.5, ord("+"), ord("-")).astype(np.uint8),
             mod_type=rng.choice([0, 1], size=len(pos)).astype(np.int8), fraction_mod=frac, Nvalid_cov=np.full(len(pos), 20))
    exp = op.prefilter({k: v.copy() for k, v in t.items()})
    eng = ScanEngine(0)
    eng.upload_assembly(["c"], [seq], ["b"])
    res = eng.ingest_pileup(t["contig"].astype(np.uint32), pos, t["mod_type"], t["strand"], frac, t["Nvalid_cov"], {0: ("m", "C"), 1: ("a", "A")})
    assert res["n_kept"] == len(exp["position"]) and 0 < res["n_kept"] < len(pos)
    eng.close()


def test_duplicate_rows_are_refused():
    """Two surviving rows on one (contig, position, strand) of a mod type: the reference's np.isin(assume_unique=True)
    would miscount silently (find_motifs_bin.py:1258); the engine refuses the pileup."""
    from nanomotif_amd._lib import NmScanError
    from nanomotif_amd.engine import ScanEngine
    rng = np.random.default_rng(4)
    L = 3000
    seq = "".join(rng.choice(list("ACGT"), size=L))
    pos = np.arange(0, L, 2, dtype=np.int64)
    frac = np.where(rng.random(len(pos)) < 0.3, 0.95, 0.05)
    cols = dict(contig=np.zeros(len(pos), np.uint32), position=pos, mod=np.ones(len(pos), np.int8),
                strand=np.full(len(pos), ord("+"), np.uint8), frac=frac, nvalid=np.full(len(pos), 20))
    eng = ScanEngine(0)
    eng.upload_assembly(["c"], [seq], ["b"])
    ok = eng.ingest_pileup(cols["contig"], cols["position"], cols["mod"], cols["strand"], cols["frac"], cols["nvalid"], {1: ("a", "A")})
    assert ok["n_kept"] > 0
    k = int(np.flatnonzero(frac < 0.3)[5])                        # an unmethylated row: never touched by the adjacency filter
    dup = {n: np.concatenate([v, v[k:k + 1]]) for n, v in cols.items()}
    with pytest.raises(NmScanError, match="duplicate"):
        eng.ingest_pileup(dup["contig"], dup["position"], dup["mod"], dup["strand"], dup["frac"], dup["nvalid"], {1: ("a", "A")})
    eng.close()


def test_ingest_in_parts_equals_one_piece():
    """nm_ingest_pileup_part: the pileup cut at contig boundaries gives the same planes, tables and confident rows."""
    from nanomotif_amd.engine import ScanEngine
    spec = synth.SynthSpec(n_contigs=9, total_bp=700_000, n_bins=3, mod_types=("a", "m"), seed=83, min_contig_bp=30_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    t = _raw_table(mg, np.random.default_rng(6))
    order = np.argsort(t["contig"], kind="stable")                  # rows grouped by contig, mod types interleaved
    t = {k: v[order] for k, v in t.items()}
    labels = {0: ("m", "C"), 1: ("a", "A")}
    motifs = [(Motif("GATC", 1), "a"), (Motif("A", 0), "a"), (Motif("CC[AT]GG", 1), "m"), (Motif("C", 0), "m")]
    cands = [(m, mt, b) for b in sorted(set(mg.bin_names)) for m, mt in motifs]
    out = []
    for max_rows in (None, 50_000):
        eng = ScanEngine(0)
        eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(len(mg.names))], mg.bin_names)
        cid = t["contig"].astype(np.uint32)
        if max_rows:
            assert len(ScanEngine._pileup_parts(cid, max_rows)) >= 4
        res = eng.ingest_pileup(cid, t["position"], t["mod_type"], t["strand"], t["fraction_mod"], t["Nvalid_cov"], labels,
                                max_part_rows=max_rows)
        rows = sorted(zip(*[x.tolist() for x in res["confident"]]))
        out.append((res["n_kept"], res["n_confident"], res["kept"].tolist(), rows, eng.score(cands).tolist(),
                    eng.methylated_row_counts("a", 20).tolist()))
        eng.close()
    assert out[0] == out[1] and out[0][0] > 0 and out[0][1] > 0
    # rows of one contig in two places: no parts possible
    assert ScanEngine._pileup_parts(np.array([0, 0, 1, 1, 0], np.uint32), 2) is None


def test_5mc_and_4mc_rows_share_positions_and_the_adjacency_filter():
    """'m' (5mC) and '21839' (4mC) rows sit on the SAME cytosines; the adjacency filter groups by (contig, strand) with
    the mod types mixed (dataload.py:236-245), so a strong 4mC call can remove a weaker 5mC neighbour and vice versa.
    Device filters + per-mod-type classification against the oracle filters, then scoring on both classifications."""
    from nanomotif_amd.engine import ScanEngine
    from oracle import pileup as op
    from oracle.scan import ContigPileup, score_candidates
    spec = synth.SynthSpec(n_contigs=4, total_bp=500_000, n_bins=2, mod_types=("m", "21839"), seed=83, min_contig_bp=60_000,
                           fixed_motifs=(("CCWGG", 1, "m"), ("GGCC", 2, "21839"), ("CACAG", 1, "21839")))
    mg = synth.make_metagenome(spec)
    cols = []
    for code, mt in ((0, "m"), (2, "21839")):
        c = mg.pileup_columns(mt)
        c["mod"] = np.full(len(c["position"]), code, np.int8)
        cols.append(c)
    cat = lambda k: np.concatenate([c[k] for c in cols])
    t = dict(contig=cat("contig_id").astype(np.int64), position=cat("position"), strand=cat("strand"), mod_type=cat("mod"),
             fraction_mod=cat("fraction_mod").copy(), Nvalid_cov=cat("nvalid").astype(np.int64))
    # modkit writes a contig's rows together, ordered by position: both mod codes of one cytosine are neighbours
    order = np.lexsort((t["mod_type"], t["position"], t["contig"]))
    t = {k: v[order] for k, v in t.items()}
    both = (t["position"][1:] == t["position"][:-1]) & (t["contig"][1:] == t["contig"][:-1]) & (t["strand"][1:] == t["strand"][:-1])
    assert both.sum() > 100_000                                     # the two mod types really share their rows' positions
    exp = op.prefilter({k: v.copy() for k, v in t.items()})
    solo = sum(len(op.prefilter({k: v[t["mod_type"] == code].copy() for k, v in t.items()})["position"]) for code in (0, 2))
    assert len(exp["position"]) < solo                              # mixing the mod types removed rows a per-type filter keeps
    eng = ScanEngine(0)
    eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(4)], mg.bin_names)
    res = eng.ingest_pileup(t["contig"].astype(np.uint32), t["position"], t["mod_type"], t["strand"], t["fraction_mod"],
                            t["Nvalid_cov"], {0: ("m", "C"), 2: ("21839", "C")})
    assert res["n_kept"] == len(exp["position"])
    kept = np.zeros((4, 8), dtype=np.int64)
    np.add.at(kept, (exp["contig"], exp["mod_type"]), 1)
    assert np.array_equal(res["kept"].astype(np.int64), kept) and (kept[:, 0] > 0).all() and (kept[:, 2] > 0).all()
    motifs = [("CC[AT]GG", 1), ("GGCC", 2), ("CACAG", 1), ("C", 0), ("GC", 1), ("[AG]C[CT]", 1)]
    for code, mt in ((0, "m"), (2, "21839")):
        for b in sorted(set(mg.bin_names)):
            idx = [i for i, x in enumerate(mg.bin_names) if x == b]
            pile = {}
            for i in idx:
                s = (exp["contig"] == i) & (exp["mod_type"] == code)
                pile[mg.names[i]] = ContigPileup(exp["position"][s], exp["strand"][s], exp["fraction_mod"][s])
            seqs = {mg.names[i]: mg.contig_str(i) for i in idx}
            expc = score_candidates(pile, seqs, motifs)
            got = eng.score([(Motif(s, p), mt, b) for s, p in motifs])
            assert np.array_equal(got, expc), (mt, b)
            assert got[0].sum() > 0
    eng.close()


def test_many_mod_codes_and_null_percentages_follow_the_reference_filters():
    """More than eight distinct mod codes (the reference builds no task for unknown codes, find_motifs_bin.py:152-153, but
    their rows form frequency-filter groups of their own and take part in the adjacency maximum, dataload.py:211-245), and
    rows with enough coverage but a NULL percentage (counted by pl.count() in n_positions, dataload.py:216, dropped by
    the adjacency filter): device filters == oracle filters, scoring on the surviving rows == oracle scoring."""
    from nanomotif_amd.engine import ScanEngine
    from oracle import pileup as op
    from oracle.scan import ContigPileup, score_candidates
    spec = synth.SynthSpec(n_contigs=4, total_bp=400_000, n_bins=2, mod_types=("a", "m"), seed=85, min_contig_bp=40_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    rng = np.random.default_rng(12)
    t = _raw_table(mg, rng)
    n = len(t["position"])
    # 17 extra codes (ids 3..19) on 6 % of the rows, some strongly "methylated" so that they win adjacency windows
    idx = rng.choice(n, size=n * 6 // 100, replace=False)
    t["mod_type"][idx] = rng.integers(3, 20, size=len(idx)).astype(np.int8)
    hot = idx[: len(idx) // 3]
    t["fraction_mod"][hot] = rng.choice([0.97, 0.99, 1.0], size=len(hot))
    # null percentages on 2.5 % of the rows: counted as positions of their group, never kept, never winning a window
    nul = rng.choice(n, size=n // 40, replace=False)
    frac = t["fraction_mod"].copy()
    frac[nul] = np.nan
    t["fraction_mod"] = frac
    exp = op.prefilter({k: v.copy() for k, v in t.items()})
    assert not np.isnan(exp["fraction_mod"]).any()
    eng = ScanEngine(0)
    eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(4)], mg.bin_names)
    dev_frac = np.where(np.isnan(frac), -1.0, frac)                 # the C ABI's null marker
    res = eng.ingest_pileup(t["contig"].astype(np.uint32), t["position"], t["mod_type"], t["strand"], dev_frac, t["Nvalid_cov"],
                            {0: ("m", "C"), 1: ("a", "A")})
    assert res["n_kept"] == len(exp["position"])
    kept = np.zeros((4, 8), dtype=np.int64)
    low = exp["mod_type"] < 8
    np.add.at(kept, (exp["contig"][low], exp["mod_type"][low]), 1)
    assert np.array_equal(res["kept"].astype(np.int64), kept)
    assert (exp["mod_type"] >= 8).sum() > 0                        # codes beyond the ABI's eight survived the filters too
    motifs = [("GATC", 1), ("A", 0), ("CC[AT]GG", 1), ("C", 0)]
    for code, mt in ((0, "m"), (1, "a")):
        for b in sorted(set(mg.bin_names)):
            ids = [i for i, x in enumerate(mg.bin_names) if x == b]
            pile = {}
            for i in ids:
                s = (exp["contig"] == i) & (exp["mod_type"] == code)
                pile[mg.names[i]] = ContigPileup(exp["position"][s], exp["strand"][s], exp["fraction_mod"][s])
            these = [(s, p) for s, p in motifs if Motif(s, p).split()[p] == ("A" if mt == "a" else "C")]
            got = eng.score([(Motif(s, p), mt, b) for s, p in these])
            assert np.array_equal(got, score_candidates(pile, {mg.names[i]: mg.contig_str(i) for i in ids}, these)), (mt, b)
    # the ratio test of the frequency filter sees the nulls as positions (oracle only: the device thresholds are fixed)
    small = dict(contig=np.zeros(100, np.int64), position=np.arange(100, dtype=np.int64), strand=np.full(100, ord("+"), np.uint8),
                 mod_type=np.ones(100, np.int8), fraction_mod=np.r_[np.full(60, 0.9), np.full(40, np.nan)], Nvalid_cov=np.full(100, 9))
    f = op.filter_pileup_minimummod_frequency(small, min_mod_frequency=0.6)
    assert len(f["position"]) == 0                                  # 60 / 100 is not > 0.6: the 40 nulls count as positions
    f = op.filter_pileup_minimummod_frequency({k: v[:60] for k, v in small.items()}, min_mod_frequency=0.6)
    assert len(f["position"]) == 60
    eng.close()


def test_modkit_order_takes_the_store_path_and_equals_any_other_order():
    """Rows in modkit's order (every contig in one run, positions non-decreasing, mod codes and strands interleaved) let the
    classification pass write complete plane words with plain stores; any other order (a contig in two runs, a position
    going down) falls back to atomicOr.  Same kept counts, same confident rows, same scores either way — and equal to the
    oracle filters."""
    import os
    from nanomotif_amd.engine import ScanEngine
    from oracle import pileup as op
    spec = synth.SynthSpec(n_contigs=7, total_bp=900_000, n_bins=3, mod_types=("a", "m"), seed=87, min_contig_bp=30_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    t = _raw_table(mg, np.random.default_rng(8))
    exp = op.prefilter({k: v.copy() for k, v in t.items()})
    labels = {0: ("m", "C"), 1: ("a", "A")}
    motifs = [(Motif("GATC", 1), "a"), (Motif("A", 0), "a"), (Motif("CC[AT]GG", 1), "m"), (Motif("C", 0), "m"), (Motif("T", 0), "a")]
    cands = [(m, mt, b) for b in sorted(set(mg.bin_names)) for m, mt in motifs]
    modkit = np.lexsort((t["mod_type"], t["position"], t["contig"]))
    rng = np.random.default_rng(1)
    two_runs = np.concatenate([modkit[: len(modkit) // 3], modkit[len(modkit) // 3:][::-1]])        # reversed tail: positions go down
    orders = {"modkit": modkit, "contigs in another order": modkit[np.argsort(-t["contig"][modkit], kind="stable")],
              "as generated (two runs per contig)": np.arange(len(modkit)), "reversed tail": two_runs, "shuffled": rng.permutation(len(modkit))}
    out = {}
    for name, o in orders.items():
        for env in ((None, "1") if name == "modkit" else (None,)):
            if env:
                os.environ["NM_INGEST_ATOMIC"] = env
            try:
                eng = ScanEngine(0)
                eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(len(mg.names))], mg.bin_names)
                res = eng.ingest_pileup(t["contig"][o].astype(np.uint32), t["position"][o], t["mod_type"][o], t["strand"][o],
                                        t["fraction_mod"][o], t["Nvalid_cov"][o], labels)
                rows = sorted(zip(*[x.tolist() for x in res["confident"]]))
                out[(name, env)] = (res["n_kept"], res["n_confident"], res["kept"].tolist(), rows, eng.score(cands).tolist())
                eng.close()
            finally:
                os.environ.pop("NM_INGEST_ATOMIC", None)
    first = out[("modkit", None)]
    assert first[0] == len(exp["position"]) and first[1] > 0
    for k, v in out.items():
        assert v == first, k


def test_g10_frequency_and_coverage_bounds_recorded_from_the_reference():
    """The table of fixture g10 (strict bounds > 5 / > 50 / > 1e-4 / fraction > 0.7, nulls counting as positions) through the
    device filters: the (contig, mod code) groups the REFERENCE's filter_pileup + filter_pileup_minimummod_frequency keep
    (recorded by tests/golden/gen_golden.py from dataload.py:191-226) are the groups with surviving rows, and each holds
    what the adjacency filter leaves of it."""
    from helpers import load_golden
    from nanomotif_amd.engine import ScanEngine
    from oracle import pileup as op
    from test_oracle_golden import g10_table
    from helpers import sha1
    g = load_golden("g10_frequency_filter.json")
    t = g10_table(g)
    ref = op.filter_pileup_minimummod_frequency(op.filter_pileup(t))
    assert sha1(ref["row"]) == g["after_frequency_rows_sha1"]           # the restatement reproduces the reference's rows ...
    exp = op.filter_pileup_adjacency_filter(ref)                        # ... and the KAT-pinned adjacency filter follows
    names = sorted(set(t["contig"].tolist()))
    codes = {"m": 0, "a": 1, "21839": 2}
    rng = np.random.default_rng(3)
    lengths = {c: int(t["position"][t["contig"] == c].max()) + 1 for c in names}
    eng = ScanEngine(0)
    eng.upload_assembly(names, ["".join(rng.choice(list("ACGT"), size=lengths[c])) for c in names], ["b"] * len(names))
    cid = np.array([names.index(c) for c in t["contig"].tolist()], dtype=np.uint32)
    mod = np.array([codes[m] for m in t["mod_type"].tolist()], dtype=np.int8)
    frac = np.where(np.isnan(t["fraction_mod"]), -0.01, t["fraction_mod"])       # the reader's null: percentage -1
    for order in ("grouped", "shuffled"):                                # modkit order (store path) and any other order
        idx = np.arange(len(cid))
        if order == "grouped":
            idx = np.lexsort((t["position"], mod, cid))
        else:
            np.random.default_rng(5).shuffle(idx)
        res = eng.ingest_pileup(cid[idx], t["position"][idx], mod[idx], t["strand"][idx], frac[idx], t["Nvalid_cov"][idx],
                                {0: ("m", "C"), 1: ("a", "A"), 2: ("21839", "C")}, want_rows=False)
        kept = {}
        for c, m in zip(exp["contig"].tolist(), exp["mod_type"].tolist()):
            kept[(names.index(c), codes[m])] = kept.get((names.index(c), codes[m]), 0) + 1
        want = np.zeros((len(names), 8), dtype=np.int64)
        for (c, m), n in kept.items():
            want[c, m] = n
        assert np.array_equal(res["kept"].astype(np.int64), want), order
        assert {f"{names[c]}|{[k for k, v in codes.items() if v == m][0]}" for c, m in kept} == set(g["kept_groups"])
    eng.close()


def test_g14_adjacency_filter_recorded_from_the_reference_on_the_device():
    """Fixture g14 — the reference's filter_pileup_adjacency_filter (dataload.py:228-247) EXECUTED at distance 8 on gapped, tied,
    null-bearing rows of three mod codes, 'm' and '21839' on the same positions — through the device pre-filters
    (ingest_judge_kernel and the dense-table path): the surviving rows per (contig, mod code) and the confident rows the
    planes hold are the recorded ones, in modkit order and shuffled."""
    from helpers import g14_table, load_golden
    from nanomotif_amd.engine import ScanEngine
    g = load_golden("g14_adjacency_filter.json")
    t = g14_table(g)
    names = sorted(set(t["contig"].tolist()))
    codes = {"m": 0, "a": 1, "21839": 2}
    cid = np.array([names.index(c) for c in t["contig"].tolist()], dtype=np.uint32)
    mod = np.array([codes[m] for m in t["mod_type"].tolist()], dtype=np.int8)
    st = np.array([ord(s) for s in t["strand"].tolist()], dtype=np.uint8)
    frac = np.where(np.isnan(t["fraction_mod"]), -0.01, t["fraction_mod"])             # the readers' null
    kept = np.array(g["kept_rows"]["8"], dtype=np.int64)
    want_kept = np.zeros((len(names), 8), dtype=np.int64)
    np.add.at(want_kept, (cid[kept], mod[kept]), 1)
    conf = kept[t["fraction_mod"][kept] >= 0.7]
    want_conf = sorted(zip(cid[conf].tolist(), t["position"][conf].tolist(), st[conf].tolist(), mod[conf].tolist()))
    assert len(want_conf) == g["counts"]["confident_kept"]
    rng = np.random.default_rng(2)
    length = int(t["position"].max()) + 50
    # 'm' and '21839' share one canonical base, so they get one sequence letter; the planes need the right base under a row
    seqs = []
    for c in range(len(names)):
        s = rng.choice(list("ACGT"), size=length)
        for code, base, comp in ((0, "C", "G"), (1, "A", "T"), (2, "C", "G")):
            for strand, letter in ((ord("+"), base), (ord("-"), comp)):
                sel = (cid == c) & (mod == code) & (st == strand)
                s[t["position"][sel]] = letter
        seqs.append("".join(s))
    eng = ScanEngine(0)
    eng.upload_assembly(names, seqs, ["b"] * len(names))
    for order in ("modkit", "shuffled"):
        idx = np.lexsort((mod, st, t["position"], cid)) if order == "modkit" else np.random.default_rng(7).permutation(len(cid))
        for env in ({}, {"NM_INGEST_DENSE": "1"}):
            os.environ.update(env)
            try:
                res = eng.ingest_pileup(cid[idx], t["position"][idx], mod[idx], st[idx], frac[idx], t["Nvalid_cov"][idx],
                                        {0: ("m", "C"), 1: ("a", "A"), 2: ("21839", "C")})
            finally:
                for k in env:
                    del os.environ[k]
            assert res["n_kept"] == len(kept), (order, env)
            assert np.array_equal(res["kept"].astype(np.int64), want_kept), (order, env)
            cc, cp, cs, cm = res["confident"]
            assert sorted(zip(cc.tolist(), cp.tolist(), cs.tolist(), cm.tolist())) == want_conf, (order, env)
    eng.close()
