"""nm_readstats_upload / nm_contig_methylation (csrc/nmmeth.hip) and nanomotif_amd.contig_methylation: the per-contig READ
methylation table binnary starts from (nanomotif/main.py:142-193) against oracle/contig_methylation.read_methylation
(restated from epimetheus' published behaviour — parity unpinned there; the product is pinned to the restatement)."""
import numpy as np
import pytest

from nanomotif_amd import synth
from nanomotif_amd.motif import iupac_to_regex

pytestmark = pytest.mark.gpu

ZOO = ["GATC_a_1", "CCWGG_m_1", "A_a_0", "C_m_0", "GAAGNNNNNTAC_a_2", "GATC_m_3", "AA_a_0", "AA_a_1", "GCGC_m_1", "TTAA_a_2", "RGATCY_a_2",
       "G" + "N" * 35 + "AT_a_36", "CNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNG_m_0", "ACCCA_a_4", "GGCC_m_3", "VCB_m_1",
       "T" + "N" * 70 + "A_a_71", "C" + "N" * 90 + "G_m_0"]       # more than 63 positions from the modified base: the three-halo-word kernels


def _records(mg, rng):
    """{(contig, mod_type): columns} with N_mod consistent with the synthetic percentages and an N_diff column that makes
    the valid-to-diff filter bite on ~10 % of the records."""
    rec = {}
    for i, name in enumerate(mg.names):
        for mt in mg.spec.mod_types:
            p = mg.contig_pileup(i, mt)
            cov = p["nvalid"].astype(np.int64)
            nmod = np.rint(cov * p["pct_hundredths"] / 10000).astype(np.int64)
            ndiff = np.where(rng.random(len(cov)) < 0.1, rng.integers(0, 12, len(cov)), 0).astype(np.int64)
            ndiff[cov == 4] = 1                                     # 4 / 5 = 0.8 exactly: kept
            rec[(name, mt)] = dict(position=p["position"].astype(np.int64), strand=p["strand"], n_valid=cov, n_mod=nmod, n_diff=ndiff)
    return rec


def _upload(eng, mg, rec, min_cov=3, min_frac=0.8, shard=None):
    from nanomotif_amd.contig_methylation import upload_read_statistics
    kept = {}
    for mt in mg.spec.mod_types:
        cols = {k: [] for k in ("contig", "position", "strand", "n_valid", "n_mod", "n_diff")}
        for i, name in enumerate(mg.names):
            r = rec[(name, mt)]
            local = i if shard is None else shard.get(i, 0xFFFFFFFF)
            cols["contig"].append(np.full(len(r["position"]), local, np.uint32))
            for k in ("position", "strand", "n_valid", "n_mod", "n_diff"):
                cols[k].append(r[k])
        cat = {k: np.concatenate(v) for k, v in cols.items()}
        kept[mt] = upload_read_statistics(eng, mt, cat["contig"], cat["position"], cat["strand"], cat["n_valid"], cat["n_mod"], cat["n_diff"],
                                          min_cov, min_frac)
    return kept


def _oracle_rows(mg, rec, motifs, output_type, min_cov=3, min_frac=0.8, contigs=None):
    from nanomotif_amd.contig_methylation import parse_motif_mod
    from oracle.contig_methylation import read_methylation
    triples = [parse_motif_mod(m) for m in motifs]
    idx = range(len(mg.names)) if contigs is None else contigs
    seqs = {mg.names[i]: mg.contig_str(i) for i in idx}
    rows = read_methylation(rec, seqs, [(iupac_to_regex(m), mt, pos) for m, mt, pos in triples], min_cov, min_frac, output_type)
    return [(r["contig"], triples[r["motif"]][0], triples[r["motif"]][1], triples[r["motif"]][2], r["methylation_value"], r["mean_read_cov"],
             r["n_motif_obs"]) for r in rows]


def test_read_methylation_table_equals_the_oracle():
    from nanomotif_amd.contig_methylation import COLUMNS, read_methylation_table
    from nanomotif_amd.engine import ScanEngine
    spec = synth.SynthSpec(n_contigs=9, total_bp=700_000, n_bins=3, mod_types=("a", "m"), seed=41, min_contig_bp=9_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m"), ("GAAGNNNNNTAC", 2, "a")))
    mg = synth.make_metagenome(spec)
    rec = _records(mg, np.random.default_rng(3))
    eng = ScanEngine(0)
    eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(len(mg.names))], mg.bin_names)
    kept = _upload(eng, mg, rec)
    for mt in ("a", "m"):
        want = sum(int(((r["n_valid"] >= 3) & (r["n_valid"] / (r["n_valid"] + r["n_diff"]) >= 0.8)).sum()) for (n, m), r in rec.items() if m == mt)
        assert kept[mt] == want and 0 < want < sum(len(r["position"]) for (n, m), r in rec.items() if m == mt)
    for output_type in ("median", "weighted-mean"):
        rows = read_methylation_table(eng, ZOO, output_type)
        assert rows and list(rows[0]) == COLUMNS
        got = [tuple(r[c] for c in COLUMNS) for r in rows]
        want = _oracle_rows(mg, rec, ZOO, output_type)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g == w, (output_type, g, w)                      # counts, means, medians: bit-exact doubles
    planted = {(r["contig"], r["motif"]): r for r in read_methylation_table(eng, ["GATC_a_1", "CCWGG_m_1"], "median")}
    hit = [planted[(n, "GATC")]["methylation_value"] for n, b in zip(mg.names, mg.bin_names) if ("GATC", 1, "a") in mg.bin_motifs[b] and (n, "GATC") in planted]
    assert hit and min(hit) > 0.8
    # main.py:193: n_motif_obs * mean_read_cov >= --methylation_threshold (default 24)
    kept_rows = read_methylation_table(eng, ZOO, "median", methylation_threshold=24)
    assert [r for r in read_methylation_table(eng, ZOO, "median") if r["n_motif_obs"] * r["mean_read_cov"] >= 24] == kept_rows
    # other read filters: a re-upload replaces the slot
    _upload(eng, mg, rec, min_cov=12, min_frac=0.95)
    got = [tuple(r[c] for c in COLUMNS) for r in read_methylation_table(eng, ZOO[:6], "median")]
    assert got == _oracle_rows(mg, rec, ZOO[:6], "median", 12, 0.95)
    # a shard (multi-GPU: contigs of other ranks carry 0xFFFFFFFF) reports its own contigs only
    eng2 = ScanEngine(0)
    mine = [1, 4, 6]
    eng2.upload_assembly([mg.names[i] for i in mine], [mg.contig_ascii(i) for i in mine], [mg.bin_names[i] for i in mine])
    _upload(eng2, mg, rec, shard={g: k for k, g in enumerate(mine)})
    got = [tuple(r[c] for c in COLUMNS) for r in read_methylation_table(eng2, ZOO, "weighted-mean")]
    assert got == _oracle_rows(mg, rec, ZOO, "weighted-mean", contigs=mine)
    eng.close(); eng2.close()


def test_methylation_pattern_from_files_and_error_paths(tmp_path):
    """The drop-in for main.py:167-178: FASTA + bedMethyl text in, motifs-scored-read-methylation_<type>.tsv out; the shape
    of the reference's own check (tests/binnary/test_utils.py:38-59: GATC_m_3 and GATC_a_1 on two contigs -> 4 rows, 7 columns)."""
    from nanomotif_amd._lib import NmScanError
    from nanomotif_amd.contig_methylation import COLUMNS, methylation_pattern, upload_read_statistics
    from nanomotif_amd.engine import ScanEngine
    spec = synth.SynthSpec(n_contigs=2, total_bp=160_000, n_bins=1, mod_types=("a", "m"), seed=43, min_contig_bp=60_000,
                           fixed_motifs=(("GATC", 1, "a"),))
    mg = synth.make_metagenome(spec)
    mg.write_fasta(str(tmp_path / "a.fasta"))
    mg.write_bed(str(tmp_path / "p.bed"))
    with open(tmp_path / "p.bed", "a") as f:                      # a contig the assembly does not have: ignored (allow_assembly_pileup_mismatch)
        f.write("stranger\t7\t8\ta\t9\t+\t7\t8\t255,0,0\t9\t50.00\t4\t5\t0\t0\t0\t0\t0\n")
    rec = {}
    for i, name in enumerate(mg.names):
        for mt in ("a", "m"):
            p = mg.contig_pileup(i, mt)
            cov = p["nvalid"].astype(np.int64)
            rec[(name, mt)] = dict(position=p["position"].astype(np.int64), strand=p["strand"], n_valid=cov,
                                   n_mod=np.array([int(round(c * h / 10000)) for c, h in zip(cov.tolist(), p["pct_hundredths"].tolist())], np.int64),
                                   n_diff=np.zeros(len(cov), np.int64))
    motifs = ["GATC_m_3", "GATC_a_1"]
    for output_type in ("median", "weighted-mean"):
        out = tmp_path / f"motifs-scored-read-methylation_{output_type}.tsv"
        rows = methylation_pattern(str(tmp_path / "p.bed"), str(tmp_path / "a.fasta"), motifs, threads=2, min_valid_read_coverage=3,
                                   min_valid_cov_to_diff_fraction=0.8, output=str(out), output_type=output_type)
        want = _oracle_rows(mg, rec, motifs, output_type)
        assert [tuple(r[c] for c in COLUMNS) for r in rows] == want and len(rows) == 4
        text = out.read_text().splitlines()
        assert text[0].split("\t") == ['contig', 'motif', 'mod_type', 'mod_position', 'methylation_value', 'mean_read_cov', 'n_motif_obs']
        assert len(text) == 5
        back = [ln.split("\t") for ln in text[1:]]
        assert [(b[0], b[1], b[2], int(b[3]), float(b[4]), float(b[5]), int(b[6])) for b in back] == want      # repr round-trips doubles
    with pytest.raises(ValueError, match="not in the assembly"):
        methylation_pattern(str(tmp_path / "p.bed"), str(tmp_path / "a.fasta"), motifs, allow_assembly_pileup_mismatch=False)
    with pytest.raises(ValueError, match="median or weighted-mean"):
        methylation_pattern(str(tmp_path / "p.bed"), str(tmp_path / "a.fasta"), motifs, output_type="mean")
    eng = ScanEngine(0)
    eng.upload_assembly(["c"], ["ACGATCGATCGGATCCA" * 10], ["b"])
    from nanomotif_amd.contig_methylation import read_methylation_table
    with pytest.raises(NmScanError, match="holds no pileup"):
        read_methylation_table(eng, ["GATC_a_1"])
    one = dict(contig_local=[0, 0], position=[3, 3], strand=np.frombuffer(b"++", np.uint8), n_valid_cov=[9, 9], n_modified=[4, 5])
    with pytest.raises(NmScanError, match="duplicate"):
        upload_read_statistics(eng, "a", **one)
    with pytest.raises(NmScanError, match="holds no pileup"):          # the failed upload left no half-built slot behind
        read_methylation_table(eng, ["GATC_a_1"])
    with pytest.raises(NmScanError, match="n_modified outside"):
        upload_read_statistics(eng, "a", [0], [3], np.frombuffer(b"+", np.uint8), [9], [10])
    assert upload_read_statistics(eng, "a", [0, 0], [3, 7], np.frombuffer(b"++", np.uint8), [9, 2], [4, 1]) == 1     # coverage 2 < 3
    rows = read_methylation_table(eng, ["GATC_a_1", "GATC_a_1"])      # duplicates collapse (.unique(), main.py:133)
    assert rows == [dict(contig="c", motif="GATC", mod_type="a", mod_position=1, methylation_value=4 / 9, mean_read_cov=9.0, n_motif_obs=1)]
    assert upload_read_statistics(eng, "a", [], [], np.zeros(0, np.uint8), [], []) == 0
    assert read_methylation_table(eng, ["GATC_a_1"]) == []
    eng.close()
