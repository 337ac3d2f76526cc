"""nm_score_batch_per_contig: counters keyed by (candidate, contig) — motif_model_contig (find_motifs_bin.py:1285-1331)
for every contig of the candidate's bin in one launch — against the oracle contig by contig, and the per-contig motif
methylation table built on it (nanomotif_amd/contig_methylation.py; input shape of binnary, main.py:140-178)."""
import numpy as np
import pytest

from helpers import oracle_bin_inputs
from nanomotif_amd import synth
from nanomotif_amd.motif import Motif

pytestmark = pytest.mark.gpu


def _engine(mg, mod_types):
    from nanomotif_amd.engine import ScanEngine
    eng = ScanEngine(0)
    eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(len(mg.names))], mg.bin_names)
    for mt in mod_types:
        cols = mg.pileup_columns(mt)
        keep = cols["nvalid"] > 5
        eng.upload_pileup(mt, cols["contig_id"][keep], cols["position"][keep], cols["strand"][keep], cols["fraction_mod"][keep])
    return eng


def test_per_contig_counts_match_oracle_and_sum_to_the_bin_table():
    from oracle.contig_methylation import per_contig_counts
    spec = synth.SynthSpec(n_contigs=14, total_bp=900_000, n_bins=3, mod_types=("a", "m"), seed=31, min_contig_bp=9_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m"), ("GAAGNNNNNTAC", 2, "a")))
    mg = synth.make_metagenome(spec)
    eng = _engine(mg, ("a", "m"))
    zoo = [("GATC", 1, "a"), ("CC[AT]GG", 1, "m"), ("GAAG.....TAC", 2, "a"), ("A", 0, "a"), ("C", 0, "m"), ("G[AG].GAAG[CT]", 5, "a"),
           ("." * 19 + "GATC" + "." * 18, 20, "a"), ("GATC", 3, "m"), ("T" + "." * 39 + "A", 40, "a")]
    zoo += [(s, p, mt) for s, p, mt in synth.random_candidates(60, seed=9)]          # > 32 per group: several LDS-free passes
    bins = sorted(set(mg.bin_names))
    cands = [(Motif(s, p), mt, b) for b in bins for s, p, mt in zoo]
    per = eng.score_per_contig(cands)
    total = eng.score(cands)
    k = 0
    for b in bins:
        idx = [i for i, x in enumerate(mg.bin_names) if x == b]
        inputs = {mt: oracle_bin_inputs(mg, mt, contigs=idx) for mt in ("a", "m")}
        for s, p, mt in zoo:
            names, table = per[k]
            assert names == [mg.names[i] for i in idx]                  # upload order inside the bin
            pile, seqs = inputs[mt]
            want = per_contig_counts(pile, seqs, s, p)
            assert table.tolist() == [list(want[n]) for n in names], (b, s, p, mt)
            assert np.array_equal(table.sum(axis=0), total[k])
            k += 1
    assert sum(t.sum() for _, t in per) > 0
    eng.close()


def test_contig_methylation_table_for_binnary():
    """Every bin-consensus motif on every contig (also the contigs of bins that do not carry it): a contaminant contig
    shows up as a row whose methylation differs from its bin's."""
    from nanomotif_amd.contig_methylation import CONFIDENT_COLUMNS as COLUMNS, confident_site_table as contig_methylation
    spec = synth.SynthSpec(n_contigs=8, total_bp=800_000, n_bins=2, mod_types=("a",), seed=32, min_contig_bp=40_000)
    mg = synth.make_metagenome(spec)
    # different motifs per bin
    mg.bin_motifs = {"bin_000": [("GATC", 1, "a")], "bin_001": [("ACCCA", 4, "a")]}
    eng = _engine(mg, ("a",))
    rows = contig_methylation(eng, [("GATC", "a", 1), ("ACCCA", "a", 4)])
    assert len(rows) <= 2 * 8 and list(rows[0]) == COLUMNS
    by = {(r["contig"], r["motif"]): r for r in rows}
    for i, name in enumerate(mg.names):
        own, other = ("GATC", "ACCCA") if mg.bin_names[i] == "bin_000" else ("ACCCA", "GATC")
        assert by[(name, own)]["confident_methylated_fraction"] > 0.9 and by[(name, own)]["n_confident_sites"] > 50
        assert by[(name, other)]["confident_methylated_fraction"] < 0.1
        assert by[(name, own)]["n_mod"] + by[(name, own)]["n_nomod"] == by[(name, own)]["n_confident_sites"]
    eng.close()
