"""The torch/device synthetic generator must be bit-identical to the numpy one."""
import numpy as np
import pytest

from nanomotif_amd import synth

pytestmark = pytest.mark.gpu


def test_device_generator_matches_host():
    import torch
    from nanomotif_amd import synth_device
    spec = synth.SynthSpec(n_contigs=9, total_bp=400_000, n_bins=3, mod_types=("a", "m"), seed=5, min_contig_bp=5_000)
    mg = synth.make_metagenome(spec)
    dev = torch.device("cuda:0")
    for b in sorted(set(mg.bin_names)):
        db = synth_device.generate_bin(mg, b, dev, min_cov=-1)
        st = db.starts.tolist()
        for k, i in enumerate(db.contigs):
            got = db.ascii_cat[st[k]:st[k] + int(mg.lengths[i])].cpu().numpy()
            assert np.array_equal(got, mg.contig_ascii(i)), (b, i)
        for mt in ("a", "m"):
            p = db.pileups[mt]
            cid = p["contig_id"].cpu().numpy()
            for i in db.contigs:
                sel = cid == i
                host = mg.contig_pileup(i, mt)
                order = np.lexsort((p["strand"].cpu().numpy()[sel], p["position"].cpu().numpy()[sel]))
                horder = np.lexsort((host["strand"], host["position"]))
                assert np.array_equal(p["position"].cpu().numpy()[sel][order], host["position"][horder])
                assert np.array_equal(p["strand"].cpu().numpy()[sel][order], host["strand"][horder])
                assert np.array_equal(p["fraction_mod"].cpu().numpy()[sel][order],
                                      synth.pct_to_fraction(host["pct_hundredths"])[horder])
                assert np.array_equal(p["nvalid"].cpu().numpy()[sel][order], host["nvalid"][horder])


def test_engine_loaded_from_device_matches_host_upload():
    import torch
    from nanomotif_amd import synth_device
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.motif import Motif
    spec = synth.SynthSpec(n_contigs=9, total_bp=400_000, n_bins=3, mod_types=("a", "m"), seed=6, min_contig_bp=5_000)
    mg = synth.make_metagenome(spec)
    cands = [(Motif(s, p), mt, b) for b in sorted(set(mg.bin_names)) for s, p, mt in synth.random_candidates(30, seed=3)]
    a = ScanEngine(0)
    synth_device.load_engine_from_device(a, mg, torch.device("cuda:0"))
    b = ScanEngine(0)
    b.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(9)], mg.bin_names)
    for mt in ("a", "m"):
        cols = mg.pileup_columns(mt)
        keep = cols["nvalid"] > 5
        b.upload_pileup(mt, cols["contig_id"][keep], cols["position"][keep], cols["strand"][keep], cols["fraction_mod"][keep])
    ra, rb = a.score(cands), b.score(cands)
    assert np.array_equal(ra, rb) and ra.sum() > 0
    # a two-rank shard of the same metagenome sums to the same table
    from nanomotif_amd.shard import assign_contigs
    parts = assign_contigs(mg.lengths, 2)
    tot = np.zeros_like(ra)
    for part in parts:
        e = ScanEngine(0)
        synth_device.load_engine_from_device(e, mg, torch.device("cuda:0"), contigs=part)
        tot += e.score(cands)
        e.close()
    assert np.array_equal(tot, ra)
    a.close(); b.close()


def test_end_to_end_synthetic_run_matches_oracle_pipeline():
    """Device generation -> device filters -> lock-step search -> rows, against the CPU oracle's full pipeline."""
    import torch
    from helpers import oracle_pipeline
    from nanomotif_amd import e2e_synth, postprocess
    from nanomotif_amd.engine import ScanEngine
    spec = synth.SynthSpec(n_contigs=6, total_bp=700_000, n_bins=2, mod_types=("a", "m"), seed=91, min_contig_bp=50_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m"), ("ACCCA", 4, "a")))
    mg = synth.make_metagenome(spec)
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    rows = [r for r in rows if r.n_mod + r.n_nomod >= 50]
    assert postprocess.format_bin_motifs(rows) == oracle_pipeline(mg)
    assert t["rows_raw"] == 700_000 and t["rows_kept"] < t["rows_raw"] and t["rounds"] < t["candidates"]


@pytest.mark.parametrize("seed", [5, 13, 38])
def test_random_metagenomes_end_to_end_equal_the_oracle_pipeline(seed):
    """Seeds of tools/e2e_fuzz.py (90 seeds there: 0 mismatches): random contigs / bins / mod types / planted motifs /
    methylation rates through device filters -> windows -> native search -> native post-processing, against the CPU oracle's
    whole pipeline — bin-motifs.tsv text for text."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import e2e_fuzz
    assert "motif rows" in e2e_fuzz.one(seed, 6)

