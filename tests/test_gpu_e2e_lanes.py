"""Bins dealt to lanes on ONE device (e2e_synth.run_lanes: one engine + one host thread per lane, whole bins each — the split of a
multi-GPU run, find_motifs_bin.py:152-171 hands bins to a process pool): the rows are those of the single engine and of the oracle
pipeline; the device phases of all lanes lie on one time line (nm_timing_intervals)."""
import numpy as np
import pytest

from nanomotif_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1800)
def test_lanes_write_the_rows_of_one_engine():
    import torch
    from helpers import oracle_pipeline
    from nanomotif_amd import e2e_synth, postprocess
    from nanomotif_amd.engine import ScanEngine
    device = torch.device("cuda:0")
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=30, total_bp=4_500_000, n_bins=10, mod_types=("a", "m"), seed=61, min_contig_bp=60_000))
    text = lambda rows: postprocess.format_bin_motifs([r for r in rows if r.n_mod + r.n_nomod >= 50])
    eng = ScanEngine(0)
    rows, t1 = e2e_synth.run(mg, eng, device)
    eng.close()
    one = text(rows)
    assert one.count("\n") > 5
    assert 0 < t1["gpu_busy_union_s"] <= t1["gpu_busy_s"] * (1 + 1e-6) + 1e-6           # a union is never longer than the sum of its parts
    for n in (2, 3):
        engines = [ScanEngine(0) for _ in range(n)]
        try:
            rows, t = e2e_synth.run_lanes(mg, engines, device)
        finally:
            for e in engines:
                e.close()
        assert text(rows) == one, n
        assert t["n_lanes"] == n and len(t["lanes"]) == n
        assert sum(x["rows_raw"] for x in t["lanes"]) == t1["rows_raw"] and t["rows_confident"] == t1["rows_confident"]
        assert 0 < t["gpu_busy_union_s"] <= t["gpu_busy_s"] * (1 + 1e-6) + 1e-6
        assert t["gpu_busy_union_s"] <= t["wall_s"]                                      # all of it happened between start and end
    assert one == oracle_pipeline(mg)


def test_timing_intervals_on_a_common_clock():
    """Two engines, scoring launches one after the other: each engine's intervals are ordered, as long as nm_timing_total_ms says, and
    engine B's — measured from engine A's first launch — begin after A's last one ended."""
    import torch
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.motif import Motif
    from nanomotif_amd import synth_device
    device = torch.device("cuda:0")
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=4, total_bp=800_000, n_bins=2, mod_types=("a",), seed=62, min_contig_bp=50_000))
    a, b = ScanEngine(0), ScanEngine(0)
    try:
        for e in (a, b):
            synth_device.load_engine_from_device(e, mg, device)
        cands = [(Motif("GATC", 1), "a", mg.bin_names[0]), (Motif("CCAGG", 2), "a", mg.bin_names[0])]
        a.timing_reset(True)
        b.timing_reset(True)
        for _ in range(3):
            a.score(cands)
        torch.cuda.synchronize(device)
        for _ in range(2):
            b.score(cands)
        ia, ib = a.timing_intervals(), b.timing_intervals(a)
        assert ia.shape == (3, 2) and ib.shape == (2, 2)
        assert ia[0, 0] == 0.0 and (ia[:, 1] >= ia[:, 0]).all() and (ia[1:, 0] >= ia[:-1, 1] - 1e-3).all()
        assert (ib[:, 0] >= ia[-1, 1] - 1e-3).all()
        ms, n = a.timing_total()
        assert n == 3 and abs(float((ia[:, 1] - ia[:, 0]).sum()) - ms) < 1e-3 * max(ms, 1.0) + 2e-3
        with pytest.raises(Exception, match="recorded no phase"):
            c = ScanEngine(0)
            try:
                a.timing_intervals(c)
            finally:
                c.close()
    finally:
        a.close()
        b.close()
