"""Speculative children of the native search (csrc/nmwindows.hip: spec_children_kernel; csrc/nmsearch.cpp: take_speculation): the
window batch of a lock-step round also picks the arg-max-KL column of every PSSM request on the device and scores the children there
(find_motifs_bin.py:957-1023, :1116-1135).  The host's own pick stays authoritative, so the rows must be IDENTICAL with and without
the speculation (NM_SEARCH_NO_SPEC=1), and equal to the oracle pipeline; what changes is the number of lock-step iterations."""
import os

import pytest

from nanomotif_amd import synth

pytestmark = pytest.mark.gpu


def _run(mg, spec_on):
    import torch
    from nanomotif_amd import e2e_synth, postprocess
    from nanomotif_amd.engine import ScanEngine
    if spec_on:
        os.environ.pop("NM_SEARCH_NO_SPEC", None)
    else:
        os.environ["NM_SEARCH_NO_SPEC"] = "1"
    try:
        eng = ScanEngine(0)
        rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
        eng.close()
    finally:
        os.environ.pop("NM_SEARCH_NO_SPEC", None)
    return postprocess.format_bin_motifs([r for r in rows if r.n_mod + r.n_nomod >= 50]), t


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("seed, kw", [(31, {}), (32, dict(methylated_fraction=0.9)), (33, dict(n_bins=3, n_contigs=9, total_bp=1_500_000))])
def test_rows_are_identical_with_and_without_speculation(seed, kw):
    from helpers import oracle_pipeline
    base = dict(n_contigs=12, total_bp=3_000_000, n_bins=6, mod_types=("a", "m"), seed=seed, min_contig_bp=50_000)
    base.update(kw)
    mg = synth.make_metagenome(synth.SynthSpec(**base))
    on, t_on = _run(mg, True)
    off, t_off = _run(mg, False)
    assert on == off
    assert t_off["speculation_hits"] == 0 and t_off["speculation_misses"] == 0
    assert t_on["speculation_hits"] > 0
    # most children are answered by the speculation: fewer scoring batches and fewer lock-step iterations
    assert t_on["speculation_hits"] > 5 * t_on["speculation_misses"], t_on
    assert t_on["search_iterations"] < t_off["search_iterations"], (t_on["search_iterations"], t_off["search_iterations"])
    assert on == oracle_pipeline(mg)
    assert on.count("\n") > 3
