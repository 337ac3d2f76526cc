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


def _run_env(mg, env):
    import torch
    from nanomotif_amd import e2e_synth, postprocess
    from nanomotif_amd.engine import ScanEngine
    keys = ("NM_SEARCH_NO_SENDER", "NM_SEARCH_FLIGHTS", "NM_SEARCH_ONE_FLIGHT", "NM_SEARCH_NO_SPEC")
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        eng = ScanEngine(0)
        rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
        eng.close()
    finally:
        for k in keys:
            os.environ.pop(k, None)
    return postprocess.format_bin_motifs([r for r in rows if r.n_mod + r.n_nomod >= 50]), t


@pytest.mark.timeout(1800)
def test_flights_and_the_sending_thread_change_no_row():
    """The native search splits its tasks into flights (groups that take turns on the device) and sends a flight's batches from a
    thread of their own (csrc/nmsearch.cpp: Sender) — 40 tasks here, so both are on by default.  A task's requests depend only on its
    own replies: one flight, two / three / four flights, with and without the sending thread and without the speculation must write
    the same rows, equal to the oracle pipeline; the number of lock-step iterations is what differs."""
    from helpers import oracle_pipeline
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=40, total_bp=5_000_000, n_bins=20, mod_types=("a", "m"), seed=35, min_contig_bp=60_000))
    base, t_base = _run_env(mg, {})
    assert t_base["search_iterations"] > 0 and base.count("\n") > 10
    seen = {t_base["search_iterations"]}
    for env in ({"NM_SEARCH_NO_SENDER": "1"}, {"NM_SEARCH_ONE_FLIGHT": "1"}, {"NM_SEARCH_FLIGHTS": "3"}, {"NM_SEARCH_FLIGHTS": "4", "NM_SEARCH_NO_SENDER": "1"},
                {"NM_SEARCH_FLIGHTS": "3", "NM_SEARCH_NO_SPEC": "1"}):
        got, t = _run_env(mg, env)
        assert got == base, env
        seen.add(t["search_iterations"])
    assert len(seen) >= 3, seen                       # (the groupings really differed)
    assert base == oracle_pipeline(mg)
