"""BASELINE.json's configurations at their STATED shapes, against the CPU oracle (SURVEY.md §7 minimum slice, §8(d)):

* cfg 2 — one 5 Mbp contig, 6mA: >= 200 candidate motifs bit-exact;
* cfg 5 — the 1 Gbp / 10 000-contig / 500-bin metagenome with the 10 000-candidate table bench.py times: counts of
  >= 32 whole bins bit-exact against the oracle, and the 8-way contig shard sums to the unsharded table;
* cfg 5, full candidate-expansion loop at 1 Gbp (raw rows -> device filters -> windows -> lock-step greedy search ->
  post-processing): the motif rows of >= 8 bins byte-equal to the oracle's pipeline run on those bins.

The oracle legs run in spawn pools on the host cores (reference: find_motifs_bin.py:606-839, 1265-1331).
"""
import os

import numpy as np
import pytest

from helpers import oracle_pipeline_parallel, spec_kwargs
from nanomotif_amd import synth
from nanomotif_amd.motif import Motif

pytestmark = pytest.mark.gpu

PROCS = max(1, min(32, (os.cpu_count() or 2) - 1))


def _pool_scores(jobs):
    import multiprocessing as mp
    from oracle.pipeline import score_worker
    with mp.get_context("spawn").Pool(min(PROCS, len(jobs))) as pool:
        return pool.map(score_worker, jobs, chunksize=1)


@pytest.mark.timeout(1200)
def test_cfg2_5mbp_contig_200_candidates_bit_exact():
    """'counts bit-exact for >= 200 candidate motifs' on the 5 Mbp contig of cfg 2."""
    from helpers import motif_zoo
    from nanomotif_amd.engine import ScanEngine
    spec = synth.config("cfg2")
    mg = synth.make_metagenome(spec)
    cands = []
    for s, p in motif_zoo():
        m = Motif(s, p).new_stripped_motif()
        if m.string[m.mod_position] == "A":                 # cfg 2 carries a 6mA pileup only
            cands.append((s, p, "a"))
    cands += [c for c in synth.random_candidates(400, seed=2, mod_types=("a",))][:260 - len(cands)]
    assert len(cands) >= 200
    eng = ScanEngine(0)
    eng.upload_assembly(mg.names, [mg.contig_ascii(0)], mg.bin_names)
    cols = mg.pileup_columns("a")
    keep = cols["nvalid"] > 5
    eng.upload_pileup("a", cols["contig_id"][keep], cols["position"][keep], cols["strand"][keep], cols["fraction_mod"][keep])
    got = eng.score([(Motif(s, p), mt, mg.bin_names[0]) for s, p, mt in cands])
    eng.close()
    n_jobs = min(PROCS, 16)
    jobs = [(spec_kwargs(spec), mg.bin_names[0], cands[k::n_jobs]) for k in range(n_jobs)]
    exp = np.zeros_like(got)
    for k, (_, table, _, _, _) in enumerate(_pool_scores(jobs)):
        exp[k::n_jobs] = np.asarray(table, dtype=np.int64)
    assert np.array_equal(got, exp), np.flatnonzero((got != exp).any(axis=1))[:10]
    assert got.sum() > 0 and (got.sum(axis=1) > 0).sum() > 150


@pytest.mark.timeout(2400)
def test_cfg3_full_loop_every_bin():
    """cfg 3 (100 Mbp, 1000 contigs, 50 bins, 6mA + 5mC): the whole pipeline on the device, and EVERY bin's motif rows
    byte-equal to the oracle's pipeline (filters -> search -> post-processing) — 100 searches, none sampled away."""
    import torch
    from nanomotif_amd import e2e_synth, postprocess
    from nanomotif_amd.engine import ScanEngine
    spec = synth.config("cfg3")
    mg = synth.make_metagenome(spec)
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    bins = sorted(set(mg.bin_names))
    got = postprocess.format_bin_motifs([r for r in rows if r.n_mod + r.n_nomod >= 50])
    exp = oracle_pipeline_parallel(mg, bins, PROCS)
    assert got == exp
    assert got.count("\n") > 50 and t["rounds"] > 20


@pytest.mark.timeout(2400)
def test_cfg5_10k_candidates_1gbp():
    """The exact workload of the default bench.py run: 10 000 seed-2 candidates (20 per bin) on the seed-1 1 Gbp
    metagenome.  >= 32 whole bins against the oracle, the shard-sum invariant the RCCL all-reduce relies on, and the
    sharded table through the same assignment bench.py --gpus 8 uses."""
    import torch
    import bench
    from nanomotif_amd import synth_device
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.shard import assign_contigs
    spec = synth.config("cfg5")
    mg = synth.make_metagenome(spec)
    cands = bench.build_candidates(mg, "cfg5", 10_000, 2)
    assert len(cands) == 10_000
    eng = ScanEngine(0)
    synth_device.load_engine_from_device(eng, mg, torch.device("cuda:0"))
    whole = eng.score(cands)
    eng.close()
    torch.cuda.empty_cache()
    bins = sorted(set(mg.bin_names))
    sample = [bins[(k * 37) % len(bins)] for k in range(max(32, min(PROCS * 2, 64)))]
    jobs = []
    for b in sample:
        idx = [k for k, c in enumerate(cands) if c[2] == b]
        jobs.append((spec_kwargs(spec), b, [(cands[k][0].string, cands[k][0].mod_position, cands[k][1]) for k in idx]))
    checked = 0
    for (b, table, _, _, _), job in zip(_pool_scores(jobs), jobs):
        idx = [k for k, c in enumerate(cands) if c[2] == b]
        assert np.array_equal(whole[idx], np.asarray(table, dtype=np.int64)), b
        checked += len(idx)
    assert checked >= 32 * 20
    total = np.zeros_like(whole)
    for part in assign_contigs(mg.lengths, 8, bins=mg.bin_names):
        e = ScanEngine(0)
        synth_device.load_engine_from_device(e, mg, torch.device("cuda:0"), contigs=part)
        total += e.score(cands)
        e.close()
        torch.cuda.empty_cache()
    assert np.array_equal(total, whole) and whole.sum() > 0


@pytest.mark.timeout(3000)
def test_cfg5_full_loop_1gbp():
    """cfg 5 'full candidate-expansion loop' at 1 Gbp: 1e9 raw pileup rows through the device filters, 1000
    (bin, mod type) searches in lock-step, post-processing; the rows of 8+ seeded bins must equal the oracle's."""
    import torch
    from nanomotif_amd import e2e_synth, postprocess
    from nanomotif_amd.engine import ScanEngine
    spec = synth.config("cfg5")
    mg = synth.make_metagenome(spec)
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    assert t["rows_raw"] > 900_000_000 and t["rounds"] > 50
    bins = sorted(set(mg.bin_names))
    rng = np.random.Generator(np.random.PCG64(2025))
    sample = sorted(rng.choice(len(bins), size=max(8, min(PROCS, 16)), replace=False).tolist())
    sample = [bins[i] for i in sample]
    got = postprocess.format_bin_motifs([r for r in rows if r.reference in set(sample) and r.n_mod + r.n_nomod >= 50])
    exp = oracle_pipeline_parallel(mg, sample, PROCS)
    assert got == exp
    assert got.count("\n") > len(sample)                      # at least one motif per sampled bin on average
    planted = {(b, m[0]) for b in sample for m in mg.bin_motifs[b]}
    found = {(r.reference, r.motif_iupac) for r in rows}
    assert len(planted & found) >= 0.5 * len(planted)
