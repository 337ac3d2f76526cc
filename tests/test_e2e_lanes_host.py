"""Host logic of the lanes of the end-to-end run (e2e_synth.lane_bins / union_ms): no GPU."""
import numpy as np

from nanomotif_amd import synth
from nanomotif_amd.e2e_synth import lane_bins, union_ms


def test_union_of_intervals():
    assert union_ms(np.zeros((0, 2))) == 0.0
    assert union_ms([[0, 1], [0.5, 2], [3, 4], [3.5, 3.7], [10, 10]]) == 3.0
    assert union_ms([[5, 6], [0, 10]]) == 10.0                           # one phase inside another
    assert union_ms([[2, 3], [0, 1], [1, 2]]) == 3.0                     # touching, out of order
    rng = np.random.default_rng(5)
    for _ in range(50):
        iv = np.sort(rng.integers(0, 200, size=(rng.integers(1, 40), 2)), axis=1).astype(np.float64)
        covered = np.zeros(200, dtype=bool)
        for a, b in iv.astype(int):
            covered[a:b] = True
        assert union_ms(iv) == float(covered.sum())


def test_lane_bins_are_whole_bins_in_balance():
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=60, total_bp=3_000_000, n_bins=12, mod_types=("a",), seed=3, min_contig_bp=10_000))
    size = {}
    for i, b in enumerate(mg.bin_names):
        size[b] = size.get(b, 0) + int(mg.lengths[i])
    for n in (1, 2, 3, 5):
        groups = lane_bins(mg, n)
        assert len(groups) == n
        assert sorted(b for g in groups for b in g) == sorted(size)       # every bin once
        load = [sum(size[b] for b in g) for g in groups]
        assert max(load) - min(load) <= max(size.values())                 # longest-first to the lightest lane
    assert lane_bins(mg, 2) == lane_bins(mg, 2)
