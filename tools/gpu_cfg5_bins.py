"""The 1 Gbp end-to-end run (cfg 5: 1e9 raw rows, 1 000 searches) against the oracle pipeline on MANY of its 500 bins (the suite's
test samples 16): `python3 tools/gpu_cfg5_bins.py [n_bins [seed]]` on the GPU box — bin-motifs.tsv of the sampled bins, text for text."""
import os
import sys
import time

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch

from helpers import oracle_pipeline_parallel
from nanomotif_amd import e2e_synth, postprocess, synth
from nanomotif_amd.engine import ScanEngine

def main():
    n_bins = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    procs = max(1, min(32, (os.cpu_count() or 2) - 1))
    mg = synth.make_metagenome(synth.config("cfg5"))
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    bins = sorted(set(mg.bin_names))
    rng = np.random.Generator(np.random.PCG64(seed))
    sample = [bins[i] for i in sorted(rng.choice(len(bins), size=n_bins, replace=False).tolist())]
    got = postprocess.format_bin_motifs([r for r in rows if r.reference in set(sample) and r.n_mod + r.n_nomod >= 50])
    t0 = time.time()
    bad = 0
    for a in range(0, n_bins, 4 * procs):                       # (in groups: a mismatch names its bins early)
        part = sample[a:a + 4 * procs]
        exp = oracle_pipeline_parallel(mg, part, procs)
        mine = postprocess.format_bin_motifs([r for r in rows if r.reference in set(part) and r.n_mod + r.n_nomod >= 50])
        ok = mine == exp
        bad += not ok
        print(f"bins {a}..{a + len(part) - 1}: {'equal' if ok else 'MISMATCH'} ({exp.count(chr(10)) - 1} motif rows, oracle {time.time() - t0:.0f} s so far)", flush=True)
        if not ok:
            print(mine[:3000], "\n---\n", exp[:3000])
    print(f"cfg 5 full loop: {n_bins} of {len(bins)} bins compared with the oracle pipeline, mismatching groups: {bad}; rounds {t['rounds']}, rows raw {t['rows_raw']}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":          # (the oracle workers are SPAWNED: they import this file)
    main()
