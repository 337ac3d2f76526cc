"""cProfile of discover() (everything of the 1 Gbp end-to-end run after the device filters) — the Python that is left."""
import cProfile, pstats, sys, io
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, find_motifs_bin as fmb
from nanomotif_amd.engine import ScanEngine
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=10_000, total_bp=1_000_000_000, n_bins=500, mod_types=("a", "m"), seed=1))
prof = cProfile.Profile()
orig = fmb.discover
def wrapped(*a, **k):
    prof.enable()
    try:
        return orig(*a, **k)
    finally:
        prof.disable()
e2e_synth.discover = wrapped
for rep in range(2):
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    print({k: round(v, 4) for k, v in t.items() if k.endswith("_s")}, len(rows), flush=True)
s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:7000])
