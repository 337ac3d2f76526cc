"""cProfile of the 1 Gbp end-to-end run after the device filters (window pipeline + discover()) — the Python that is left."""
import cProfile, pstats, sys, io, os
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, find_motifs_bin as fmb, main as nm_main
from nanomotif_amd.engine import ScanEngine
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # size in Gbp (10 000 contigs / 500 bins per Gbp)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=10_000 * G, total_bp=1_000_000_000 * G, n_bins=500 * G, mod_types=("a", "m"), seed=1))
prof = cProfile.Profile()
def wrap(orig):
    def wrapped(*a, **k):
        prof.enable()
        try:
            return orig(*a, **k)
        finally:
            prof.disable()
    return wrapped
e2e_synth.discover = wrap(fmb.discover)
nm_main.device_window_pipeline = wrap(nm_main.device_window_pipeline)
for rep in range(3):
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    print({k: round(v, 4) for k, v in t.items() if k.endswith("_s")}, len(rows), flush=True)
for key in ("tottime", "cumulative"):
    s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats(key).print_stats(32); print(s.getvalue()[:9000])
