"""cProfile of the post-processing coroutines of the 1 Gbp end-to-end run (the part of e2e that is still Python)."""
import cProfile, pstats, sys, time, io
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, find_motifs_bin as fmb
from nanomotif_amd.engine import ScanEngine
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=10_000, total_bp=1_000_000_000, n_bins=500, mod_types=("a", "m"), seed=1))
prof = cProfile.Profile()
orig = fmb.run_lockstep
def wrapped(*a, **k):
    t0 = time.perf_counter()
    if len(sys.argv) > 1: prof.enable()
    try:
        return orig(*a, **k)
    finally:
        prof.disable()
        print("run_lockstep wall %.4f s" % (time.perf_counter() - t0), flush=True)
fmb.run_lockstep = wrapped
for rep in range(2):
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    print({k: round(v, 4) for k, v in t.items() if k.endswith("_s")}, len(rows), flush=True)
if len(sys.argv) > 1:
    s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
    s = io.StringIO(); pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:6000])
