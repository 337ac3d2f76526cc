import sys, time, json, cProfile, pstats
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth
from nanomotif_amd.engine import ScanEngine
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
spec = synth.config(name) if name.startswith("cfg") else synth.SynthSpec(n_contigs=40, total_bp=8_000_000, n_bins=4, mod_types=("a", "m"), seed=1)
mg = synth.make_metagenome(spec)
eng = ScanEngine(0)
pr = cProfile.Profile(); pr.enable()
rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
pr.disable()
print(json.dumps(t))
print(len(rows), sorted({(r.reference, r.motif_iupac) for r in rows})[:12])
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
