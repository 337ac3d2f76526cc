import sys, time
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth
from nanomotif_amd.engine import ScanEngine
mg = synth.make_metagenome(synth.config(sys.argv[1] if len(sys.argv) > 1 else "cfg3"))
eng = ScanEngine(0)
e2e_synth.load_and_filter(eng, mg, torch.device("cuda:0"))
for base in "AACCG":
    t0 = time.perf_counter(); eng.contig_base_counts(base, 20); print(base, f"{(time.perf_counter()-t0)*1e3:.2f} ms")
