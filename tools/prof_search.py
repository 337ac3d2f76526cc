import sys, time, json, cProfile, pstats
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth
from nanomotif_amd.engine import ScanEngine
mg = synth.make_metagenome(synth.config("cfg4"))
eng = ScanEngine(0)
pr = cProfile.Profile(); pr.enable()
rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
pr.disable()
print(json.dumps(t))
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
