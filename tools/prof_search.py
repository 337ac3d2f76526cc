"""cProfile of the whole search at 1 Gbp (second, warm run) — where the host time of e2e_synth.run goes."""
import sys, json, cProfile, pstats
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, _lib
from nanomotif_amd.engine import ScanEngine
_lib.use_torch_allocator()
mg = synth.make_metagenome(synth.config("cfg4"))
for i in range(2):
    eng = ScanEngine(0)
    pr = cProfile.Profile()
    if i == 1:
        pr.enable()
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    if i == 1:
        pr.disable()
    eng.close() if hasattr(eng, "close") else None
    print(json.dumps({k: v for k, v in t.items() if isinstance(v, (int, float))}))
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(60)
st.sort_stats("cumtime").print_stats(50)
