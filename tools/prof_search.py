"""cProfile of the whole search at 1 Gbp — where the host time of e2e_synth.run goes.  argv[1]: which run to profile
(0 = the first, cold one: imports and first-use costs included; 1 = the second)."""
import sys, json, cProfile, pstats
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, _lib
from nanomotif_amd.engine import ScanEngine
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
_lib.use_torch_allocator()
mg = synth.make_metagenome(synth.config("cfg4"))
for i in range(which + 1):
    eng = ScanEngine(0)
    pr = cProfile.Profile()
    if i == which:
        pr.enable()
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    if i == which:
        pr.disable()
    eng.close()
    print(json.dumps({k: round(v, 4) for k, v in t.items() if isinstance(v, float)}))
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats("find_motifs_bin|postprocess|search.py|native_search|motif.py|importlib|networkx", 45)
