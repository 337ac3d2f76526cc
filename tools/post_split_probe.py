import sys, time
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, native_search as ns
from nanomotif_amd.engine import ScanEngine
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=10_000, total_bp=1_000_000_000, n_bins=500, mod_types=("a", "m"), seed=1))
orig_post = ns.SearchResults.postprocess
orig_final = ns.PostResults.final
acc = {"native": 0.0, "rows": 0.0, "init": 0.0}
orig_init = ns.PostResults.__init__
def post(self, *a, **k):
    t0 = time.perf_counter(); r = orig_post(self, *a, **k); acc["native"] += time.perf_counter() - t0; return r
def init(self, *a, **k):
    t0 = time.perf_counter(); orig_init(self, *a, **k); acc["init"] += time.perf_counter() - t0
def final(self, t):
    t0 = time.perf_counter(); r = orig_final(self, t); acc["rows"] += time.perf_counter() - t0; return r
ns.SearchResults.postprocess = post; ns.PostResults.final = final; ns.PostResults.__init__ = init
for rep in range(3):
    for k in acc: acc[k] = 0.0
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    print("postprocess_s %.4f: nm_post_run + export %.4f (export/arrays %.4f), building the final rows %.4f" % (t["postprocess_s"], acc["native"], acc["init"], acc["rows"]), flush=True)
