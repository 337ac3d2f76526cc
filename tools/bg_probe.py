"""Where background_s of the end-to-end run goes: the deferred random.sample draws, the concatenations, nm_bg_counts."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from nanomotif_amd import synth, e2e_synth, _lib, engine as E
from nanomotif_amd.engine import ScanEngine
_lib.use_torch_allocator()
mg = synth.make_metagenome(synth.config("cfg4"))
T = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return w
E._draw_deferred_impl = timed("draw_deferred", E._draw_deferred_impl)
orig_cat = np.concatenate
for i in range(2):
    T.clear()
    eng = ScanEngine(0)
    real = eng.lib.nm_bg_counts
    class L:
        def __getattr__(self, k):
            return getattr(real_lib, k)
    real_lib = eng.lib
    calls = []
    def bg(*a):
        t0 = time.perf_counter(); r = real(*a); calls.append(time.perf_counter() - t0); return r
    proxy = type("P", (), {"__getattr__": lambda s, k: bg if k == "nm_bg_counts" else getattr(real_lib, k)})()
    eng.lib = proxy
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    print(json.dumps({k: round(v, 4) for k, v in t.items() if k in ("plan_s", "background_s", "native_search_s", "coroutines_s", "search_s")}),
          {k: round(v, 4) for k, v in T.items()}, "nm_bg_counts calls", [round(c, 4) for c in calls])
    eng.lib = real_lib
    eng.close()
