"""Fixed cost of the two batches of a lock-step round: a window batch of ONE request and a scoring batch of ONE candidate
on a small bin, synchronous round trips and the begin / end halves back to back (what a round of nm_search_run pays when
only a few tasks are left)."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np
from nanomotif_amd import synth, _lib
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.motif import Motif
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=200, total_bp=20_000_000, n_bins=10, mod_types=("a", "m"), seed=3))
eng = ScanEngine(0)
eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(len(mg.names))], mg.bin_names)
for mt in ("a", "m"):
    for i in range(len(mg.names)):
        p = mg.contig_pileup(i, mt)
        eng.upload_pileup(mt, np.full(len(p["position"]), i, np.uint32), p["position"], p["strand"], synth.pct_to_fraction(p["pct_hundredths"]), append=i > 0)
lib = eng.lib
bins = sorted(set(mg.bin_names))
batch = eng.make_batch([(Motif("GATC", 1), "a", bins[0])])
args = eng._batch_args(batch)
out = np.zeros((1, 2), dtype=np.int64)
rng = np.random.default_rng(3)
sets = rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), size=(15000, 41))
tid = C.c_uint32(0)
_lib.check(lib.nm_win_add_task(eng.ctx, 15000, 41, sets.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(tid)))
p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
req_task = np.array([tid.value], dtype=np.uint32); req_kind = np.zeros(1, dtype=np.uint8); req_sets = np.full((1, 64), 15, dtype=np.uint8); req_sets[0, 20] = 1
wout = np.zeros((1, 2 + 4 * 64), dtype=np.int32)
def timeit(fn, n=300):
    for _ in range(30): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
def score_sync(): _lib.check(lib.nm_score_batch(eng.ctx, *args, p(out, C.c_int64)))
def win_sync(): _lib.check(lib.nm_win_batch_w(eng.ctx, 1, p(req_task, C.c_uint32), p(req_kind, C.c_uint8), p(req_sets, C.c_uint8), 64, p(wout, C.c_int32)))
def both():
    _lib.check(lib.nm_win_batch_w_begin(eng.ctx, 1, p(req_task, C.c_uint32), p(req_kind, C.c_uint8), p(req_sets, C.c_uint8), 64))
    _lib.check(lib.nm_score_batch_begin(eng.ctx, *args))
    _lib.check(lib.nm_win_batch_w_end(eng.ctx, p(wout, C.c_int32)))
    _lib.check(lib.nm_score_batch_end(eng.ctx, p(out, C.c_int64)))
print("scoring batch, 1 candidate on a 2 Mbp bin, round trip: %.1f us" % timeit(score_sync))
print("window batch, 1 request on 15 000 windows, round trip: %.1f us" % timeit(win_sync))
print("both halves back to back (a round): %.1f us" % timeit(both))
f = C.c_float(0); lib.nm_last_kernel_ms(eng.ctx, C.byref(f)); print("last scoring kernel: %.1f us" % (f.value * 1e3))
eng.close()
