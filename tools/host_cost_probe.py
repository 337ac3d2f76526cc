"""Host time of ONE nm_score_batch_device call with the GPU idle (synchronised before every call): the API + validation +
sort + staging cost that bounds the step rate when the kernel is short."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nanomotif_amd import synth, synth_device
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.shard import assign_contigs
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=10000, total_bp=1_000_000_000, n_bins=500, mod_types=("a", "m"), seed=1))
dev = torch.device("cuda:0")
cands = bench.build_candidates(mg, "cfg5", 10000, 2)
greedy = bench.build_candidates(mg, "greedy", 0, 2)
for label, part in (("whole 1 Gbp", None), ("1/8 shard", assign_contigs(mg.lengths, 8, bins=mg.bin_names)[0])):
    eng = ScanEngine(0)
    synth_device.load_engine_from_device(eng, mg, dev, contigs=part)
    for name, c in (("cfg5 10k", cands), ("greedy 2k", greedy)):
        b = eng.make_batch(c)
        out = torch.zeros((len(c), 2), dtype=torch.int64, device=dev)
        ts = []
        for k in range(60):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.score_into_device(b, out.data_ptr())
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        print(label, name, "host us per call: median %.1f min %.1f" % (np.median(ts[10:]) * 1e6, np.min(ts[10:]) * 1e6), "resident candidates", eng.stats()["last_compact"] + eng.stats()["last_general"])
    eng.close()
    torch.cuda.empty_cache()
