"""Wall time per scoring step (no HIP events) for a given shard size: what sits between two scoring kernels.
argv: total_bp contigs bins candidates [lanes]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nanomotif_amd import synth, synth_device
from nanomotif_amd.engine import ScanEngine
bp, contigs, bins, cands_n = (int(x) for x in sys.argv[1:5])
lanes = int(sys.argv[5]) if len(sys.argv) > 5 else 1
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=contigs, total_bp=bp, n_bins=bins, mod_types=("a", "m"), seed=1))
dev = torch.device("cuda:0")
eng = ScanEngine(0)
synth_device.load_engine_from_device(eng, mg, dev)
for w in ("cfg5", "greedy"):
    c = bench.build_candidates(mg, w, cands_n, 2)
    b = eng.make_batch(c)
    out = [torch.zeros((len(c), 2), dtype=torch.int64, device=dev) for _ in range(2)]
    eng.set_score_lanes(lanes)
    for k in range(10):
        eng.score_into_device(b, out[k & 1].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for k in range(n):
        eng.score_into_device(b, out[k & 1].data_ptr())
    torch.cuda.synchronize()
    eng.set_score_lanes(1)
    print(os.environ.get("TAG", ""), w, bp, "lanes", lanes, "ms/step %.4f" % ((time.perf_counter() - t0) / n * 1e3), int(out[1].sum()))
