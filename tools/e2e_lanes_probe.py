"""The end-to-end run (bench.py --workload e2e, 1 Gbp) with the bins dealt to 1 / 2 / 3 / 4 / 6 lanes on ONE GPU (e2e_synth.run_lanes):
wall clock per lane count, rows compared with the single-lane run's."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from nanomotif_amd import _lib, e2e_synth, synth
from nanomotif_amd.engine import ScanEngine

device = torch.device("cuda:0")
torch.cuda.set_device(device)
_lib.use_torch_allocator(True)
spec = synth.config(os.environ.get("NM_PROBE_CFG", "cfg5"))
mg = synth.make_metagenome(spec)
lanes_list = [int(x) for x in os.environ.get("NM_PROBE_LANES", "1,2,3,4,6,1,2,4").split(",")]
reps = int(os.environ.get("NM_PROBE_REPS", "3"))
key = lambda r: (r.reference, r.motif, r.mod_type, r.mod_position, r.n_mod, r.n_nomod, round(r.score, 9))
first = None
for n in lanes_list:
    for rep in range(reps):
        engines = [ScanEngine(0) for _ in range(n)]
        rows, t = e2e_synth.run_lanes(mg, engines, device)
        for e in engines:
            e.close()
        ks = [key(r) for r in rows]
        if first is None:
            first = ks
        lane_walls = [round((x["upload_filter_s"] + x["search_s"]) * 1e3, 1) for x in t["lanes"]]
        busy = sum(x["gpu_busy_s"] for x in t["lanes"])
        phases = [{k[:-2]: round(x.get(k, 0.0) * 1e3, 1) for k in ("upload_assembly_s", "ingest_call_s", "plan_s", "background_s", "native_search_s", "postprocess_s", "run_call_s")}
                  for x in t["lanes"]]
        print(json.dumps({"lanes": n, "rep": rep, "wall_ms": round(t["wall_s"] * 1e3, 2), "lane_ms": lane_walls, "sum_of_phase_ms": round(busy * 1e3, 2), "busy_union_ms": round(t["gpu_busy_union_s"] * 1e3, 2), "phases_ms": phases,
                          "rows": len(rows), "same_rows_as_first": ks == first, "prewarm_ms": round(t["allocator_prewarm_s"] * 1e3, 1)}), flush=True)
        del rows, t, engines
        torch.cuda.empty_cache() if os.environ.get("NM_PROBE_EMPTY") else None
