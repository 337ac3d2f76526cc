#!/bin/bash
# round 4: the end-to-end pipeline beyond BASELINE's 1 Gbp on one GPU (2 and 4 Gbp: 2e9 / 4e9 raw pileup rows resident)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r4r
for gbp in 2 4; do
  timeout 900 python3 bench.py --workload e2e --total-bp ${gbp}000000000 --contigs ${gbp}0000 --bins $((gbp * 500)) > gpurun_out/r4r/e2e_${gbp}gbp.json 2> gpurun_out/r4r/e2e_${gbp}gbp.log
  echo "rc=$?"
  python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r4r/e2e_${gbp}gbp.json")); t = d["timings_rank0"]
    print("${gbp} Gbp: wall %.3f s; rows raw %d kept %d; upload_filter %.3f search %.3f (native %.3f) rounds %d candidates %d motif rows %s gpu_busy %.3f" % (
        d["value"], t["rows_raw"], t["rows_kept"], t["upload_filter_s"], t["search_s"], t["native_search_s"], t["rounds"], t["candidates"], t.get("motif_rows"), t["gpu_busy_s"]))
except Exception as e:
    print("no line:", e)
PY
  tail -2 gpurun_out/r4r/e2e_${gbp}gbp.log
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
