#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_ingest_prof.sh r3k > /dev/null 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r3k/kernel_stats.csv")):
    n=r["Name"]
    if any(k in n for k in ("ingest","popcount","compact")):
        print("  ",n[23:50], r["Calls"], round(float(r["AverageNs"])/1e6,3),"ms")
PY
grep ingest gpurun_out/r3k/probe.log
