#!/bin/bash
# round 5: Infinity Cache probe of the pre-filters (prefixes of the 1e9-row pileup), per-launch durations of count / decide
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5z
export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pm -- python3 tools/ingest_mall_probe.py > gpurun_out/r5z/probe.log 2>&1; echo "rc=$?"
grep "PREFIX\|runs" gpurun_out/r5z/probe.log
f=$(find /tmp/pm -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/r5z/per_launch.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ingest_count_kernel" in r["Kernel_Name"] or "ingest_decide_kernel" in r["Kernel_Name"] or "ingest_judge" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    name = "count" if "count" in r["Kernel_Name"] else "decide" if "decide" in r["Kernel_Name"] else "judge"
    print(name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us  grid", r.get("Grid_Size"), r.get("Grid_Size_X", ""))
PY
