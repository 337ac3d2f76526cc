#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/trace
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 tools/gap_probe.py 125000000 1250 63 1260 > gpurun_out/trace/run.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the cfg5 loop: take 12 consecutive kernels from the middle of the first 200-step loop
sc = [i for i, r in enumerate(rows) if "score_kernel" in r["Kernel_Name"]]
mid = sc[100]
t0 = int(rows[mid]["Start_Timestamp"])
for r in rows[mid - 1: mid + 14]:
    name = r["Kernel_Name"].split("(")[0][-60:]
    print("%-62s start %8.2f us  dur %7.2f us" % (name, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
