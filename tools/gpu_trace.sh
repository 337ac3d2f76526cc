#!/bin/bash
# kernel trace of consecutive scoring steps: start / duration / queue of every kernel around the middle of the loop
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/trace
LANES=${LANES:-1}
SHAPE=${SHAPE:-125000000 1250 63 1260}
rm -rf /tmp/tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 tools/gap_probe.py $SHAPE $LANES > gpurun_out/trace/run_$LANES.log 2>&1
tail -2 gpurun_out/trace/run_$LANES.log
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sc = [i for i, r in enumerate(rows) if "score_kernel" in r["Kernel_Name"]]
mid = sc[100]
t0 = int(rows[mid]["Start_Timestamp"])
print("columns:", [k for k in rows[0].keys()])
for r in rows[mid - 1: mid + 16]:
    name = r["Kernel_Name"].split("(")[0][-50:]
    print("%-52s q %-3s start %8.2f us  end %8.2f us  dur %7.2f us" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
