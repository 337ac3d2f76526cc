#!/bin/bash
# per-kernel split of the device-side pre-filters at 1e9 rows
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-ingest}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
rm -rf /tmp/pi
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- python3 tools/ingest_probe.py > gpurun_out/$tag/probe.log 2>&1
cat gpurun_out/$tag/probe.log | grep ingest
f=$(find /tmp/pi -name "*kernel_stats.csv" | head -1)
grep -i "ingest\|popcount\|compact" $f | cut -d, -f1-4 | sed 's/(.*)"/"/' | cut -c1-120
cp $f gpurun_out/$tag/kernel_stats.csv
