#!/bin/bash
# round 4: HIP API time of the 1 Gbp end-to-end run (which calls a lock-step round is made of)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
mkdir -p gpurun_out/r4k
rm -rf /tmp/ph
timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/ph -- python3 bench.py --workload e2e > gpurun_out/r4k/bench.log 2>&1
f=$(find /tmp/ph -name "*hip_api_stats.csv" | head -1)
cp $f gpurun_out/r4k/hip_api_stats.csv
head -25 $f | cut -d, -f1-6
