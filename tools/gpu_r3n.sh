#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3n
export TMPDIR=/tmp
rm -rf /tmp/pb
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/pb -- python3 tools/bed_probe.py 20000000 5 > gpurun_out/r3n/probe.log 2>&1
grep "device parse\|bed " gpurun_out/r3n/probe.log
for f in $(find /tmp/pb -name "*stats.csv"); do echo $f; head -12 $f | cut -c1-160; done
