#!/bin/bash
# the bgzip device parse (both inflate forms) with per-slab timing, then under rocprofv3 --kernel-trace --stats (two-phase only)
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-inflate_pipeline}
mkdir -p gpurun_out/$tag
timeout 900 python3 tools/inflate_pipeline_probe.py 20000000 8 2 > gpurun_out/$tag/probe.log 2>&1; echo "probe rc=$?"
grep -E "MODE|slab 1:|slab 2:|allocated|text " gpurun_out/$tag/probe.log | cut -c1-260
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/tools/inflate_pipeline_probe.py 20000000 8 1 two > $R/gpurun_out/$tag/trace.log 2>&1; echo "trace rc=$?"
find /tmp/pp -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
cd $R; head -14 gpurun_out/$tag/kernel_stats.csv | cut -c1-200
