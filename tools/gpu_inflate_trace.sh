#!/bin/bash
# kernel trace of the bgzip device parse (two-phase inflate): per-kernel GPU time of the whole pipeline
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-inflate_trace}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/tools/inflate_pipeline_probe.py 20000000 8 2 ${2:-two} > $R/gpurun_out/$tag/trace.log 2>&1; echo "trace rc=$?"
find /tmp/pp -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
cd $R; grep MODE gpurun_out/$tag/trace.log; head -24 gpurun_out/$tag/kernel_stats.csv | cut -c1-160
