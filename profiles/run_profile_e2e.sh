#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash profiles/run_profile_e2e.sh <tag>
# Per-kernel time of the whole motif_discovery pipeline (bench.py --workload e2e) under rocprofv3 --kernel-trace --stats.
tag=${1:-r1_e2e}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/trace -- python3 bench.py --workload e2e > $out/bench_under_trace.log 2>&1
find /tmp/prof_$tag/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats_all.csv \;
grep -h '"metric"' $out/bench_under_trace.log | tail -1 > $out/bench_under_trace.json
ls -la $out
