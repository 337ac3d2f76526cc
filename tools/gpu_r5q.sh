#!/bin/bash
# round 5: where the host time of the 1 Gbp end-to-end run goes (cProfile over 3 runs; NM_PLAN_TIMING / NM_SEARCH_TIMING of one more)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5q
timeout 900 python tools/discover_profile.py > gpurun_out/r5q/profile.log 2>&1; echo "rc=$?"
NM_PLAN_TIMING=1 NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5q/e2e_timing.log 2>&1; echo "rc=$?"
grep -v "^{" gpurun_out/r5q/e2e_timing.log | tail -40
tail -90 gpurun_out/r5q/profile.log
