#!/bin/bash
# round 5: the four score-workload profiles again (nmscan.hip / nmscan_internal.h changed: nm_score_batch_wide, flights) -> profiles/traffic.json is re-keyed from them
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=r5b
bash profiles/run_profile.sh ${R} > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_greedy2 --workload greedy --per-group 2 > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_greedy4 --workload greedy --per-group 4 > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_cfg5all --workload cfg5_all > /dev/null 2>&1
ls gpurun_out/prof_${R} gpurun_out/prof_${R}_greedy2 | head -40
cat gpurun_out/prof_${R}/bench_under_trace.json | cut -c1-400
