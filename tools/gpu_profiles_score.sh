#!/bin/bash
# the four score-workload profiles only (what profiles/traffic.json is keyed to: nmscan.hip + nmscan_device.h + nmscan_internal.h):
#   tools/gpu_profiles_score.sh r6   then   python profiles/summarize.py gpurun_out/prof_r6 r6 [; ... r6_greedy2 r6 greedy2 ; ...]
R=${1:-r6}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
bash profiles/run_profile.sh ${R} > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_greedy2 --workload greedy --per-group 2 > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_greedy4 --workload greedy --per-group 4 > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_cfg5all --workload cfg5_all > /dev/null 2>&1
ls gpurun_out/prof_${R} | head -20
