#!/bin/bash
# round-2 profiles: headline cfg5 batch, the greedy round (2 children per group), the all-bins table, the e2e pipeline
cd ${GRAFT_REPO_ROOT:-.}
bash profiles/run_profile.sh r2_cfg5 > /dev/null 2>&1
bash profiles/run_profile.sh r2_greedy2 --workload greedy --per-group 2 > /dev/null 2>&1
bash profiles/run_profile.sh r2_greedy4 --workload greedy --per-group 4 > /dev/null 2>&1
bash profiles/run_profile_e2e.sh r2_e2e > /dev/null 2>&1
ls gpurun_out/prof_r2_cfg5 gpurun_out/prof_r2_greedy2 gpurun_out/prof_r2_e2e | head -40
cat gpurun_out/prof_r2_cfg5/kernel_stats.csv | head -5
