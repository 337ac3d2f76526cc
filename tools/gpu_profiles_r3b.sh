#!/bin/bash
# round 3, closing: the end-to-end and ingest profiles again on the final code (device draws, list-based adjacency test)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
bash profiles/run_profile_e2e.sh r3_e2e > /dev/null 2>&1
bash tools/gpu_ingest_prof.sh prof_r3_ingest > /dev/null 2>&1
bash tools/gpu_ingest_pmc.sh prof_r3_ingest_pmc > gpurun_out/prof_r3_ingest_pmc_summary.txt 2>&1
ls gpurun_out/prof_r3_e2e gpurun_out/prof_r3_ingest gpurun_out/prof_r3_ingest_pmc
