#!/bin/bash
# every profile the summaries under profiles/<round> come from (one gpurun call, ~25 GPU-minutes):  tools/gpu_profiles.sh r5
R=${1:-r5}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
bash profiles/run_profile.sh ${R} > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_greedy2 --workload greedy --per-group 2 > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_greedy4 --workload greedy --per-group 4 > /dev/null 2>&1
bash profiles/run_profile.sh ${R}_cfg5all --workload cfg5_all > /dev/null 2>&1
bash profiles/run_profile_e2e.sh ${R}_e2e > /dev/null 2>&1
bash tools/gpu_ingest_prof.sh prof_${R}_ingest > /dev/null 2>&1
bash tools/gpu_ingest_pmc.sh prof_${R}_ingest_pmc > gpurun_out/prof_${R}_ingest_pmc_summary.txt 2>&1
mkdir -p gpurun_out/prof_${R}_meth gpurun_out/prof_${R}_bed
rm -rf /tmp/pm; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- python3 tools/meth_probe.py > gpurun_out/prof_${R}_meth/probe.log 2>&1
find /tmp/pm -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${R}_meth/kernel_stats.csv \;
rm -rf /tmp/pb; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/pb -- python3 tools/bed_probe.py 20000000 5 > gpurun_out/prof_${R}_bed/probe.log 2>&1
find /tmp/pb -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${R}_bed/kernel_stats.csv \;
find /tmp/pb -name "*memory_copy_stats.csv" -exec cp {} gpurun_out/prof_${R}_bed/memory_copy_stats.csv \;
grep -h "readstats\|contig methylation" gpurun_out/prof_${R}_meth/probe.log
grep -h "device parse" gpurun_out/prof_${R}_bed/probe.log
cat gpurun_out/prof_${R}_ingest_pmc_summary.txt | tail -8
ls gpurun_out/prof_${R} gpurun_out/prof_${R}_e2e | head -30
