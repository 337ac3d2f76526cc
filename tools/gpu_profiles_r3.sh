#!/bin/bash
# round 3: every profile the summaries under profiles/r3 come from (one gpurun call, ~25 GPU-minutes)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k extra_wide 2>&1 | tail -2
bash profiles/run_profile.sh r3 > /dev/null 2>&1
bash profiles/run_profile.sh r3_greedy2 --workload greedy --per-group 2 > /dev/null 2>&1
bash profiles/run_profile.sh r3_greedy4 --workload greedy --per-group 4 > /dev/null 2>&1
bash profiles/run_profile.sh r3_cfg5all --workload cfg5_all > /dev/null 2>&1
bash profiles/run_profile_e2e.sh r3_e2e > /dev/null 2>&1
bash tools/gpu_ingest_prof.sh prof_r3_ingest > /dev/null 2>&1
bash tools/gpu_ingest_pmc.sh prof_r3_ingest_pmc > gpurun_out/prof_r3_ingest_pmc_summary.txt 2>&1
mkdir -p gpurun_out/prof_r3_meth gpurun_out/prof_r3_bed
rm -rf /tmp/pm; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- python3 tools/meth_probe.py > gpurun_out/prof_r3_meth/probe.log 2>&1
find /tmp/pm -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r3_meth/kernel_stats.csv \;
rm -rf /tmp/pb; rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/pb -- python3 tools/bed_probe.py 20000000 5 > gpurun_out/prof_r3_bed/probe.log 2>&1
find /tmp/pb -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r3_bed/kernel_stats.csv \;
find /tmp/pb -name "*memory_copy_stats.csv" -exec cp {} gpurun_out/prof_r3_bed/memory_copy_stats.csv \;
grep -h "readstats\|contig methylation" gpurun_out/prof_r3_meth/probe.log
grep -h "device parse" gpurun_out/prof_r3_bed/probe.log
ls gpurun_out/prof_r3 gpurun_out/prof_r3_e2e | head -30
