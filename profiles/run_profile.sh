#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash profiles/run_profile.sh <round-tag> [bench args...]
# rocprofv3 output goes to /tmp (large); only the per-kernel summaries are copied into gpurun_out/<tag>/ and,
# from there, committed under profiles/.
tag=${1:-r1}; shift
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
B="python3 bench.py --cpu-bins 0 --extras none --hbm-round-steps 0 $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/trace -- $B --steps 10 --warmup 2 > $out/bench_under_trace.log 2>&1
find /tmp/prof_$tag/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find /tmp/prof_$tag/trace -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > '$out'/kernel_trace_score.csv; grep score_kernel "$1" >> '$out'/kernel_trace_score.csv' _ {} \;
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 900 rocprofv3 --pmc $pass --output-format csv -d /tmp/prof_$tag/pmc_$name -- $B --steps 3 --warmup 1 --prewarm 0 > $out/bench_under_pmc_$name.log 2>&1
  find /tmp/prof_$tag/pmc_$name -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > '$out'/pmc_'$name'.csv; grep score_kernel "$1" >> '$out'/pmc_'$name'.csv' _ {} \;
done
grep -h '"metric"' $out/bench_under_trace.log | tail -1 > $out/bench_under_trace.json
ls -la $out
