#!/bin/bash
# round 5: host time inside nm_ingest_pileup at 1e9 rows (NM_INGEST_TIMING=1) + the end-to-end run after the plan's vectorised presence table
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5u
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r5u/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5u/tests.log
for rep in 1 2; do
  NM_INGEST_TIMING=1 NM_SEARCH_TIMING=1 NM_PLAN_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5u/e2e_${rep}.log 2>&1
  echo "rep $rep rc=$?"
  grep "nm_ingest\]\|nm_search\] 1000\|nm_plan" gpurun_out/r5u/e2e_${rep}.log | tail -3
  tail -1 gpurun_out/r5u/e2e_${rep}.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print({k: round(t.get(k, 0), 4) for k in ('upload_filter_s', 'window_pipeline_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s')}, 'ms/step', round(d.get('ms_per_step'), 2), 'iters', t.get('search_iterations'))
"
done
