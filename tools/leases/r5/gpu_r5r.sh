#!/bin/bash
# round 5: how many groups of tasks the native search keeps in flight (NM_SEARCH_FLIGHTS = 2 / 3 / 4), two repetitions each on one box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5r
timeout 600 python -m pytest tests/test_gpu_search_speculation.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r5r/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5r/tests.log
for rep in 1 2; do
for F in 2 3 4; do
  NM_SEARCH_FLIGHTS=$F NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5r/e2e_f${F}_${rep}.log 2>&1
  echo "flights $F rep $rep rc=$?"
  grep "nm_search\]" gpurun_out/r5r/e2e_f${F}_${rep}.log | tail -2
  tail -1 gpurun_out/r5r/e2e_f${F}_${rep}.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print({k: round(t.get(k, 0), 4) for k in ('upload_filter_s', 'window_pipeline_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s')}, 'ms/step', d.get('ms_per_step'), 'rows', d.get('motif_rows', d.get('e2e', {}).get('motif_rows')), 'iters', t.get('search_iterations'))
"
done
done
