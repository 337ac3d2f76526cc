#!/bin/bash
# round 5: window counts asked for AHEAD in the tail of the search (NM_SEARCH_NO_AHEAD=1: off): search tests, then the end-to-end run, two repetitions each
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5ai
timeout 1500 python -m pytest tests/test_gpu_search_speculation.py tests/test_gpu_baseline_configs.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r5ai/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5ai/tests.log
run() {
  local name=$1; shift
  env "$@" NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5ai/e2e_${name}.log 2>&1
  echo "$name rc=$?"
  grep "nm_search\]" gpurun_out/r5ai/e2e_${name}.log | tail -2 | cut -c1-330
  tail -1 gpurun_out/r5ai/e2e_${name}.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print({k: round(t.get(k, 0), 4) for k in ('upload_filter_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s')}, 'ms/step', round(d.get('ms_per_step'), 2), 'rows', t.get('motif_rows'))
"
}
for rep in 1 2; do
  run ahead_$rep NM_X=1
  run noahead_$rep NM_SEARCH_NO_AHEAD=1
done
