#!/bin/bash
# round 5: the sending thread of the native search (NM_SEARCH_NO_SENDER=1: off) x flights 2 / 3 / 4, two repetitions each on one box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5t
timeout 900 python -m pytest tests/test_gpu_search_speculation.py tests/test_gpu_comm.py tests/test_gpu_windows.py -x -q -m gpu > gpurun_out/r5t/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5t/tests.log
run() {  # name env...
  local name=$1; shift
  env "$@" NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5t/e2e_${name}.log 2>&1
  echo "$name rc=$?"
  grep "nm_search\] 1000" gpurun_out/r5t/e2e_${name}.log | tail -1
  tail -1 gpurun_out/r5t/e2e_${name}.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print({k: round(t.get(k, 0), 4) for k in ('upload_filter_s', 'window_pipeline_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s')}, 'ms/step', round(d.get('ms_per_step'), 2), 'iters', t.get('search_iterations'))
"
}
for rep in 1 2; do
  run nosender_f2_$rep NM_SEARCH_NO_SENDER=1 NM_SEARCH_FLIGHTS=2
  run sender_f2_$rep NM_SEARCH_FLIGHTS=2
  run sender_f3_$rep NM_SEARCH_FLIGHTS=3
  run sender_f4_$rep NM_SEARCH_FLIGHTS=4
done
