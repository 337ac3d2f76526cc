#!/bin/bash
# round 5: the draws kernel's bitmap in LDS (NM_DRAW_GLOBAL_BITMAP=1: as before), where nm_post_run's time goes (NM_POST_TIMING)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5ae
timeout 900 python -m pytest tests/test_gpu_windows.py tests/test_gpu_baseline_configs.py -x -q -m gpu > gpurun_out/r5ae/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5ae/tests.log
run() {
  local name=$1; shift
  env "$@" NM_PLAN_TIMING=1 NM_POST_TIMING=1 timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 > gpurun_out/r5ae/e2e_${name}.log 2>&1
  echo "$name rc=$?"
  grep "nm_plan_windows\]\|nm_post\]" gpurun_out/r5ae/e2e_${name}.log | tail -2
  tail -1 gpurun_out/r5ae/e2e_${name}.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); t = d.get('timings_rank0', d.get('e2e', {}).get('timings_rank0', {}))
print({k: round(t.get(k, 0), 4) for k in ('upload_filter_s', 'window_pipeline_s', 'plan_s', 'background_s', 'native_search_s', 'postprocess_s', 'gpu_busy_s')}, 'ms/step', round(d.get('ms_per_step'), 2))
"
}
for rep in 1 2; do
  run lds_$rep NM_X=1
  run global_$rep NM_DRAW_GLOBAL_BITMAP=1
done
