#!/bin/bash
# round 5: staging by kernels instead of copy engines: parity suites + the end-to-end A/B (speculation on / off, NM_STAGE_COPIES=1 = the copy engines)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5f
timeout 1500 python -m pytest tests/test_gpu_search_speculation.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_windows.py tests/test_gpu_comm.py tests/test_gpu_synth.py -x -q -m gpu > gpurun_out/r5f/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5f/tests.log
tail -6 gpurun_out/r5f/tests.log
run() {
  tag=$1; shift
  env "$@" NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/r5f/e2e_$tag.log 2>&1
  echo "== $tag rc=$?"; grep "nm_search" gpurun_out/r5f/e2e_$tag.log | tail -2
  tail -1 gpurun_out/r5f/e2e_$tag.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); t = d['timings_rank0']; print('wall', round(d['value'], 4), {k: round(t.get(k), 4) for k in ('gpu_busy_s', 'search_s', 'native_search_s', 'postprocess_s', 'upload_filter_s', 'background_s', 'plan_s', 'window_pipeline_s')})"
}
for rep in 1 2; do
run spec A=1
run nospec NM_SEARCH_NO_SPEC=1
run spec_copies NM_STAGE_COPIES=1
run nospec_copies NM_SEARCH_NO_SPEC=1 NM_STAGE_COPIES=1
run spec_oneflight NM_SEARCH_ONE_FLIGHT=1
done
