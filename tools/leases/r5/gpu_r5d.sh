#!/bin/bash
# round 5: speculative children of the search — identical rows with and without, the suites that run the native search, the 1 Gbp
# end-to-end run with NM_SEARCH_TIMING in both modes
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5d
timeout 1500 python -m pytest tests/test_gpu_search_speculation.py tests/test_gpu_synth.py tests/test_gpu_windows.py tests/test_gpu_cli.py tests/test_gpu_comm.py -x -q -m gpu > gpurun_out/r5d/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5d/tests.log
tail -25 gpurun_out/r5d/tests.log
for mode in spec nospec spec nospec; do
  if [ $mode = nospec ]; then export NM_SEARCH_NO_SPEC=1; else unset NM_SEARCH_NO_SPEC; fi
  NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/r5d/e2e_$mode.log 2>&1
  echo "== $mode rc=$?"; grep "nm_search" gpurun_out/r5d/e2e_$mode.log | tail -3
  tail -1 gpurun_out/r5d/e2e_$mode.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); t = d['timings_rank0']; print('wall', d['value'], {k: t.get(k) for k in ('gpu_busy_s', 'search_s', 'native_search_s', 'postprocess_s', 'upload_filter_s', 'rounds', 'search_iterations', 'speculation_hits', 'speculation_misses')}, d['per_rank'])"
done
