#!/bin/bash
# round-2 kernel A/B: parity suite on the new kernel variants, then greedy-round timings per variant switch
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r2b
cat /sys/fs/cgroup/cpu.max > gpurun_out/r2b/cpu_quota.txt 2>&1; python3 -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())" >> gpurun_out/r2b/cpu_quota.txt
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_baseline_configs.py --deselect tests/test_gpu_bench_contract.py > gpurun_out/r2b/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2b/tests.log
for env in "X=1" "NM_PREFETCH=1" "NM_NO_CF=1" "NM_NO_LIT=1" "NM_NO_LIT=1 NM_NO_CF=1"; do
  for r in 1 2; do
    NM_PREFETCH=1 timeout 300 python tests/test_gpu_fuzz.py >/dev/null 2>&1
    for w in "--workload greedy --per-group 2" "--workload greedy --per-group 4" "--workload cfg5"; do
      env $env python bench.py $w --steps 20 --warmup 3 --cpu-bins 0 --extras none --hbm-round-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$env', '$w', 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], d['counts_checksum'])" >> gpurun_out/r2b/ab.log
    done
  done
done
NM_PREFETCH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r2b/tests_prefetch.log 2>&1
tail -3 gpurun_out/r2b/tests.log; tail -3 gpurun_out/r2b/tests_prefetch.log; cat gpurun_out/r2b/ab.log; cat gpurun_out/r2b/cpu_quota.txt
