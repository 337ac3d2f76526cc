#!/bin/bash
# round-2 first GPU pass: new BASELINE-shape tests, bench contract, default bench line, e2e host profile
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/host.txt; free -g >> gpurun_out/r2a/host.txt
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r2a/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2a/tests.log
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.log
echo "bench rc=$?" >> gpurun_out/r2a/bench.log
timeout 300 python -m cProfile -o gpurun_out/r2a/e2e.prof bench.py --workload e2e > gpurun_out/r2a/e2e.json 2> gpurun_out/r2a/e2e.log
tail -5 gpurun_out/r2a/tests.log; tail -3 gpurun_out/r2a/bench.log; head -c 600 gpurun_out/r2a/bench.json
