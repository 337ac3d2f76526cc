#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r2d
timeout 1700 python -m pytest tests/test_gpu_comm.py tests/test_gpu_windows.py tests/test_gpu_ingest.py tests/test_gpu_cli.py tests/test_gpu_parity.py tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r2d/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2d/tests.log
tail -15 gpurun_out/r2d/tests.log
timeout 600 python bench.py > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.log
echo "bench rc=$?"; tail -3 gpurun_out/r2d/bench.log
