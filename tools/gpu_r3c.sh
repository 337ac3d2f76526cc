#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_ingest.py -q -m gpu > gpurun_out/r3c/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3c/tests.log
tail -60 gpurun_out/r3c/tests.log
