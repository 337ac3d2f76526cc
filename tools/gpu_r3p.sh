#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3p
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r3p/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3p/tests.log
tail -8 gpurun_out/r3p/tests.log
timeout 1200 python bench.py > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.log
echo "bench rc=$?"; tail -3 gpurun_out/r3p/bench.log
