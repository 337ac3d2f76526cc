#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3e
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r3e/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3e/tests.log
tail -15 gpurun_out/r3e/tests.log
NM_SEARCH_TIMING=1 timeout 600 python tools/bg_probe.py > gpurun_out/r3e/bg.txt 2>&1
tail -3 gpurun_out/r3e/bg.txt
NANOMOTIF_PLAN_PER_TASK=1 timeout 600 python tools/bg_probe.py > gpurun_out/r3e/bg_old.txt 2>&1
tail -2 gpurun_out/r3e/bg_old.txt
