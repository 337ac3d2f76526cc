#!/bin/bash
# the GPU suite (or the tests given as arguments) on one box; log under gpurun_out/tests/
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/tests
timeout 3000 python -m pytest ${@:-tests} -x -q -m gpu > gpurun_out/tests/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/tests/tests.log
tail -15 gpurun_out/tests/tests.log
