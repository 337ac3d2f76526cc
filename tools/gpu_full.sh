#!/bin/bash
# the driver's round-end sequence: whole GPU suite, smoke, default bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/full/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/full/tests.log
tail -6 gpurun_out/full/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.log
echo "bench rc=$?"; tail -2 gpurun_out/full/bench.log
