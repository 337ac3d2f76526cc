#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3r
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r3r/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3r/tests.log
tail -30 gpurun_out/r3r/tests.log
timeout 900 python bench.py --extras e2e,two_lanes --cpu-bins 0 2>/dev/null | python tools/bench_brief.py
