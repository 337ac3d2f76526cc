#!/bin/bash
# round 5: frames above 191 (nm_score_batch_wide, host windows): the two new tests first, then the whole GPU suite
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5p
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -m gpu -k "wide or frames or reach" > gpurun_out/r5p/wide.log 2>&1
echo "wide rc=$?"; tail -30 gpurun_out/r5p/wide.log
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r5p/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5p/tests.log
tail -5 gpurun_out/r5p/tests.log
