#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_synth.py tests/test_gpu_windows.py -x -q -m gpu > gpurun_out/r2c/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2c/tests.log
tail -4 gpurun_out/r2c/tests.log
bash tools/gpu_ab.sh "X=1;NM_SEG_CHUNKS=8;NM_SEG_CHUNKS=4;NM_SEG_CHUNKS=32" 2 dyn
