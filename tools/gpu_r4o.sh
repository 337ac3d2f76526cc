#!/bin/bash
# CRC-32 of every inflated BGZF block checked on the device: the parser tests, then what the check costs on cfg 3 as bgzip
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r4o
timeout 1200 python3 -m pytest tests/test_gpu_bed_device.py -q -m gpu -x > gpurun_out/r4o/tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4o/tests.log
tail -3 gpurun_out/r4o/tests.log
timeout 900 bash tools/gpu_r4c.sh 2>&1 | grep -E "rep|plain" | head -14
